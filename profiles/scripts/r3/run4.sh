cd $GRAFT_REPO_ROOT
B=$PWD/multimodal_eeg_fmri_amd/csrc/build
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -x -q -m gpu > gpurun_out/r3_t3.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t3.log
: > gpurun_out/r3_attn.log
for v in base prod vf; do
  echo "== $v" >> gpurun_out/r3_attn.log
  if [ $v = prod ]; then unset MMEEG_HIP_LIB; else export MMEEG_HIP_LIB=$B/alt_$v.so; fi
  timeout -k 10 120 python tools/kbench.py attn >> gpurun_out/r3_attn.log 2>&1
done
: > gpurun_out/r3_step_ab.log
for rep in 1 2; do
for v in base prod vf; do
  if [ $v = prod ]; then unset MMEEG_HIP_LIB; else export MMEEG_HIP_LIB=$B/alt_$v.so; fi
  echo "== $v (rep $rep)" >> gpurun_out/r3_step_ab.log
  timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 --profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/r3_step_ab.log 2>&1
done
done
unset MMEEG_HIP_LIB
tail -n 5 gpurun_out/r3_t3.log; grep -v amdgpu.ids gpurun_out/r3_attn.log; cat gpurun_out/r3_step_ab.log
