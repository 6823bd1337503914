cd $GRAFT_REPO_ROOT
B=$PWD/multimodal_eeg_fmri_amd/csrc/build
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r3_t4.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3_t4.log
tail -n 3 gpurun_out/r3_t4.log
if [ $rc -ne 0 ]; then exit 1; fi
: > gpurun_out/r3_attn2.log
for v in head prod; do
  echo "== $v" >> gpurun_out/r3_attn2.log
  if [ $v = prod ]; then unset MMEEG_HIP_LIB; else export MMEEG_HIP_LIB=$B/alt_attn_$v.so; fi
  timeout -k 10 120 python tools/kbench.py attn >> gpurun_out/r3_attn2.log 2>&1
done
: > gpurun_out/r3_step_ab2.log
for rep in 1 2; do
for v in head prod; do
  if [ $v = prod ]; then unset MMEEG_HIP_LIB; else export MMEEG_HIP_LIB=$B/alt_attn_$v.so; fi
  echo "== $v (rep $rep)" >> gpurun_out/r3_step_ab2.log
  timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 --profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/r3_step_ab2.log 2>&1
done
done
unset MMEEG_HIP_LIB
grep -v amdgpu.ids gpurun_out/r3_attn2.log; cat gpurun_out/r3_step_ab2.log
timeout -k 10 1000 python -m pytest tests/test_trainer_gpu.py tests/test_pipelines_gpu.py tests/test_models_gpu.py -x -q -m gpu > gpurun_out/r3_t5.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t5.log
tail -n 3 gpurun_out/r3_t5.log
timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps > gpurun_out/r3_stamps.log 2>&1
grep -v "^{" gpurun_out/r3_stamps.log | grep -v amdgpu
