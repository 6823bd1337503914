cd $GRAFT_REPO_ROOT
B=$PWD/multimodal_eeg_fmri_amd/csrc/build
export MM_NO_CONV1D_WRES=1
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or attn" > gpurun_out/r3_t6.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t6.log
tail -n 3 gpurun_out/r3_t6.log
: > gpurun_out/r3_attn3.log
for v in prev nobwdpipe prod; do
  echo "== $v" >> gpurun_out/r3_attn3.log
  if [ $v = prod ]; then unset MMEEG_HIP_LIB; else export MMEEG_HIP_LIB=$B/alt_$v.so; fi
  timeout -k 10 120 python tools/kbench.py attn >> gpurun_out/r3_attn3.log 2>&1
done
grep -v amdgpu gpurun_out/r3_attn3.log
: > gpurun_out/r3_step_ab4.log
for rep in 1 2; do
for v in prev nobwdpipe prod; do
  if [ $v = prod ]; then unset MMEEG_HIP_LIB; else export MMEEG_HIP_LIB=$B/alt_$v.so; fi
  echo "== $v (rep $rep)" >> gpurun_out/r3_step_ab4.log
  timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 --profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/r3_step_ab4.log 2>&1
done
done
unset MMEEG_HIP_LIB
cat gpurun_out/r3_step_ab4.log
timeout -k 10 600 python bench.py --steps 50 --warmup 10 > gpurun_out/r3_bench3.log 2>&1; tail -c 2500 gpurun_out/r3_bench3.log
tail -n 3 gpurun_out/r3_t7.log
