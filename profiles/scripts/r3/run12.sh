cd $GRAFT_REPO_ROOT
B=$PWD/multimodal_eeg_fmri_amd/csrc/build
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv3d or attention" > gpurun_out/r3_t8.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3_t8.log
tail -n 3 gpurun_out/r3_t8.log
if [ $rc -ne 0 ]; then grep -n "Error\|assert" gpurun_out/r3_t8.log | head; exit 1; fi
echo "== prev" > gpurun_out/r3_wgrad3.log
MMEEG_HIP_LIB=$B/alt_prev.so timeout -k 10 120 python tools/kbench.py wgrad3 >> gpurun_out/r3_wgrad3.log 2>&1
echo "== prod" >> gpurun_out/r3_wgrad3.log
timeout -k 10 120 python tools/kbench.py wgrad3 >> gpurun_out/r3_wgrad3.log 2>&1
grep -v amdgpu gpurun_out/r3_wgrad3.log
: > gpurun_out/r3_step_ab5.log
for rep in 1 2; do
for v in nobwdpipe prod; do
  if [ $v = prod ]; then unset MMEEG_HIP_LIB; else export MMEEG_HIP_LIB=$B/alt_$v.so; fi
  echo "== $v (rep $rep)" >> gpurun_out/r3_step_ab5.log
  timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 --profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/r3_step_ab5.log 2>&1
done
done
unset MMEEG_HIP_LIB
cat gpurun_out/r3_step_ab5.log
timeout -k 10 900 python -m pytest tests/test_trainer_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "volume or c2 or trainer or repro" > gpurun_out/r3_t9.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t9.log
tail -n 3 gpurun_out/r3_t9.log
