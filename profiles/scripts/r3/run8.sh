cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/r3_t4.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3_t4.log
tail -n 3 gpurun_out/r3_t4.log
timeout -k 10 1000 python -m pytest tests/test_trainer_gpu.py tests/test_pipelines_gpu.py tests/test_models_gpu.py -x -q -m gpu > gpurun_out/r3_t5.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t5.log
tail -n 3 gpurun_out/r3_t5.log
: > gpurun_out/r3_step_ab3.log
for rep in 1 2; do
for v in nowres prod; do
  if [ $v = prod ]; then unset MM_NO_CONV1D_WRES; else export MM_NO_CONV1D_WRES=1; fi
  echo "== $v (rep $rep)" >> gpurun_out/r3_step_ab3.log
  timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 --profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/r3_step_ab3.log 2>&1
done
done
unset MM_NO_CONV1D_WRES
cat gpurun_out/r3_step_ab3.log
timeout -k 10 200 python tools/kbench.py conv1 > gpurun_out/r3_conv1.log 2>&1; grep -v amdgpu gpurun_out/r3_conv1.log
MM_NO_CONV1D_WRES=1 timeout -k 10 200 python tools/kbench.py conv1 > gpurun_out/r3_conv1_old.log 2>&1; grep -v amdgpu gpurun_out/r3_conv1_old.log
timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps > gpurun_out/r3_stamps.log 2>&1
grep -v "^{" gpurun_out/r3_stamps.log | grep -v amdgpu
