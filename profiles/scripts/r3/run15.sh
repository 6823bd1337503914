cd $GRAFT_REPO_ROOT
B=$PWD/multimodal_eeg_fmri_amd/csrc/build
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r3_full_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3_full_gpu.log
tail -n 3 gpurun_out/r3_full_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1; tail -n 1 gpurun_out/r3_smoke.log
: > gpurun_out/r3_step_final.log
for rep in 1 2 3; do
for v in base prod; do
  if [ $v = prod ]; then unset MMEEG_HIP_LIB; else export MMEEG_HIP_LIB=$B/alt_$v.so; fi
  echo "== $v (rep $rep)" >> gpurun_out/r3_step_final.log
  timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 --profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/r3_step_final.log 2>&1
done
done
unset MMEEG_HIP_LIB
cat gpurun_out/r3_step_final.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_driver_style.log 2>&1; tail -c 600 gpurun_out/r3_bench_driver_style.log
