cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fit-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['value_with_input_transfer'], d['roofline']['frac'], d['roofline_c4']['frac'])" >> gpurun_out/r3_bench20.log 2>&1
done
cat gpurun_out/r3_bench20.log
timeout -k 10 600 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu -k "bench" > gpurun_out/r3_t11.log 2>&1; tail -n 2 gpurun_out/r3_t11.log
