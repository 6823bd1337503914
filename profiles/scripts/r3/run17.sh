cd $GRAFT_REPO_ROOT
: > gpurun_out/r3_step_ab6.log
for rep in 1 2; do
for v in 256 192 128; do
  export MM_W3_CUS=$v
  echo "== MM_W3_CUS=$v (rep $rep)" >> gpurun_out/r3_step_ab6.log
  timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 --profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> gpurun_out/r3_step_ab6.log 2>&1
done
done
cat gpurun_out/r3_step_ab6.log
