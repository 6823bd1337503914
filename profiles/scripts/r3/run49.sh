#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests/test_trainer_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])"
done
