#!/usr/bin/env bash
# which conv-block weight gradients leave the chain: blocks 3 + 2 (2), block 2 only (1), none (0)
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
rm -f "$out/r3_ab48.log"
for rep in 1 2 3; do
  for v in 2 1 0; do
    echo "== handed=$v (rep $rep)" >> "$out/r3_ab48.log"
    MM_CONV_WGRADS_HANDED=$v python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> "$out/r3_ab48.log"
  done
done
cat "$out/r3_ab48.log"
