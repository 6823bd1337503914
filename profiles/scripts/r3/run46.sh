#!/usr/bin/env bash
# voxel head's gradient as one row per sample (bcast BatchNorm-backward passes): tests, A/B (MM_NO_BCAST=1 = full tensor)
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$out/r3_t47.log" 2>&1
echo "rc=$?" >> "$out/r3_t47.log"
tail -4 "$out/r3_t47.log"
grep -q "rc=0" "$out/r3_t47.log" || exit 1
rm -f "$out/r3_ab47.log"
for rep in 1 2 3; do
  for v in 1 0; do
    if [ $v = 1 ]; then export MM_NO_BCAST=1; else unset MM_NO_BCAST; fi
    echo "== no_bcast=$v (rep $rep)" >> "$out/r3_ab47.log"
    python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> "$out/r3_ab47.log"
  done
done
cat "$out/r3_ab47.log"
