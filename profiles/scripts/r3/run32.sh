#!/usr/bin/env bash
# grouped Linear weight-gradient launch: workgroups per GEMM (MM_WGM_TARGET), step time
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
rm -f "$out/r3_wgm.log"
for rep in 1 2; do
  for t in 32 24 48 64 96; do
    echo "== MM_WGM_TARGET=$t (rep $rep)" >> "$out/r3_wgm.log"
    MM_WGM_TARGET=$t python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> "$out/r3_wgm.log"
  done
done
cat "$out/r3_wgm.log"
