#!/usr/bin/env bash
# grouped Linear weight gradients: the same number of row chunks for every GEMM (MM_WGM_CHUNKS) vs 32 workgroups per GEMM
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
rm -f "$out/r3_wgmc.log"
for rep in 1 2; do
  for c in 0 8 9 10 11 12 16 20; do
    echo "== MM_WGM_CHUNKS=$c (rep $rep)" >> "$out/r3_wgmc.log"
    MM_WGM_CHUNKS=$c python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> "$out/r3_wgmc.log"
  done
done
cat "$out/r3_wgmc.log"
