#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
for rep in 1 2; do for v in 512 768 1024 2048; do
  echo "rep $rep MM_POOL_FIN_GRID=$v: $(MM_POOL_FIN_GRID=$v timeout -k 10 200 python3 tools/h2d_probe.py 300 2>&1 | grep 'resident batches (mm' | cut -c80-118)"
done; done
