#!/usr/bin/env bash
# grid cap of the BatchNorm apply passes that carry the finalize in their prologue (each workgroup re-reads the statistics)
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
for rep in 1 2; do for v in 512 256 768 1024 2048; do
  echo "rep $rep MM_FIN_GRID=$v: $(MM_FIN_GRID=$v timeout -k 10 200 python3 tools/h2d_probe.py 300 2>&1 | grep 'resident batches (mm' | cut -c80-130)"
done; done
