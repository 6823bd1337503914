#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
MM_EARLY_FORK=1 timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps 2>&1 | grep -v amdgpu | grep -v "^{" > "$out/r4_early_fork.log"; cat "$out/r4_early_fork.log"
