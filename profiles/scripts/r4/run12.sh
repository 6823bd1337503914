#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 600 python3 -m pytest tests/test_trainer_gpu.py -m gpu -x -q -k "config5" > "$out/r4_t12.log" 2>&1; echo "rc=$?" >> "$out/r4_t12.log"
tail -30 "$out/r4_t12.log"
