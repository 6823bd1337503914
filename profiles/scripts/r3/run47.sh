#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "row_form or second_gemm or one_dout_row" 2>&1 | tail -3
