#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
timeout -k 10 600 python3 -m pytest tests/test_trainer_gpu.py -x -q -k "fused_launches" 2>&1 | tail -15
