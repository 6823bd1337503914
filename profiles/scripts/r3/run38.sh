#!/usr/bin/env bash
# DP readiness after the round's kernel changes: RCCL world-1 capture (bit-identity), its timing, the two-rank gloo rehearsal
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
export MASTER_ADDR=127.0.0.1
{ timeout -k 10 300 python3 tools/dp_rehearsal.py rccl1 && timeout -k 10 300 python3 tools/dp_rehearsal.py rccl1time && timeout -k 10 600 python3 tools/dp_rehearsal.py 2; } 2>&1 | grep -v amdgpu > "$out/r3_dp38.txt"
echo "rc=$?" >> "$out/r3_dp38.txt"
tail -20 "$out/r3_dp38.txt"
