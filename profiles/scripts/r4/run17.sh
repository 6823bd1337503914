#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 300 python3 tools/dp_rehearsal.py rccl1 > "$out/r4_abort_rehearsal.log" 2>&1; echo "rc=$?" >> "$out/r4_abort_rehearsal.log"
grep -v "^frame #" "$out/r4_abort_rehearsal.log" | tail -40 | cut -c1-600
