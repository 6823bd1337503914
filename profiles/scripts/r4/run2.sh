#!/usr/bin/env bash
# round 4, call 2: full GPU suite after the routing-check fix; H2D probe with deeper staging rings and more hardware queues
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 1500 python3 -m pytest tests -m gpu -q > "$out/r4_t2.log" 2>&1; echo "rc=$?" >> "$out/r4_t2.log"
tail -25 "$out/r4_t2.log"
timeout -k 10 300 python3 tools/h2d_probe.py 200 > "$out/r4_h2d2.log" 2>&1; echo "rc=$?" >> "$out/r4_h2d2.log"; cat "$out/r4_h2d2.log"
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python3 tools/h2d_probe.py 200 > "$out/r4_h2d2_q8.log" 2>&1; echo "rc=$?" >> "$out/r4_h2d2_q8.log"; echo "== GPU_MAX_HW_QUEUES=8"; grep -v "^  .*MB:" "$out/r4_h2d2_q8.log"
