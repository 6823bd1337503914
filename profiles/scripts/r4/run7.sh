#!/usr/bin/env bash
# round 4, call 7: full GPU suite; the driver-style default bench run; RCCL world-1 rehearsal records
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 1500 python3 -m pytest tests -m gpu -q > "$out/r4_t7.log" 2>&1; echo "rc=$?" >> "$out/r4_t7.log"
tail -12 "$out/r4_t7.log"
timeout -k 10 120 python3 tools/kbench.py l1 > "$out/r4_l1f.log" 2>&1; cat "$out/r4_l1f.log"
( time timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 ) > "$out/r4_bench7.log" 2>&1; echo "rc=$?" >> "$out/r4_bench7.log"
tail -6 "$out/r4_bench7.log" | cut -c1-400
timeout -k 10 300 python3 tools/dp_rehearsal.py rccl1 > "$out/r4_rccl1b.log" 2>&1; echo "rc=$?" >> "$out/r4_rccl1b.log"; tail -3 "$out/r4_rccl1b.log"
timeout -k 10 300 python3 tools/dp_rehearsal.py rccl1time > "$out/r4_rccl1timeb.log" 2>&1; echo "rc=$?" >> "$out/r4_rccl1timeb.log"; tail -2 "$out/r4_rccl1timeb.log"
