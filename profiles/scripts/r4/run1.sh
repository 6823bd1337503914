#!/usr/bin/env bash
# round 4, call 1: full GPU suite on the new host logic, then the measurements the next steps are planned on:
# H2D probe, RCCL world-1 rehearsal with REAL all-reduce nodes, layer-1 voxel kernels alone + PMC
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > "$out/r4_t1.log" 2>&1; echo "rc=$?" >> "$out/r4_t1.log"
tail -15 "$out/r4_t1.log"
timeout -k 10 300 python3 tools/h2d_probe.py 200 > "$out/r4_h2d.log" 2>&1; echo "rc=$?" >> "$out/r4_h2d.log"; cat "$out/r4_h2d.log"
timeout -k 10 300 python3 tools/dp_rehearsal.py rccl1 > "$out/r4_rccl1.log" 2>&1; echo "rc=$?" >> "$out/r4_rccl1.log"; tail -5 "$out/r4_rccl1.log"
timeout -k 10 300 python3 tools/dp_rehearsal.py rccl1time > "$out/r4_rccl1time.log" 2>&1; echo "rc=$?" >> "$out/r4_rccl1time.log"; tail -3 "$out/r4_rccl1time.log"
timeout -k 10 120 python3 tools/kbench.py l1 > "$out/r4_l1.log" 2>&1; cat "$out/r4_l1.log"
(cd /tmp && TMPDIR=/tmp timeout -k 10 60 rocprofv3 -L > "$out/r4_counters.txt" 2>&1) || true
timeout -k 10 900 bash profiles/run_pmc_kernels.sh r04_l1 pmcl1 conv3d_l1_kernel > "$out/r4_pmc_l1.log" 2>&1; tail -60 "$out/r4_pmc_l1.log"
