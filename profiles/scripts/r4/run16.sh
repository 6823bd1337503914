#!/usr/bin/env bash
# full GPU suite at the state after the capture-agreement rework and the per-head attention mask
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 1500 python3 -m pytest tests -m gpu -q -x > "$out/r4_full_tests2.log" 2>&1; rc=$?; tail -5 "$out/r4_full_tests2.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$out/r4_bench_final.log" 2>&1 && tail -1 "$out/r4_bench_final.log" | cut -c1-1500
