#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "pooled or head or trainer_modes or benchmarked_batch or c2_shaped or a3 or a4" > "$out/r4_heads_tests.log" 2>&1; rc=$?; tail -4 "$out/r4_heads_tests.log"; [ $rc -eq 0 ] || { grep -n "Error\|assert" "$out/r4_heads_tests.log" | head -20; exit $rc; }
timeout -k 10 300 python3 tools/h2d_probe.py 300 2>&1 | grep "resident batches"
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps 2>&1 | grep -v amdgpu | grep -v "^{"
