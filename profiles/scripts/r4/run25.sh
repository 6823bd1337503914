#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "adamw or AdamW or bit_reproducible or graph_step_at_the_benchmarked or checkpoint or trainer_modes or rccl or two_rank" > "$out/r4_adamw_tests.log" 2>&1; rc=$?; tail -4 "$out/r4_adamw_tests.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/h2d_probe.py 300 2>&1 | grep "resident batches"
