#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "wgrad or bit_reproducible or benchmarked_batch or a3 or config5" > "$out/r4_mk_tests.log" 2>&1; rc=$?; tail -4 "$out/r4_mk_tests.log"; [ $rc -eq 0 ] || { grep -n "Error\|assert" "$out/r4_mk_tests.log" | head -20; exit $rc; }
timeout -k 10 300 python3 tools/kbench.py conv1 2>&1 | grep -i "wgrad"
timeout -k 10 300 python3 tools/h2d_probe.py 300 2>&1 | grep "resident batches"
timeout -k 10 300 python3 bench.py --config c5 --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("c5", d["ms_per_step"], d["value"])'
