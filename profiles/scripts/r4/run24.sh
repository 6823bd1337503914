#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 600 python3 -m pytest tests/test_trainer_gpu.py -m gpu -q -x -k "packed or bench_two" > "$out/r4_feeder_tests.log" 2>&1; rc=$?; tail -4 "$out/r4_feeder_tests.log"; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > "$out/r4_bench_feeder.log" 2>&1 || { tail -20 "$out/r4_bench_feeder.log"; exit 1; }
tail -1 "$out/r4_bench_feeder.log" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value", d["value"], "with_input_transfer", d["value_with_input_transfer"], d["ms_per_step"])'
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 > "$out/r4_bench_feeder200.log" 2>&1 || exit 1
tail -1 "$out/r4_bench_feeder200.log" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("200 steps: value", d["value"], "with_input_transfer", d["value_with_input_transfer"], d["ms_per_step"])'
