#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "config5 or stft or power" > "$out/r4_pside_tests.log" 2>&1; rc=$?; tail -4 "$out/r4_pside_tests.log"; [ $rc -eq 0 ] || { grep -n "Error\|assert" "$out/r4_pside_tests.log" | head -20; exit $rc; }
for v in 0 1 0 1; do
  MM_POWER_SIDE=$v timeout -k 10 300 python3 bench.py --config c5 --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 > "$out/r4_c5_pside$v.log" 2>&1 || { tail -20 "$out/r4_c5_pside$v.log"; exit 1; }
  echo "c5 MM_POWER_SIDE=$v: $(tail -1 "$out/r4_c5_pside$v.log" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
timeout -k 10 300 python3 bench.py --config c5 --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps 2>&1 | grep -v amdgpu | grep -v "^{"
