#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "power or a4 or config5 or stft or trimodal or v4" > "$out/r4_pm3_tests.log" 2>&1; rc=$?; tail -4 "$out/r4_pm3_tests.log"; [ $rc -eq 0 ] || { grep -n "Error\|assert" "$out/r4_pm3_tests.log" | head -20; exit $rc; }
timeout -k 10 300 python3 bench.py --config c5 --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 > "$out/r4_c5_pm3.log" 2>&1 || { tail -20 "$out/r4_c5_pm3.log"; exit 1; }
echo "c5: $(tail -1 "$out/r4_c5_pm3.log" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
