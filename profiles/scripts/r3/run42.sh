#!/usr/bin/env bash
# final state: full GPU suite, smoke, driver-style bench line
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$out/r3_final_tests.log" 2>&1
echo "rc=$?" >> "$out/r3_final_tests.log"
tail -3 "$out/r3_final_tests.log"
grep -q "rc=0" "$out/r3_final_tests.log" || exit 1
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2
python3 bench.py --steps 20 --warmup 5 > "$out/r3_final_bench.json" 2> "$out/r3_final_bench.err"
tail -1 "$out/r3_final_bench.json" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps','warmup','final_loss','value_with_input_transfer')})
print('roofline', {k:d['roofline'][k] for k in ('frac','avg_launch_ms','min_launch_ms','max_launch_ms','frac_raw_bracket')})
print('c2 standalone', d['roofline_c2_standalone']['frac'], 'c4', d['roofline_c4']['frac'])
"
