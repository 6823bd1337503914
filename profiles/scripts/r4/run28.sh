#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "batched_launches or wgrad or power or config5 or a4 or bit_reproducible" > "$out/r4_scatter_tests.log" 2>&1; rc=$?; tail -4 "$out/r4_scatter_tests.log"; [ $rc -eq 0 ] || { grep -n "Error\|assert" "$out/r4_scatter_tests.log" | head -20; exit $rc; }
timeout -k 10 300 python3 bench.py --config c5 --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 > "$out/r4_c5_scatter.log" 2>&1 || { tail -20 "$out/r4_c5_scatter.log"; exit 1; }
echo "c5: $(tail -1 "$out/r4_c5_scatter.log" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r04_prof_c5c" -- python3 "$root/bench.py" --config c5 --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/r04_prof_c5c.log" 2>&1
f=$(ls "$out"/r04_prof_c5c/*/*_kernel_trace.csv | head -1)
python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r04_step_kernel_summary_c5.txt"; head -22 "$out/r04_step_kernel_summary_c5.txt" | cut -c1-150
