#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "wgrad or bit_reproducible or benchmarked_batch or linear or a2 or a3 or fused_launches or batched" > "$out/r4_lw_tests.log" 2>&1; rc=$?; tail -3 "$out/r4_lw_tests.log"; [ $rc -eq 0 ] || { grep -n "Error\|assert" "$out/r4_lw_tests.log" | head -20; exit $rc; }
for rep in 1 2 3; do echo "rep $rep: $(timeout -k 10 200 python3 tools/h2d_probe.py 300 2>&1 | grep 'resident batches (mm' | cut -c80-118)"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r04_prof_lw" -- python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/r04_prof_lw.log" 2>&1
f=$(ls "$out"/r04_prof_lw/*/*_kernel_trace.csv | head -1)
python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r04_step_kernel_summary_lw.txt"; grep "total kernel\|wgrad" "$out/r04_step_kernel_summary_lw.txt" | cut -c1-150
