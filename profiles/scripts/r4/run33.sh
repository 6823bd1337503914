#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r04_prof_c5e" -- python3 "$root/bench.py" --config c5 --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/r04_prof_c5e.log" 2>&1
f=$(ls "$out"/r04_prof_c5e/*/*_kernel_trace.csv | head -1)
python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r04_step_kernel_summary_c5.txt"; head -30 "$out/r04_step_kernel_summary_c5.txt" | cut -c1-150
