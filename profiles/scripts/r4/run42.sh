#!/usr/bin/env bash
# final-state kernel tables of the --config c4 and c5 legs
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd /tmp && export TMPDIR=/tmp
for c in c4 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r04_prof_final_$c" -- python3 "$root/bench.py" --config $c --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/r04_prof_final_$c.log" 2>&1
  f=$(ls "$out"/r04_prof_final_$c/*/*_kernel_trace.csv | head -1)
  python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r04_step_kernel_summary_$c.txt"; head -3 "$out/r04_step_kernel_summary_$c.txt" | cut -c1-140
done
