#!/usr/bin/env bash
# round 4, call 13: kernel table of the --config c5 and c4 steps (where do the extra legs spend their time?)
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
for cfg in c5 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$cfg" -- python3 "$root/bench.py" --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/prof_$cfg.log" 2>&1
  f=$(ls "$out"/prof_$cfg/*/*_kernel_trace.csv | head -1)
  python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r04_step_kernel_summary_$cfg.txt"
  head -24 "$out/r04_step_kernel_summary_$cfg.txt"
done
