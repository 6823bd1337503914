#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 300 python3 bench.py --config c5 --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps 2>&1 | grep -v amdgpu | grep -v "^{" > "$out/r4_c5_stamps.log"; cat "$out/r4_c5_stamps.log"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r04_prof_c5d" -- python3 "$root/bench.py" --config c5 --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/r04_prof_c5d.log" 2>&1
f=$(ls "$out"/r04_prof_c5d/*/*_kernel_trace.csv | head -1)
python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r04_step_kernel_summary_c5.txt"
python3 "$root/tools/step_timeline.py" "$f" full > "$out/r04_step_timeline_c5.txt" 2>&1; head -120 "$out/r04_step_timeline_c5.txt" | cut -c1-150
