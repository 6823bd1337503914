#!/usr/bin/env bash
# step kernel table (rocprofv3 --kernel-trace) of the current build
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r3_p27" -- \
    python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/r3_p27.log" 2>&1
f=$(ls "$out"/r3_p27/*/*_kernel_trace.csv | head -1)
python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r3_p27_summary.txt"
python3 "$root/tools/step_timeline.py" "$f" full > "$out/r3_p27_timeline.txt" 2>&1
rm -rf "$out/r3_p27"
head -60 "$out/r3_p27_summary.txt"
