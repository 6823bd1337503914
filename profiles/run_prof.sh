#!/usr/bin/env bash
# usage (on the GPU box, from the repo root): profiles/run_prof.sh <tag> [bench args]
set -euo pipefail
tag="$1"; shift || true
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/gpurun_out/prof_$tag" -- \
    python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --profile "$@" > "$root/gpurun_out/prof_$tag.log" 2>&1
f=$(ls "$root"/gpurun_out/prof_$tag/*/*_kernel_trace.csv | head -1)
python3 "$root/profiles/summarize.py" "$f" 15 | tee "$root/gpurun_out/prof_$tag.summary.txt"
tail -1 "$root/gpurun_out/prof_$tag.log" | cut -c1-200
