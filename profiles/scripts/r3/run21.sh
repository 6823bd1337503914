#!/usr/bin/env bash
# one-stream profile of the step: per-kernel durations without the other stream's kernels on the chip
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"
mkdir -p "$out"
cd "$root"
MM_ONE_STREAM=1 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps 2>&1 | grep -v amdgpu > "$out/r3_one_stream_stamps.txt"
python3 tools/kbench.py bn 2>&1 | grep bn_bwd > "$out/r3_bn.txt"
cd /tmp && export TMPDIR=/tmp
export MM_ONE_STREAM=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r3_one" -- \
    python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/r3_one.log" 2>&1
f=$(ls "$out"/r3_one/*/*_kernel_trace.csv | head -1)
python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r3_one_stream_kernel_summary.txt"
python3 "$root/tools/step_timeline.py" "$f" full > "$out/r3_one_stream_timeline.txt" 2>&1
rm -rf "$out/r3_one"
echo done
