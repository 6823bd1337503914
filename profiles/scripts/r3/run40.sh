#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps 2>&1 | grep -v "^{" | grep -v amdgpu > "$out/r3_stamps40.txt"
cat "$out/r3_stamps40.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/r3_p40" -- \
    python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/r3_p40.log" 2>&1
f=$(ls "$out"/r3_p40/*/*_kernel_trace.csv | head -1)
python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r3_p40_summary.txt"
rm -rf "$out/r3_p40"
grep "conv3d_l1\|total kernel" "$out/r3_p40_summary.txt"
