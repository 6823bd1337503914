#!/usr/bin/env bash
# round 4, call 14: config-#5 leg after the sample-group weight gradient, the per-(b, c) STFT workgroups and the float4 z-score
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "wgrad or stft or aX3 or config5 or zscore or power or a4 or lite or trainer_modes" > "$out/r4_t15.log" 2>&1; echo "rc=$?" >> "$out/r4_t15.log"
tail -6 "$out/r4_t15.log"
grep -q "rc=0" "$out/r4_t15.log" || exit 1
timeout -k 10 300 python3 bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline > "$out/r4_bench_c5c.log" 2>&1
python3 -c "
import json
d=json.loads([l for l in open('$out/r4_bench_c5c.log') if l.startswith('{')][-1]); print('c5', d['ms_per_step'], d['value'], d['value_with_input_transfer'], d['final_loss'])"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_c5c" -- python3 "$root/bench.py" --config c5 --steps 10 --warmup 3 --no-cpu-baseline --profile > "$out/prof_c5c.log" 2>&1
f=$(ls "$out"/prof_c5c/*/*_kernel_trace.csv | head -1)
python3 "$root/profiles/summarize.py" "$f" 15 > "$out/r04_step_kernel_summary_c5.txt"; head -14 "$out/r04_step_kernel_summary_c5.txt"
