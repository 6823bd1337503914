#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
s=$(date +%s); python3 bench.py > "$out/r3_bench_default.json" 2> "$out/r3_bench_default.err"; e=$(date +%s)
echo "default run: $((e-s)) s wall"
tail -1 "$out/r3_bench_default.json" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps','warmup','final_loss','value_with_input_transfer')})
print(d['top1_retrieval_acc'])
print('roofline', d['roofline']['frac'], 'c2', d['roofline_c2_standalone']['frac'], 'c4', d['roofline_c4']['frac'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
