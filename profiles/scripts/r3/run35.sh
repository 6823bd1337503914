#!/usr/bin/env bash
# driver-style bench runs (default flags and --steps 20 --warmup 5), timed
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
s=$(date +%s); python3 bench.py --steps 20 --warmup 5 > "$out/r3_bench_driver.json" 2> "$out/r3_bench_driver.err"; e=$(date +%s)
echo "driver-style run: $((e-s)) s wall"; tail -1 "$out/r3_bench_driver.json" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step','steps','warmup','final_loss','value_with_input_transfer')})
print('roofline', {k:d['roofline'][k] for k in ('frac','avg_launch_ms','min_launch_ms','max_launch_ms','frac_raw_bracket')})
print('c2 standalone', d['roofline_c2_standalone']['frac'], d['roofline_c2_standalone']['avg_launch_ms'])
print('c4', d['roofline_c4']['frac'], d['roofline_c4']['avg_launch_ms'])
print('cpu', d['cpu_baseline'])
print('config', d['config'])
"
python3 -c "
import torch
from multimodal_eeg_fmri_amd.bridge_trainer import BridgeTrainer, synthetic_pairs
tr = BridgeTrainer(eeg_channels=64, dropout=0.3).train()
e, f = synthetic_pairs(32, 64, 1024, (32,32,32), seed=1)
tr.train_step(e, f); torch.cuda.synchronize()
g = tr._cap['graphs'][0]
print('capture mode:', tr.capture_mode)
" 2>&1 | grep -v amdgpu
