#!/usr/bin/env bash
# 64- vs 128-row tiles in the k > 1 convolution weight-gradient kernel (MM_CW_MK), C2 step and the config-#5 leg
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x -k "wgrad or bit_reproducible or benchmarked_batch or config5" > "$out/r4_mk_tests.log" 2>&1; rc=$?; tail -3 "$out/r4_mk_tests.log"; [ $rc -eq 0 ] || { grep -n "Error\|assert" "$out/r4_mk_tests.log" | head -20; exit $rc; }
for rep in 1 2 3; do for v in 64 128; do
  echo "rep $rep MM_CW_MK=$v: $(MM_CW_MK=$v timeout -k 10 200 python3 tools/h2d_probe.py 300 2>&1 | grep 'resident batches (mm' | cut -c80-118)"
done; done
echo "c5 default: $(timeout -k 10 300 python3 bench.py --config c5 --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
