#!/usr/bin/env bash
# the fMRI-first / no-hand-over schedule for config #4's volumes: tests, then the c4 leg both ways, c2 for reference
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 900 python3 -m pytest tests/test_trainer_gpu.py -m gpu -q -x -k "fused_launches or config4_volume or rccl or bit_reproducible" > "$out/r4_sched_tests.log" 2>&1; rc=$?; tail -4 "$out/r4_sched_tests.log"; [ $rc -eq 0 ] || exit $rc
for v in 0 1; do
  MM_FMRI_LONGER=$v timeout -k 10 300 python3 bench.py --config c4 --steps 100 --warmup 20 > "$out/r4_c4_longer$v.log" 2>&1 || exit 1
  echo "c4 MM_FMRI_LONGER=$v: $(tail -1 "$out/r4_c4_longer$v.log" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done
timeout -k 10 300 python3 bench.py --config c4 --steps 100 --warmup 20 > "$out/r4_c4_auto.log" 2>&1 || exit 1
echo "c4 auto: $(tail -1 "$out/r4_c4_auto.log" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
