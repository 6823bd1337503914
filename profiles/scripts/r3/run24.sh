#!/usr/bin/env bash
# packed-fp32 A/B: prod lib (packed ops on) / igemm1d.hip without / every TU without: bit-reproducibility test, then step time
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
b="$root/multimodal_eeg_fmri_amd/csrc/build"
rm -f "$out/r3_nopk.log"
for lib in alt_nopk_igemm alt_nopk_all; do
  echo "== bit-repro with $lib" >> "$out/r3_nopk.log"
  MMEEG_HIP_LIB="$b/$lib.so" timeout -k 10 600 python3 -m pytest tests/test_trainer_gpu.py -q -x -k "bit_reproducible" 2>&1 | tail -3 >> "$out/r3_nopk.log"
done
for rep in 1 2 3; do
  for v in nobnred prod alt_nopk_igemm alt_nopk_all; do
    unset MM_NO_BNRED MMEEG_HIP_LIB
    case $v in nobnred) export MM_NO_BNRED=1;; prod) ;; *) export MMEEG_HIP_LIB="$b/$v.so";; esac
    echo "== $v (rep $rep)" >> "$out/r3_nopk.log"
    python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> "$out/r3_nopk.log"
  done
done
cat "$out/r3_nopk.log"
