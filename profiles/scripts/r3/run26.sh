#!/usr/bin/env bash
# tapsum rewrite + adamw tail fold: GPU suite, kernel timing, step A/B against the previous commit's library
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
b="$root/multimodal_eeg_fmri_amd/csrc/build"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$out/r3_t26.log" 2>&1
echo "rc=$?" >> "$out/r3_t26.log"
tail -4 "$out/r3_t26.log"
grep -q "rc=0" "$out/r3_t26.log" || exit 1
rm -f "$out/r3_ab26.log"
for rep in 1 2 3; do
  for v in prev prod; do
    unset MMEEG_HIP_LIB
    [ $v = prev ] && export MMEEG_HIP_LIB="$b/alt_prev.so"
    echo "== $v (rep $rep)" >> "$out/r3_ab26.log"
    python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> "$out/r3_ab26.log"
  done
done
cat "$out/r3_ab26.log"
