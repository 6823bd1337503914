#!/usr/bin/env bash
# v_rcp in GELU + bn grid 768 + prefetch: full suite, kernel timings, step A/B vs alt_prev (two commits back)
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
b="$root/multimodal_eeg_fmri_amd/csrc/build"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$out/r3_t31.log" 2>&1
echo "rc=$?" >> "$out/r3_t31.log"
tail -4 "$out/r3_t31.log"
grep -q "rc=0" "$out/r3_t31.log" || exit 1
python3 tools/kbench.py bn 2>&1 | grep bn_bwd
rm -f "$out/r3_ab31.log"
for rep in 1 2 3; do
  for v in prev prod; do
    unset MMEEG_HIP_LIB
    [ $v = prev ] && export MMEEG_HIP_LIB="$b/alt_prev2.so"
    echo "== $v (rep $rep)" >> "$out/r3_ab31.log"
    python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> "$out/r3_ab31.log"
  done
done
cat "$out/r3_ab31.log"
