#!/usr/bin/env bash
# build with -fno-slp-vectorize: full GPU suite, then step-time A/B against the two packed-fp32-free reference libs
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
b="$root/multimodal_eeg_fmri_amd/csrc/build"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$out/r3_t25.log" 2>&1
echo "rc=$?" >> "$out/r3_t25.log"
tail -4 "$out/r3_t25.log"
grep -q "rc=0" "$out/r3_t25.log" || exit 1
rm -f "$out/r3_noslp.log"
for rep in 1 2 3; do
  for v in prod nobnred alt_nopk_all; do
    unset MM_NO_BNRED MMEEG_HIP_LIB
    case $v in nobnred) export MM_NO_BNRED=1;; prod) ;; *) export MMEEG_HIP_LIB="$b/$v.so";; esac
    echo "== $v (rep $rep)" >> "$out/r3_noslp.log"
    python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> "$out/r3_noslp.log"
  done
done
cat "$out/r3_noslp.log"
python3 tools/kbench.py attn 2>&1 | grep attention
python3 tools/kbench.py lin 2>&1 | grep -v amdgpu | head -12
