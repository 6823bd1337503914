#!/usr/bin/env bash
# fused BatchNorm-backward reduce behind the conv data gradients: tests, then in-step A/B (MM_NO_BNRED=1 = separate launches)
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"
mkdir -p "$out"
cd "$root"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "bn_backward_reduce or a3_ or erp or c2_shaped or bit_reproducible or frozen" > "$out/r3_t22.log" 2>&1
echo "rc=$?" >> "$out/r3_t22.log"
tail -5 "$out/r3_t22.log"
grep -q "rc=0" "$out/r3_t22.log" || exit 1
for rep in 1 2 3; do
  for v in 1 0; do
    if [ $v = 1 ]; then export MM_NO_BNRED=1; else unset MM_NO_BNRED; fi
    echo "== no_bnred=$v (rep $rep)" >> "$out/r3_bnred_ab.log"
    python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_loss'])" >> "$out/r3_bnred_ab.log"
  done
done
cat "$out/r3_bnred_ab.log"
