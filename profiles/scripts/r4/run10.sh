#!/usr/bin/env bash
# round 4, call 10: BatchNorm finalize folded into its consumers' prologues (six graph nodes fewer): tests, then the step
# with and without the fold (MM_NO_FIN_FOLD=1), alternating on one box
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > "$out/r4_t10.log" 2>&1; echo "rc=$?" >> "$out/r4_t10.log"
tail -12 "$out/r4_t10.log"
grep -q "rc=0" "$out/r4_t10.log" || exit 1
rm -f "$out/r4_fin_fold_ab.log"
for rep in 1 2 3; do
  for nf in "" 1; do
    echo "== rep $rep MM_NO_FIN_FOLD='$nf'" >> "$out/r4_fin_fold_ab.log"
    MM_NO_FIN_FOLD=$nf timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 --profile 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4), round(d['value']), d['final_loss'])" >> "$out/r4_fin_fold_ab.log"
  done
done
cat "$out/r4_fin_fold_ab.log"
timeout -k 10 200 python3 tools/h2d_probe.py 200 2>&1 | grep "resident batches\|ONE packed H2D" | head -3
