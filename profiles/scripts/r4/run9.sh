#!/usr/bin/env bash
# round 4, call 9: re-balance knobs after the fMRI branch got shorter: which EEG conv weight gradients go to the side stream
# (MM_CONV_WGRADS_HANDED) and how many CUs the 3-D weight-gradient kernel takes (MM_W3_CUS); one box, alternating
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
rm -f "$out/r4_balance_ab.log"
run() {
  echo "== $*" >> "$out/r4_balance_ab.log"
  env "$@" timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 --profile 2>/dev/null \
    | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4), round(d['value']))" >> "$out/r4_balance_ab.log"
}
for rep in 1 2 3; do
  run MM_CONV_WGRADS_HANDED=1
  run MM_CONV_WGRADS_HANDED=2
  run MM_CONV_WGRADS_HANDED=0
  run MM_CONV_WGRADS_HANDED=1 MM_W3_CUS=192
  run MM_CONV_WGRADS_HANDED=2 MM_W3_CUS=192
  run MM_CONV_WGRADS_HANDED=1 MM_W3_CUS=128
done
cat "$out/r4_balance_ab.log"
