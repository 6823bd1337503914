#!/usr/bin/env bash
# round 4, call 11: a third stream for the handed-over weight-gradient work (MM_THIRD_STREAM=1) against the two-stream step
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
rm -f "$out/r4_third_stream_ab.log"
for rep in 1 2 3; do
  for v in "" "MM_THIRD_STREAM=1" "MM_THIRD_STREAM=1 MM_CONV_WGRADS_HANDED=2"; do
    echo "== rep $rep $v" >> "$out/r4_third_stream_ab.log"
    env $v timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 --profile 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4), round(d['value']), d['final_loss'])" >> "$out/r4_third_stream_ab.log"
  done
done
cat "$out/r4_third_stream_ab.log"
python3 -c "import __graft_entry__ as g; g.smoke()"
