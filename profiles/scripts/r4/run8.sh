#!/usr/bin/env bash
# round 4, call 8: one-stream A/B of the step (is the two-stream overlap still worth its contention?), then the round's
# evidence collector (profiles/run_round4.sh)
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
rm -f "$out/r4_one_stream_ab.log"
for rep in 1 2; do
  for one in "" 1; do
    echo "== rep $rep MM_ONE_STREAM='$one'" >> "$out/r4_one_stream_ab.log"
    MM_ONE_STREAM=$one timeout -k 10 200 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --fit-steps 0 --profile 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])" >> "$out/r4_one_stream_ab.log"
  done
done
cat "$out/r4_one_stream_ab.log"
timeout -k 10 1000 bash profiles/run_round4.sh > "$out/r4_collect.log" 2>&1; tail -8 "$out/r4_collect.log"
head -12 "$out/r04_step_kernel_summary.txt"; cat "$out"/r04_wres_standalone_*.txt "$out/r04_wres_sustained_events.txt" "$out/r04_step_phase_stamps.txt"
