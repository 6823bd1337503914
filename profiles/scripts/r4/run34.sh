#!/usr/bin/env bash
# config #5: slots of the merged convolution's weight gradient (more workgroups per tile against more slot images to sum)
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
for rep in 1 2; do for v in 384 640 900 1200; do
  MM_WG_SLOT_TARGET=$v timeout -k 10 300 python3 bench.py --config c5 --steps 100 --warmup 20 --no-cpu-baseline --fit-steps 0 > "$out/r4_c5_slots_$v.log" 2>&1 || { tail -20 "$out/r4_c5_slots_$v.log"; exit 1; }
  echo "rep $rep MM_WG_SLOT_TARGET=$v: $(tail -1 "$out/r4_c5_slots_$v.log" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
done; done
