#!/usr/bin/env bash
# when does the side branch of the step's graph start, against the runtime's batching knobs
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
log="$out/r4_side_start.log"; : > "$log"
run() {
  echo "== $*" >> "$log"
  env "$@" timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps 2>&1 | grep -v amdgpu | python3 -c '
import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        print("   ms_per_step", json.loads(l)["ms_per_step"])
    elif any(k in l for k in ("weights prepared", "fMRI fwd start", "fMRI fwd done", "EEG fwd done", "fMRI bwd start", "heads bwd done", "side stream done", "AdamW done")):
        print(l.rstrip())
' >> "$log"
}
run X=0
run DEBUG_CLR_MAX_BATCH_SIZE=1
run DEBUG_CLR_MAX_BATCH_SIZE=4
run DEBUG_CLR_MAX_BATCH_SIZE=1000
run DEBUG_CLR_BATCH_CPU_SYNC_SIZE=1
cat "$log"
