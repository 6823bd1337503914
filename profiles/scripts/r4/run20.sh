#!/usr/bin/env bash
# two executables of the step launched alternately vs one (does the runtime serialise launches of one hipGraphExec?)
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
log="$out/r4_two_execs.log"; : > "$log"
for rep in 1 2; do for v in 1 2; do
  echo "== rep $rep MM_GRAPH_EXECS=$v" >> "$log"
  MM_GRAPH_EXECS=$v timeout -k 10 200 python3 tools/h2d_probe.py 300 2>&1 | grep "resident batches" >> "$log" || echo "   (failed)" >> "$log"
done; done
MM_GRAPH_EXECS=2 timeout -k 10 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --fit-steps 0 --stamps 2>&1 | grep -v "^{" | grep -v amdgpu >> "$log"
cat "$log"
