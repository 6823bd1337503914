#!/usr/bin/env bash
# HIP runtime knobs around hipGraphLaunch (torch's bundled libamdhip64 7.0): resident C2 step time and host issue time
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
log="$out/r4_graph_knobs.log"; : > "$log"
run() {
  echo "== $*" >> "$log"
  env "$@" timeout -k 10 200 python3 tools/h2d_probe.py 300 2>&1 | grep "resident batches" >> "$log" || echo "   (failed)" >> "$log"
}
run X=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 HIP_FORCE_DEV_KERNARG=1
run X=1
cat "$log"
