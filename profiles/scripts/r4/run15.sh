#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 300 python3 tools/h2d_probe.py 200 > "$out/r4_h2d3.log" 2>&1; echo "rc=$?" >> "$out/r4_h2d3.log"; grep -v "MB:" "$out/r4_h2d3.log" | tail -20
