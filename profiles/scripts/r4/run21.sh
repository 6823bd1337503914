#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
timeout -k 10 300 python3 tools/h2d_probe.py 300 > "$out/r4_headstart.log" 2>&1; echo "rc=$?" >> "$out/r4_headstart.log"; grep -v "MB:" "$out/r4_headstart.log" | head -12
