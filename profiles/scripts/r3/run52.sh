#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
rm -f "$out/r3_fit3.log"
for off in 0 1 17 500 2000 9000; do
  python3 profiles/scripts/r3/fit_check3.py $off 2>/dev/null | tail -1 >> "$out/r3_fit3.log"
done
cat "$out/r3_fit3.log"
