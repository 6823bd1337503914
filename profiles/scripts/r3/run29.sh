#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
for g in 1024 768 512 384 256; do echo "grid cap $g"; MM_BN_GRID=$g python3 tools/kbench.py bn 2>&1 | grep bn_bwd; done
