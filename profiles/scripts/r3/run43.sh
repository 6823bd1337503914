#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
python3 profiles/scripts/r3/kb_bnred.py 2>&1 | grep -v amdgpu
