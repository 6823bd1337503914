#!/usr/bin/env bash
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$root"
for rep in 1 2 3; do
  echo "== prod"; python3 tools/kbench.py c4b 2>&1 | grep conv3d
  echo "== wres TU with SLP"; MMEEG_HIP_LIB="$root/multimodal_eeg_fmri_amd/csrc/build/alt_wres_slp.so" python3 tools/kbench.py c4b 2>&1 | grep conv3d
done
