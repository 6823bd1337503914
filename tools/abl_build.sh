#!/usr/bin/env bash
# ablation builds of the W-resident conv3d kernel: csrc/build/abl_<N>.so for each WR_ABL value given
set -euo pipefail
here="$(cd "$(dirname "$0")/../multimodal_eeg_fmri_amd/csrc" && pwd)"
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DWR_ABL=$n -I"$here" -I"$here/../../include" \
      -c "$here/conv3d.hip" -o "$here/build/conv3d_abl$n.o" 2>/dev/null
  objs=()
  for o in "$here"/build/*.o; do
    case "$o" in *conv3d.o|*conv3d_abl*) ;; *) objs+=("$o");; esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/build/abl_$n.so" "${objs[@]}" "$here/build/conv3d_abl$n.o"
  echo "built abl_$n.so"
done
