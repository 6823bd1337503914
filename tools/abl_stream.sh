#!/usr/bin/env bash
# ablation builds of one kernel file: csrc/build/sabl_<N>.so for each value N of the file's ablation macro
# (default: conv3d_stream.hip / STREAM_ABL; ABL_FILE=conv3d_wgrad tools/abl_stream.sh s0 = the 3-D weight-gradient kernel with cycle stamps).
# Diagnostic only: these libraries compute wrong results by construction.
# a leading 's' (s0, s1, ...) also defines STREAM_STAMPS (in-kernel cycle stamps, kbench.py sstamp).
# usage: tools/abl_stream.sh 1 2 4 ...   then   MMEEG_HIP_LIB=.../csrc/build/sabl_N.so python tools/kbench.py stream
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
here="$root/multimodal_eeg_fmri_amd/csrc"
file="${ABL_FILE:-conv3d_stream}"; macro="${ABL_MACRO:-STREAM_ABL}"
mkdir -p "$here/build"
for n in "$@"; do
  extra=""
  if [[ "$n" == s* ]]; then extra="-DSTREAM_STAMPS"; n="${n#s}"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -D$macro=$n $extra -I"$here" -I"$here/../../include" \
      -c "$here/$file.hip" -o "$here/build/${file}_sabl$n$extra.o" 2>/dev/null
  objs=()
  for o in "$here"/build/*.o; do
    case "$o" in *"/$file.o"|*_sabl*|*conv3d_wres_abl*) ;; *) objs+=("$o");; esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/build/sabl_$n$extra.so" "${objs[@]}" "$here/build/${file}_sabl$n$extra.o"
  echo "built sabl_$n$extra.so"
done
