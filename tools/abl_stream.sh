#!/usr/bin/env bash
# ablation builds of the weight-streaming conv3d kernel: csrc/build/sabl_<N>.so for each STREAM_ABL value given
# (bits: csrc/conv3d_stream.hip).  Diagnostic only: these libraries compute wrong results by construction.
# usage: tools/abl_stream.sh 1 2 4 ...   then   MMEEG_HIP_LIB=.../csrc/build/sabl_N.so python tools/kbench.py stream
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
here="$root/multimodal_eeg_fmri_amd/csrc"
mkdir -p "$here/build"
for n in "$@"; do
  extra=""
  if [[ "$n" == s* ]]; then extra="-DSTREAM_STAMPS"; n="${n#s}"; fi       # s0, s1, ...: in-kernel cycle stamps (kbench.py sstamp)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DSTREAM_ABL=$n $extra -I"$here" -I"$here/../../include" \
      -c "$here/conv3d_stream.hip" -o "$here/build/conv3d_stream_sabl$n$extra.o" 2>/dev/null
  objs=()
  for o in "$here"/build/*.o; do
    case "$o" in *conv3d_stream.o|*_sabl*|*conv3d_wres_abl*) ;; *) objs+=("$o");; esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/build/sabl_$n$extra.so" "${objs[@]}" "$here/build/conv3d_stream_sabl$n$extra.o"
  echo "built sabl_$n$extra.so"
done
