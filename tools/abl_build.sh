#!/usr/bin/env bash
# ablation builds of the weight-resident conv3d kernel: csrc/build/abl_<N>.so for each WRES_ABL value given
# (bits: tools/gen_wres_asm.py).  Diagnostic only: these libraries compute wrong results by construction.
# a leading 's' (s0, s1, ...) also compiles the in-kernel cycle stamps in (kbench.py stamp).
# usage: tools/abl_build.sh 1 2 4 s0 ...   then   MMEEG_HIP_LIB=.../csrc/build/abl_N.so python tools/kbench.py c4b
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
here="$root/multimodal_eeg_fmri_amd/csrc"
mkdir -p "$here/build"
for n in "$@"; do
  extra=""
  if [[ "$n" == s* ]]; then extra="-DWRES_STAMPS"; n="${n#s}"; tag="s$n"; else tag="$n"; fi
  tag="$tag${WRES_TAG:-}"
  WRES_ABL=$n python3 "$root/tools/gen_wres_asm.py" "$here/build/wres_abl$n.inc" > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DWRES_ASM_INC="\"build/wres_abl$n.inc\"" $extra -I"$here" -I"$here/../../include" \
      -c "$here/conv3d_wres.hip" -o "$here/build/conv3d_wres_abl$tag.o" 2>/dev/null
  objs=()
  for o in "$here"/build/*.o; do
    case "$o" in *conv3d_wres.o|*conv3d_wres_abl*|*_sabl*|*/alt_*) ;; *) objs+=("$o");; esac
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/build/abl_$tag.so" "${objs[@]}" "$here/build/conv3d_wres_abl$tag.o"
  echo "built abl_$tag.so"
done
