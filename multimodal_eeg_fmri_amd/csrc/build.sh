#!/usr/bin/env bash
# Builds libmmeeg_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../libmmeeg_hip.so"
mkdir -p "$here/build"
objs=()
pids=()
for src in "$here"/*.hip; do
  obj="$here/build/$(basename "${src%.hip}").o"
  objs+=("$obj")
  if [[ ! -f "$obj" ]] || ! [[ "$src" -ot "$obj" ]] || ! [[ "$here/common.h" -ot "$obj" ]] || ! [[ "$0" -ot "$obj" ]]; then
    extra=()
    # attention: MFMA results straight into VGPRs (the softmax is VALU work on every score: no v_accvgpr_read per
    # score).  Per file only: A/B in the training step, profiles/r03_attention_ab.txt (as a global flag it cost 25 %)
    [[ "$(basename "$src")" == attention.hip ]] && extra=(-mllvm -amdgpu-mfma-vgpr-form=1)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function "${extra[@]}" \
        -I"$here" -I"$here/../../include" -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "built $out"
