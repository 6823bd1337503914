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
    # -fno-slp-vectorize: no compiler-formed v_pk_{fma,mul,add}_f32.  With them (op_sel-selected register halves, SGPR-pair
    # operands) the BatchNorm-reduce epilogue of igemm1d.hip gave run-to-run different sums in ~5 % of its workgroups
    # whenever it ran inside the two-stream training graph (never stand-alone); scalar fp32 code is bit-stable and the
    # step is not slower (profiles/r03_packed_fp32_ab.txt).  Hand-written v_pk_* in conv3d_wres's asm is unaffected.
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -Wall -Wno-unused-function "${extra[@]}" \
        -I"$here" -I"$here/../../include" -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "built $out"
