#!/usr/bin/env bash
# round 4, call 5: which of the first voxel layer's changes pay?  ablation builds (L1_ABL bits: 1 per-lane rare branch,
# 2 fmaxf trees, 4 member-by-member dz fragments, 8 no halo prefetch), alternating, stand-alone kernel timings on ONE box
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
b="$root/multimodal_eeg_fmri_amd/csrc/build"
rm -f "$out/r4_l1_abl.log"
for rep in 1 2 3; do
  for n in 0 1 2 4 8 7 15; do
    echo "== rep $rep L1_ABL=$n" >> "$out/r4_l1_abl.log"
    MMEEG_HIP_LIB=$b/sabl_$n.so timeout -k 10 120 python3 tools/kbench.py l1 2>&1 | grep "forward\|backward\|Gram" | cut -c1-95 >> "$out/r4_l1_abl.log"
  done
done
cat "$out/r4_l1_abl.log"
