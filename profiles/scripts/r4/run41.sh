#!/usr/bin/env bash
# first voxel layer backward at three workgroups per CU (168 registers, 20 B of scratch) against two (181 registers)
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out"; mkdir -p "$out"; cd "$root"
abl="$root/multimodal_eeg_fmri_amd/csrc/build/sabl_3.so"
for rep in 1 2; do
  echo "rep $rep two per CU  : $(timeout -k 10 200 python3 tools/kbench.py l1 2>&1 | grep 'backward')"
  echo "rep $rep three per CU: $(MMEEG_HIP_LIB=$abl timeout -k 10 200 python3 tools/kbench.py l1 2>&1 | grep 'backward')"
done
for rep in 1 2 3; do
  echo "rep $rep step, two  : $(timeout -k 10 200 python3 tools/h2d_probe.py 300 2>&1 | grep 'resident batches (mm' | cut -c80-118)"
  echo "rep $rep step, three: $(MMEEG_HIP_LIB=$abl timeout -k 10 200 python3 tools/h2d_probe.py 300 2>&1 | grep 'resident batches (mm' | cut -c80-118)"
done
