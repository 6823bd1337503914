cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "weight_resident or wgrad_at_training or accumulator" > gpurun_out/r3_t1.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t1.log
timeout -k 10 300 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "volume_encoder" > gpurun_out/r3_t2.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t2.log
: > gpurun_out/r3_ab.log
for rep in 1 2; do
for o in 0 1 2 3; do
  echo "== WRES_OPT=$o (rep $rep)" >> gpurun_out/r3_ab.log
  MMEEG_HIP_LIB=$PWD/multimodal_eeg_fmri_amd/csrc/build/abl_0_o$o.so timeout -k 10 120 python tools/kbench.py c4b >> gpurun_out/r3_ab.log 2>&1
done
done
for o in 0 1 2 3; do
  echo "== stamps WRES_OPT=$o" >> gpurun_out/r3_ab.log
  MMEEG_HIP_LIB=$PWD/multimodal_eeg_fmri_amd/csrc/build/abl_s0_o$o.so timeout -k 10 120 python tools/kbench.py stamp >> gpurun_out/r3_ab.log 2>&1
done
tail -n 3 gpurun_out/r3_t1.log gpurun_out/r3_t2.log
grep -v amdgpu.ids gpurun_out/r3_ab.log
