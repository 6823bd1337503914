cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "weight_resident or conv3d" > gpurun_out/r3_t1.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t1.log
timeout -k 10 300 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "volume_encoder" > gpurun_out/r3_t2.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t2.log
: > gpurun_out/r3_ab2.log
for rep in 1 2; do
  echo "== product (rep $rep)" >> gpurun_out/r3_ab2.log
  timeout -k 10 120 python tools/kbench.py c4b >> gpurun_out/r3_ab2.log 2>&1
  echo "== WRES_OPT=2 = round-2 schedule (rep $rep)" >> gpurun_out/r3_ab2.log
  MMEEG_HIP_LIB=$PWD/multimodal_eeg_fmri_amd/csrc/build/abl_0_o2.so timeout -k 10 120 python tools/kbench.py c4b >> gpurun_out/r3_ab2.log 2>&1
done
echo "== stamps product" >> gpurun_out/r3_ab2.log
MMEEG_HIP_LIB=$PWD/multimodal_eeg_fmri_amd/csrc/build/abl_s0_n.so timeout -k 10 120 python tools/kbench.py stamp >> gpurun_out/r3_ab2.log 2>&1
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r3_bench1.log 2>&1
tail -n 3 gpurun_out/r3_t1.log gpurun_out/r3_t2.log
grep -v amdgpu.ids gpurun_out/r3_ab2.log
