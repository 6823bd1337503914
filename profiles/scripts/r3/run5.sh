cd $GRAFT_REPO_ROOT
B=$PWD/multimodal_eeg_fmri_amd/csrc/build
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv3d" > gpurun_out/r3_t4.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t4.log
timeout -k 10 600 python -m pytest tests/test_pipelines_gpu.py tests/test_trainer_gpu.py -x -q -m gpu > gpurun_out/r3_t5.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t5.log
: > gpurun_out/r3_ab3.log
for rep in 1 2; do
  echo "== product (rep $rep)" >> gpurun_out/r3_ab3.log
  timeout -k 10 120 python tools/kbench.py c4b >> gpurun_out/r3_ab3.log 2>&1
  echo "== round-2 kernel (rep $rep)" >> gpurun_out/r3_ab3.log
  MMEEG_HIP_LIB=$B/alt_base.so timeout -k 10 120 python tools/kbench.py c4b >> gpurun_out/r3_ab3.log 2>&1
done
echo "== stamps product" >> gpurun_out/r3_ab3.log
MMEEG_HIP_LIB=$B/abl_s0_n.so timeout -k 10 120 python tools/kbench.py stamp >> gpurun_out/r3_ab3.log 2>&1
timeout -k 10 400 python bench.py --steps 50 --warmup 10 > gpurun_out/r3_bench2.log 2>&1
tail -n 3 gpurun_out/r3_t4.log gpurun_out/r3_t5.log
grep -v amdgpu.ids gpurun_out/r3_ab3.log
tail -n 1 gpurun_out/r3_bench2.log
