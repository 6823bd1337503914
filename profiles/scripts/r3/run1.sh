set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "weight_resident or wgrad_at_training or accumulator" > gpurun_out/r3_t1.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t1.log
timeout -k 10 300 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "volume_encoder" > gpurun_out/r3_t2.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t2.log
for n in s0 s1 s2 s3 s64 s67; do
  echo "== WRES_ABL=$n" >> gpurun_out/r3_abl.log
  MMEEG_HIP_LIB=$PWD/multimodal_eeg_fmri_amd/csrc/build/abl_$n.so timeout -k 10 120 python tools/kbench.py stamp >> gpurun_out/r3_abl.log 2>&1
done
timeout -k 10 200 python tools/kbench.py c4b > gpurun_out/r3_c4b.log 2>&1
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r3_bench0.log 2>&1
tail -3 gpurun_out/r3_t1.log gpurun_out/r3_t2.log gpurun_out/r3_c4b.log
