cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/dp_rehearsal.py rccl1 > gpurun_out/r3_rccl1.log 2>&1; echo "rc=$?" >> gpurun_out/r3_rccl1.log
tail -n 2 gpurun_out/r3_rccl1.log | cut -c1-900
timeout -k 10 300 python tools/dp_rehearsal.py rccl1time > gpurun_out/r3_rccl1time.log 2>&1; echo "rc=$?" >> gpurun_out/r3_rccl1time.log
tail -n 2 gpurun_out/r3_rccl1time.log
timeout -k 10 300 python tools/dp_rehearsal.py 2 > gpurun_out/r3_gloo2.log 2>&1; echo "rc=$?" >> gpurun_out/r3_gloo2.log
tail -n 2 gpurun_out/r3_gloo2.log | cut -c1-600
timeout -k 10 600 python -m pytest tests/test_models_gpu.py tests/test_trainer_gpu.py -x -q -m gpu -k "volume_encoder_train or bench_two_ranks or reproducible" > gpurun_out/r3_t10.log 2>&1; echo "rc=$?" >> gpurun_out/r3_t10.log
tail -n 3 gpurun_out/r3_t10.log
