cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5 6 7 8; do timeout -k 10 120 python profiles/scripts/r3/jitter.py 400 2>&1 | grep wall >> gpurun_out/r3_jitter.log; done
cat gpurun_out/r3_jitter.log
