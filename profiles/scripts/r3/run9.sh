cd $GRAFT_REPO_ROOT
bash profiles/run_pmc_attn.sh r03 > gpurun_out/r3_pmc_attn.log 2>&1
tail -n 70 gpurun_out/r3_pmc_attn.log
