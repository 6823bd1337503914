cd $GRAFT_REPO_ROOT
bash profiles/run_pmc_attn.sh r03b > gpurun_out/r3_pmc_attn_b.log 2>&1
grep -A17 "attn_fwd_kernel<true, true, false>\|attn_bwd_dkv_kernel<true, true, false>\|attn_bwd_dq_kernel<true, true, false>" gpurun_out/pmc_attn_r03b.summary.txt
