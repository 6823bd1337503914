cd $GRAFT_REPO_ROOT
bash profiles/run_round3.sh > gpurun_out/r03_collect.log 2>&1
tail -n 5 gpurun_out/r03_collect.log
cat gpurun_out/r03_step_phase_stamps.txt gpurun_out/r03_wres_graph_replayed.txt gpurun_out/r03_conv3d_family_standalone.txt gpurun_out/r03_attention_standalone.txt
cat gpurun_out/r03_wres_standalone_pmc3d.txt gpurun_out/r03_wres_standalone_pmc4.txt
tail -n 3 gpurun_out/pmcw_r03_c2.summary.txt gpurun_out/pmcw_r03_c4.summary.txt
head -n 30 gpurun_out/r03_step_kernel_summary.txt
