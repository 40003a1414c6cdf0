cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/flowprof -o b -- python tools/chol_flow_check.py > gpurun_out/flowprof.log 2>&1 || echo fail
python tools/kstat.py gpurun_out/flowprof/b_kernel_stats.csv k_chol k_flow k_assemble k_tvec k_ttt
