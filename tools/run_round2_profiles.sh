# the profiles behind profiles/r02_*: kernel trace + stats of the default bench command, HBM counters of the SOR
# kernel on the 8-pair series (FETCH_SIZE and WRITE_SIZE in separate passes), HBM counters of the filter kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/run_bench_profile.sh > gpurun_out/r2_benchprof.log 2>&1
bash tools/run_pmc.sh > gpurun_out/r2_pmc.log 2>&1
bash tools/run_ekf_traffic.sh > gpurun_out/r2_etraffic2.log 2>&1
bash tools/run_ekf_stats.sh > gpurun_out/r2_estats2.log 2>&1
echo done
