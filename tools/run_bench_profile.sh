# the profile behind profiles/rNN_bench_*.csv: kernel trace + stats of the default bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/benchprof -o b -- python bench.py --no-cpu-baseline > gpurun_out/benchprof.log 2>&1 || echo fail
tail -1 gpurun_out/benchprof.log | cut -c1-160
python tools/sor_by_series.py gpurun_out/benchprof/b_kernel_trace.csv > gpurun_out/benchprof/sor_by_series.csv
python tools/kernel_gbps.py gpurun_out/benchprof/b_kernel_trace.csv > gpurun_out/benchprof/kernel_gbps.csv 2> gpurun_out/benchprof/kernel_gbps.err || echo gbpsfail
head -30 gpurun_out/benchprof/b_kernel_stats.csv | cut -c1-150
cat gpurun_out/benchprof/sor_by_series.csv
