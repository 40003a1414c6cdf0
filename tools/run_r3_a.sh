# round 3, first GPU call: the whole GPU suite, the bench as the driver runs it and at its default, and a kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3a_pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3a_bench20.log 2>&1 && tail -1 gpurun_out/r3a_bench20.log | cut -c1-400
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3a_bench64.log 2>&1 && tail -1 gpurun_out/r3a_bench64.log | cut -c1-400
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3a_prof -o b -- python bench.py --no-cpu-baseline > gpurun_out/r3a_prof.log 2>&1 || echo proffail
python tools/iter_timeline.py gpurun_out/r3a_prof/b_kernel_trace.csv > gpurun_out/r3a_timeline.txt; cat gpurun_out/r3a_timeline.txt
head -40 gpurun_out/r3a_prof/b_kernel_stats.csv | cut -c1-140 > gpurun_out/r3a_stats_head.txt
rm -f gpurun_out/r3a_prof/b_kernel_trace.csv.keep; ls -la gpurun_out/r3a_prof | head
