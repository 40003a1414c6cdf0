# GPU suite + bench (20 / 64 frames) + kernel trace timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r3b_pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3b_bench20.log 2>&1 && tail -1 gpurun_out/r3b_bench20.log | cut -c1-200
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3b_bench64.log 2>&1 && tail -1 gpurun_out/r3b_bench64.log | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3b_prof -o b -- python bench.py --no-cpu-baseline > gpurun_out/r3b_prof.log 2>&1 || echo proffail
python tools/iter_timeline.py gpurun_out/r3b_prof/b_kernel_trace.csv > gpurun_out/r3b_timeline.txt; cat gpurun_out/r3b_timeline.txt
