# GPU suite + bench (20 / 64 frames) + kernel trace timeline + EKF traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r3d}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/${T}_pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${T}_bench20.log 2>&1 && tail -1 gpurun_out/${T}_bench20.log | cut -c1-200
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench64.log 2>&1 && tail -1 gpurun_out/${T}_bench64.log | cut -c1-200
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_prof -o b -- python bench.py --no-cpu-baseline > gpurun_out/${T}_prof.log 2>&1 || echo proffail
python tools/iter_timeline.py gpurun_out/${T}_prof/b_kernel_trace.csv > gpurun_out/${T}_timeline.txt; cat gpurun_out/${T}_timeline.txt
# the filter alone (no flow beside it): kernel trace of a few frames
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${T}_alone -o b -- python tools/ekf_pmc.py 4 > gpurun_out/${T}_alone.log 2>&1 || echo alonefail
python tools/iter_timeline.py gpurun_out/${T}_alone/b_kernel_trace.csv > gpurun_out/${T}_timeline_alone.txt; cat gpurun_out/${T}_timeline_alone.txt
timeout -k 10 200 python tools/gbar_bench.py > gpurun_out/${T}_gbar.txt 2>&1; cat gpurun_out/${T}_gbar.txt
