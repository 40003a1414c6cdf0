cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/stamp_render.py > gpurun_out/r3i_stamp.txt 2>&1; cat gpurun_out/r3i_stamp.txt | tail -12
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "not config4" > gpurun_out/r3i_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3i_pytest.log
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3i_alone -o b -- python tools/ekf_pmc.py 4 > gpurun_out/r3i_alone.log 2>&1 || echo alonefail
python tools/iter_timeline.py gpurun_out/r3i_alone/b_kernel_trace.csv | tee gpurun_out/r3i_timeline_alone.txt
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3i_bench20.log 2>&1 && tail -1 gpurun_out/r3i_bench20.log | cut -c1-200
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3i_bench64.log 2>&1 && tail -1 gpurun_out/r3i_bench64.log | cut -c1-200
