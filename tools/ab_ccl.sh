cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ccl
timeout -k 10 300 python -m pytest tests/test_ekf_gpu.py -m gpu -x -q -k "pruning or project" > gpurun_out/ccl/pytest.log 2>&1 || { tail -20 gpurun_out/ccl/pytest.log; exit 1; }
tail -2 gpurun_out/ccl/pytest.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ccl/p -o b -- python tools/filter_alone.py > gpurun_out/ccl/prof.log 2>&1 || { tail -5 gpurun_out/ccl/prof.log; exit 1; }
grep -E "k_ccl|k_outline|k_project|k_ms_newton" gpurun_out/ccl/p/b_kernel_stats.csv | cut -c1-110
rm -f gpurun_out/ccl/p/b_kernel_trace.csv
