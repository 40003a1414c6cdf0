cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest3.log 2>&1; echo pytest_rc=$?
bash tools/run_ekf_stats.sh > gpurun_out/r2_estats.log 2>&1
bash tools/run_ekf_traffic.sh > gpurun_out/r2_etraffic.log 2>&1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_b.log 2>&1; echo bench_rc=$?
