cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest2.log 2>&1; echo pytest_rc=$?
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_a.log 2>&1; echo bench_rc=$?
timeout -k 10 200 python tools/brox_time.py > gpurun_out/r2_brox_time_a.log 2>&1; echo broxtime_rc=$?
