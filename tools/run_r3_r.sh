cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3r_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3r_pytest.log
timeout -k 10 100 python __graft_entry__.py smoke 2>&1 | tail -1
timeout -k 10 300 python bench.py --no-cpu-baseline --videos-per-gpu 2 --steps 16 > gpurun_out/r3r_bench_v2.log 2>&1; echo "v2 rc=$?"; tail -1 gpurun_out/r3r_bench_v2.log | cut -c1-200
timeout -k 10 300 python bench.py --no-cpu-baseline --resident --steps 20 --warmup 5 > gpurun_out/r3r_bench_res.log 2>&1; echo "resident rc=$?"; tail -1 gpurun_out/r3r_bench_res.log | cut -c60-140
timeout -k 10 300 python bench.py --no-cpu-baseline --size 512 --h0 0.12 --steps 20 --warmup 5 > gpurun_out/r3r_bench512.log 2>&1; echo "512 rc=$?"; tail -1 gpurun_out/r3r_bench512.log | cut -c1-200
