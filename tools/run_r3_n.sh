cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3n_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3n_pytest.log
timeout -k 10 100 python __graft_entry__.py smoke 2>&1 | tail -2
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3n_bench20_full.log 2>&1 && tail -1 gpurun_out/r3n_bench20_full.log | cut -c1-300
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3n_bench64.log 2>&1 && tail -1 gpurun_out/r3n_bench64.log | cut -c60-130
timeout -k 10 300 python bench.py --workload flowbatch --steps 2 --warmup 1 --pairs-per-gpu 16 > gpurun_out/r3n_flowbatch.log 2>&1 && tail -1 gpurun_out/r3n_flowbatch.log | cut -c1-400
