cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "not config4" > gpurun_out/r3k_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3k_pytest.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3k_bench20.log 2>&1 && tail -1 gpurun_out/r3k_bench20.log | cut -c1-200
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3k_bench64.log 2>&1 && tail -1 gpurun_out/r3k_bench64.log | cut -c1-200
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3k_bench20b.log 2>&1 && tail -1 gpurun_out/r3k_bench20b.log | cut -c1-200
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3k_bench64b.log 2>&1 && tail -1 gpurun_out/r3k_bench64b.log | cut -c1-200
grep -o '"breakdown_ms_per_step[^}]*}' gpurun_out/r3k_bench20.log gpurun_out/r3k_bench64.log
python tools/determinism_check.py > gpurun_out/r3k_determinism.txt 2>&1; tail -5 gpurun_out/r3k_determinism.txt
