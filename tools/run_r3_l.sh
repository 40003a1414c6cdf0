cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3l_bench20_$i.log 2>&1 && tail -1 gpurun_out/r3l_bench20_$i.log | cut -c60-130
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3l_bench64_$i.log 2>&1 && tail -1 gpurun_out/r3l_bench64_$i.log | cut -c60-130
done
grep -o '"breakdown_ms_per_step[^}]*}' gpurun_out/r3l_bench20_1.log gpurun_out/r3l_bench64_1.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3l_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3l_pytest.log
