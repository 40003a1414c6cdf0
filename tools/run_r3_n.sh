cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3n_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3n_pytest.log
timeout -k 10 100 python __graft_entry__.py smoke 2>&1 | tail -2
for rep in 1 2; do
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3n_bench20_$rep.log 2>&1 && tail -1 gpurun_out/r3n_bench20_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["steady_state"]["value"], d["breakdown_ms_per_step"])'
done
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3n_bench64.log 2>&1 && tail -1 gpurun_out/r3n_bench64.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["steady_state"]["value"], d["breakdown_ms_per_step"])'
