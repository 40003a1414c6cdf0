cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 ./build_exp/chol32_bench 2000 4 > gpurun_out/r3_chol32_bench.txt 2>&1; tail -2 gpurun_out/r3_chol32_bench.txt
timeout -k 10 600 python -m pytest tests/test_ekf_gpu.py tests/test_dense_gpu.py -m gpu -q -x > gpurun_out/r3r_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3r_pytest.log
timeout -k 10 100 python tools/chol_flow_check.py > gpurun_out/r3r_cholcheck.log 2>&1; tail -3 gpurun_out/r3r_cholcheck.log
for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3r_20_$rep.log 2>&1 || echo fail
  echo "(20): $(tail -1 gpurun_out/r3r_20_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["steady_state"]["value"], d["breakdown_ms_per_step"])')"
done
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3r_64.log 2>&1 || echo fail
echo "(64): $(tail -1 gpurun_out/r3r_64.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["steady_state"]["value"], d["breakdown_ms_per_step"])')"
