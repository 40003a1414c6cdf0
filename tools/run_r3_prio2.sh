cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 ./build_exp/chol32_bench 2000 4 > gpurun_out/r3_chol32_bench.txt 2>&1; cat gpurun_out/r3_chol32_bench.txt
timeout -k 10 120 ./build_exp/chol32_bench 2000 10 >> gpurun_out/r3_chol32_bench.txt 2>&1; tail -6 gpurun_out/r3_chol32_bench.txt
for low in 0 1; do
  if [ $low = 1 ]; then export HYDRA_MI_FLOW_LOWPRIO=1; fi
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cu-reserve 0 > gpurun_out/r3prio2_$low_$rep.log 2>&1 || echo fail
    echo "lowprio $low reserve 0 (20): $(tail -1 gpurun_out/r3prio2_$low_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["steady_state"]["value"])')"
  done
  timeout -k 10 200 python bench.py --no-cpu-baseline --cu-reserve 0 > gpurun_out/r3prio2_$low_64.log 2>&1 || echo fail
  echo "lowprio $low reserve 0 (64): $(tail -1 gpurun_out/r3prio2_$low_64.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["steady_state"]["value"])')"
done
