cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1 32" "0 32" "1 0" "1 16" "0 0"; do
  set -- $cfg
  export HYDRA_MI_EKF_PRIORITY=$1
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cu-reserve $2 > gpurun_out/r3prio_$1_$2_$rep.log 2>&1 || echo fail
    echo "prio $1 reserve $2 (20): $(tail -1 gpurun_out/r3prio_$1_$2_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("steady_state"))')"
  done
  timeout -k 10 200 python bench.py --no-cpu-baseline --cu-reserve $2 > gpurun_out/r3prio_$1_$2_64.log 2>&1 || echo fail
  echo "prio $1 reserve $2 (64): $(tail -1 gpurun_out/r3prio_$1_$2_64.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d.get("steady_state"))')"
done
