# the bench with two flow series in flight (the default) against CU reserve and series size
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_conc; out=gpurun_out/ab_conc/out.txt; : > $out
run() {
  echo "== $*" >> $out
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench20 %.1f fps, steady %.1f, %s' % (d['value'], d['steady_state']['value'], {k: round(v, 2) for k, v in d['breakdown_ms_per_step'].items()}))" >> $out || exit 1
  timeout -k 10 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench64 %.1f fps, steady %.1f, %s' % (d['value'], d['steady_state']['value'], {k: round(v, 2) for k, v in d['breakdown_ms_per_step'].items()}))" >> $out || exit 1
}
run
run --cu-reserve 0
run --cu-reserve 16
run --cu-reserve 48
run --cu-reserve 64
run --flow-batch 12
run --flow-batch 16
run --flow-batch 6
run
cat $out
