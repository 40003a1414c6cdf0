# two flow series in flight (default) against one, interleaved, three times each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_handles; out=gpurun_out/ab_handles/out.txt; : > $out
for i in 1 2 3; do
 for opt in "" "--one-flow-handle"; do
  echo "== ${opt:-two handles}" >> $out
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 $opt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench20 %.1f fps, steady %.1f, %s' % (d['value'], d['steady_state']['value'], {k: round(v, 2) for k, v in d['breakdown_ms_per_step'].items()}))" >> $out || exit 1
  timeout -k 10 200 python bench.py --no-cpu-baseline $opt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench64 %.1f fps, steady %.1f, %s' % (d['value'], d['steady_state']['value'], {k: round(v, 2) for k, v in d['breakdown_ms_per_step'].items()}))" >> $out || exit 1
 done
done
cat $out
