# two flow series in flight against one, at 20, 64 and 120 frames, interleaved twice
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_handles; out=gpurun_out/ab_handles/long.txt; : > $out
for i in 1 2; do
 for opt in "" "--one-flow-handle"; do
  echo "== ${opt:-two handles}" >> $out
  for st in "20 5" "64 2" "120 5"; do
   set -- $st
   timeout -k 10 200 python bench.py --no-cpu-baseline --steps $1 --warmup $2 $opt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%d frames: %.1f fps, steady %.1f, %s' % (d['steps'], d['value'], d['steady_state']['value'], {k: round(v, 2) for k, v in d['breakdown_ms_per_step'].items()}))" >> $out || exit 1
  done
 done
done
cat $out
