# bench against GPU_MAX_HW_QUEUES (the hardware queues HIP multiplexes its streams on) and the flow's lanes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_queues; out=gpurun_out/ab_queues/out.txt; : > $out
for q in "" 1 2 3 4 5; do
 for v in "" "lanes=2"; do
  export HYDRA_MI_BROX_TUNE=$v
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  echo "== GPU_MAX_HW_QUEUES=$q HYDRA_MI_BROX_TUNE=$v" >> $out
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench20 %.1f fps, steady %.1f' % (d['value'], d['steady_state']['value']))" >> $out || exit 1
  timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench64 %.1f fps, steady %.1f' % (d['value'], d['steady_state']['value']))" >> $out || exit 1
 done
done
cat $out
