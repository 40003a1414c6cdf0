# the bench with K idle extra streams in the process, against GPU_MAX_HW_QUEUES (tools/queue_cliff.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_cliff; out=gpurun_out/ab_cliff/out.txt; : > $out
for q in "" 3 2; do
 for k in 0 1 2 4; do
  export HM_EXTRA_STREAMS=$k
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  echo "== GPU_MAX_HW_QUEUES=$q extra idle streams $k" >> $out
  timeout -k 10 200 python tools/queue_cliff.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench20 %.1f fps, steady %.1f' % (d['value'], d['steady_state']['value']))" >> $out || exit 1
 done
done
cat $out
