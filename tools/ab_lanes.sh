# A/B of the flow in two lanes (hm_brox_tune "lanes", HYDRA_MI_BROX_TUNE=lanes=2) in both benches
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_lanes; out=gpurun_out/ab_lanes/out.txt; : > $out
for v in "" "lanes=2" "" "lanes=2"; do
  export HYDRA_MI_BROX_TUNE=$v
  echo "== HYDRA_MI_BROX_TUNE=$v" >> $out
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench20 %.1f fps, steady %.1f' % (d['value'], d['steady_state']['value']))" >> $out || exit 1
  timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench64 %.1f fps, steady %.1f' % (d['value'], d['steady_state']['value']))" >> $out || exit 1
done
cat $out
