# A/B of hm_ctx_tune settings on the filter alone and the 64-frame bench: bash tools/ab_tune.sh "render_rows=24" "render_rows=32" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_tune; out=gpurun_out/ab_tune/out.txt; : > $out
for t in "" "$@" ""; do
  echo "== HYDRA_MI_TUNE=$t" >> $out
  HYDRA_MI_TUNE="$t" timeout -k 10 200 python tools/filter_alone.py 2>/dev/null >> $out || exit 1
  HYDRA_MI_TUNE="$t" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench20 %.1f fps, steady %.1f, iterations %.2f' % (d['value'], d['steady_state']['value'], d['breakdown_ms_per_step']['iekf_iterations']))" >> $out || exit 1
  HYDRA_MI_TUNE="$t" timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench64 %.1f fps, steady %.1f, iterations %.2f' % (d['value'], d['steady_state']['value'], d['breakdown_ms_per_step']['iekf_iterations']))" >> $out || exit 1
done
cat $out
