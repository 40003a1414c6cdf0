# A/B of the flow between the product's library and a reference build of it: bash tools/ab_flow.sh build_exp/<base>.so
# (Brox GPU tests with the product first; then a series of 8 x 1024^2 pairs and one pair alone, and both benches, for each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$1
mkdir -p gpurun_out/ab_flow; out=gpurun_out/ab_flow/out.txt; : > $out
timeout -k 10 900 python -m pytest tests/test_brox_gpu.py -m gpu -x -q > gpurun_out/ab_flow/pytest.log 2>&1 || { tail -15 gpurun_out/ab_flow/pytest.log; exit 1; }
tail -1 gpurun_out/ab_flow/pytest.log >> $out
for v in "" "$V" "" "$V"; do
  if [ -n "$v" ]; then export HYDRA_MI_SO=$GRAFT_REPO_ROOT/$v; else unset HYDRA_MI_SO; fi
  echo "== ${v:-product}" >> $out
  timeout -k 10 200 python tools/brox_time.py series 2>/dev/null >> $out || exit 1
done
for v in "" "$V"; do
  if [ -n "$v" ]; then export HYDRA_MI_SO=$GRAFT_REPO_ROOT/$v; else unset HYDRA_MI_SO; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-product} bench20 %.1f fps, steady %.1f' % (d['value'], d['steady_state']['value']))" >> $out || exit 1
  timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-product} bench64 %.1f fps, steady %.1f' % (d['value'], d['steady_state']['value']))" >> $out || exit 1
done
cat $out
