# A/B of ONE library build against the product: bash tools/ab_one.sh build_exp/<lib>.so [pytest -k expression]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=$1; K=${2:-"measure or track or update"}
mkdir -p gpurun_out/ab_one; out=gpurun_out/ab_one/out.txt; : > $out
HYDRA_MI_SO=$GRAFT_REPO_ROOT/$V timeout -k 10 600 python -m pytest tests/test_ekf_gpu.py tests/test_configs_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "$K" > gpurun_out/ab_one/pytest.log 2>&1 || { tail -15 gpurun_out/ab_one/pytest.log; exit 1; }
tail -1 gpurun_out/ab_one/pytest.log >> $out
for v in "" "$V" "" "$V"; do
  if [ -n "$v" ]; then export HYDRA_MI_SO=$GRAFT_REPO_ROOT/$v; else unset HYDRA_MI_SO; fi
  timeout -k 10 200 python tools/filter_alone.py 2>/dev/null >> $out || exit 1
done
for v in "" "$V"; do
  if [ -n "$v" ]; then export HYDRA_MI_SO=$GRAFT_REPO_ROOT/$v; else unset HYDRA_MI_SO; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${v:-product} bench64 %.1f fps, steady %.1f' % (d['value'], d['steady_state']['value']))" >> $out || exit 1
done
cat $out
