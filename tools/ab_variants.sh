# A/B of library builds (profiles/r04_ab_measure_occupancy_rsqrt.txt): the filter alone and both benches per build.
# The variants are built on the CPU box first, from kalman-hydra_amd/csrc, with the Makefile's flags plus one define each:
#   hipcc $FLAGS -DMEAS_OCC=5 -shared brox.hip ekf.hip predict.cpp -o ../../build_exp/libv_occ5.so      (k_measure_vertex at 96 VGPRs)
#   hipcc $FLAGS -DMEAS_OCC=6 ... -o ../../build_exp/libv_occ6.so                                       (80 VGPRs)
#   hipcc $FLAGS -DHM_RSQRT_ORDER2 ... -o ../../build_exp/libv_rsq2.so                                  (second-order reciprocal square root)
# (build_exp/ is not tracked; it travels to the GPU box with the tree)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab1
for v in "" build_exp/libv_occ5.so build_exp/libv_occ6.so build_exp/libv_rsq2.so; do
  if [ -n "$v" ]; then export HYDRA_MI_SO=$GRAFT_REPO_ROOT/$v; else unset HYDRA_MI_SO; fi
  echo "== variant ${v:-product}" >> gpurun_out/ab1/out.txt
  timeout -k 10 200 python tools/filter_alone.py >> gpurun_out/ab1/out.txt 2>&1 || exit 1
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench20 %.1f fps, steady %.1f, sor alone finest us %.1f' % (d['value'], d['steady_state']['value'], d['roofline']['alone']['finest_level']['avg_launch_us']))" >> gpurun_out/ab1/out.txt || exit 1
  timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench64 %.1f fps, steady %.1f' % (d['value'], d['steady_state']['value']))" >> gpurun_out/ab1/out.txt || exit 1
done
cat gpurun_out/ab1/out.txt
