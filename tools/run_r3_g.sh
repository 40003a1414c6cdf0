cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rows in 16 8; do
  export HYDRA_TUNE_render_rows=$rows
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3g_alone_$rows -o b -- python tools/ekf_pmc.py 4 > gpurun_out/r3g_alone_$rows.log 2>&1 || echo alonefail
  echo "render_rows $rows"; python tools/iter_timeline.py gpurun_out/r3g_alone_$rows/b_kernel_trace.csv | tee gpurun_out/r3g_timeline_alone_$rows.txt
done
unset HYDRA_TUNE_render_rows
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3g_bench20.log 2>&1 && tail -1 gpurun_out/r3g_bench20.log | cut -c1-200
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3g_bench64.log 2>&1 && tail -1 gpurun_out/r3g_bench64.log | cut -c1-200
