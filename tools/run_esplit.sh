cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in ${SPLITS:-1 2 3 4}; do
  HYDRA_MI_EDGE_SPLIT=$s timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/esplit$s -o b -- python tools/measure_split.py 5 $s > gpurun_out/esplit$s.log 2>&1 || exit 1
  grep -h "k_measure_vertex\|k_measure_edge" gpurun_out/esplit$s/b_kernel_stats.csv | cut -d, -f1-4 | sed "s/^/edge split $s: /"
done
echo rc=0
