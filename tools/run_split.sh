cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in ${SPLITS:-2 4 6}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/split$s -o b -- python tools/measure_split.py $s > gpurun_out/split$s.log 2>&1 || exit 1
  python tools/kstat.py gpurun_out/split$s/b_kernel_stats.csv k_measure | sed "s/^/split $s: /"
done
echo rc=0
