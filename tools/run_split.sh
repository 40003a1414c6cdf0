cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in 2 4 6; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/split$s -o b -- python tools/measure_split.py $s > gpurun_out/split$s.log 2>&1 || exit 1
done
echo rc=0
