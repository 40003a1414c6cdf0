cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp kalman-hydra_amd/libhydra_mi.so /tmp/lib_orig.so
for e in 1 2 3; do
  cp build_exp/lib_exp$e.so kalman-hydra_amd/libhydra_mi.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp$e -o b -- python tools/measure_split.py 3 > gpurun_out/exp$e.log 2>&1 || echo "exp $e failed"
done
cp /tmp/lib_orig.so kalman-hydra_amd/libhydra_mi.so
echo rc=0
