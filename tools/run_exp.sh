cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_ekf_gpu.py -m gpu -x -q > gpurun_out/g_test.log 2>&1; tail -2 gpurun_out/g_test.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp0 -o b -- python tools/measure_split.py 3 > gpurun_out/exp0.log 2>&1 || echo "exp 0 failed"
cp kalman-hydra_amd/libhydra_mi.so /tmp/lib_orig.so
for e in 1; do
  cp build_exp/lib_exp$e.so kalman-hydra_amd/libhydra_mi.so
  timeout -k 10 300 python -m pytest tests/test_ekf_gpu.py -m gpu -x -q > gpurun_out/g_test$e.log 2>&1; tail -2 gpurun_out/g_test$e.log
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/exp$e -o b -- python tools/measure_split.py 3 > gpurun_out/exp$e.log 2>&1 || echo "exp $e failed"
done
cp /tmp/lib_orig.so kalman-hydra_amd/libhydra_mi.so
echo rc=0
