# HBM-side traffic of the filter kernels: FETCH_SIZE and WRITE_SIZE in separate passes over one frame
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/etraf_fetch -o b -- python tools/ekf_pmc.py 1 > gpurun_out/etraf_fetch.log 2>&1 || echo failA
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/etraf_write -o b -- python tools/ekf_pmc.py 1 > gpurun_out/etraf_write.log 2>&1 || echo failB
echo done
