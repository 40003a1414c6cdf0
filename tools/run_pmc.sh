cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o b -- python tools/brox_pmc.py > gpurun_out/pmc_fetch.log 2>&1 || echo failA
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o b -- python tools/brox_pmc.py > gpurun_out/pmc_write.log 2>&1 || echo failB
echo rc=0
