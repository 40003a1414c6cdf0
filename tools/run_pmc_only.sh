# HBM traffic of the SOR kernel (the roofline block's `traffic`): two separate --pmc passes over tools/brox_pmc.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_fetch -o b -- python tools/brox_pmc.py > gpurun_out/r03/pmc_fetch.log 2>&1 || echo pmc_fetch_fail
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_write -o b -- python tools/brox_pmc.py > gpurun_out/r03/pmc_write.log 2>&1 || echo pmc_write_fail
python tools/sor_pmc_json.py gpurun_out/r03/pmc_fetch/b_counter_collection.csv gpurun_out/r03/pmc_write/b_counter_collection.csv > gpurun_out/r03/r03_sor_pmc.json || echo pmcjson_fail
rm -f gpurun_out/r03/pmc_*/b_kernel_trace.csv
head -12 gpurun_out/r03/r03_sor_pmc.json
