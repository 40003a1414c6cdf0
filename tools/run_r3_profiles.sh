# round 3: everything behind profiles/r03_* in one call (separate --pmc passes with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
# 1. bench kernel trace + stats (64 frames), timeline of an IEKF iteration
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/benchprof -o b -- python bench.py --no-cpu-baseline > gpurun_out/r03/benchprof.log 2>&1 || echo benchprof_fail
cp gpurun_out/r03/benchprof/b_kernel_stats.csv gpurun_out/r03/r03_bench_kernel_stats.csv
python tools/iter_timeline.py gpurun_out/r03/benchprof/b_kernel_trace.csv > gpurun_out/r03/r03_iter_timeline.txt
python tools/sor_by_series.py gpurun_out/r03/benchprof/b_kernel_trace.csv > gpurun_out/r03/r03_bench_sor_by_series.csv
python tools/kernel_gbps.py gpurun_out/r03/benchprof/b_kernel_trace.csv > gpurun_out/r03/r03_bench_kernel_gbps.csv 2> gpurun_out/r03/kernel_gbps.err || echo gbps_fail
# 2. the filter alone
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/alone -o b -- python tools/ekf_pmc.py 4 > gpurun_out/r03/alone.log 2>&1 || echo alone_fail
echo "--- the filter alone (tools/ekf_pmc.py 4, no flow beside it) ---" >> gpurun_out/r03/r03_iter_timeline.txt
python tools/iter_timeline.py gpurun_out/r03/alone/b_kernel_trace.csv >> gpurun_out/r03/r03_iter_timeline.txt
# 3. HBM traffic of the SOR kernel (the roofline block's `traffic`)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_fetch -o b -- python tools/brox_pmc.py > gpurun_out/r03/pmc_fetch.log 2>&1 || echo pmc_fetch_fail
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/pmc_write -o b -- python tools/brox_pmc.py > gpurun_out/r03/pmc_write.log 2>&1 || echo pmc_write_fail
python tools/sor_pmc_json.py gpurun_out/r03/pmc_fetch/b_counter_collection.csv gpurun_out/r03/pmc_write/b_counter_collection.csv > gpurun_out/r03/r03_sor_pmc.json || echo pmcjson_fail
python tools/series_breakdown.py gpurun_out/r03/pmc_fetch/b_kernel_trace.csv > gpurun_out/r03/r03_flow_series_kernels.csv 2>/dev/null || echo series_fail
# 4. SQ counters of the SOR kernel
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r03/sorsq_$tag -o b -- python tools/brox_pmc.py > gpurun_out/r03/sorsq_$tag.log 2>&1 || echo fail_$tag
done
python tools/sor_sq_summary.py gpurun_out/r03/sorsq_*/b_counter_collection.csv > gpurun_out/r03/r03_sor_sq_counters.csv || echo sqsum_fail
# 5. HBM traffic of the filter kernels
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/etraf_fetch -o b -- python tools/ekf_pmc.py 1 > gpurun_out/r03/etraf_fetch.log 2>&1 || echo etraf_fetch_fail
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r03/etraf_write -o b -- python tools/ekf_pmc.py 1 > gpurun_out/r03/etraf_write.log 2>&1 || echo etraf_write_fail
python tools/ekf_traffic_summary.py gpurun_out/r03 > gpurun_out/r03/r03_ekf_traffic.csv || echo etrafsum_fail
# 5b. SQ counters of the filter's kernels
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/r03/ekfsq_$tag -o b -- python tools/ekf_pmc.py 1 > gpurun_out/r03/ekfsq_$tag.log 2>&1 || echo fail_ekf_$tag
done
python tools/sor_sq_summary.py --prefix k_measure,k_render_iter,k_solve_prep,k_chol_flow,k_tvec,k_iter_result gpurun_out/r03/ekfsq_*/b_counter_collection.csv > gpurun_out/r03/r03_ekf_sq_counters.csv || echo ekfsqsum_fail
# 5c. the chain of the factorisation: one 32x32 diagonal block in a loop (variants), clock stamps of the chain's columns
timeout -k 10 120 ./build_exp/chol32_bench 2000 4 > gpurun_out/r03/r03_chol32_bench.txt 2>&1 || echo chol32_fail
timeout -k 10 200 python tools/stamp_chol.py > gpurun_out/r03/r03_chol_chain_stamps.txt 2>&1 || echo stampchol_fail
# 6. grid barrier against dependent launch
timeout -k 10 200 python tools/gbar_bench.py > gpurun_out/r03/r03_grid_barrier.txt 2>&1 || echo gbar_fail
# the raw traces are large: keep the summaries and the PMC collections
rm -f gpurun_out/r03/benchprof/b_kernel_trace.csv gpurun_out/r03/alone/b_kernel_trace.csv gpurun_out/r03/sorsq_*/b_kernel_trace.csv gpurun_out/r03/pmc_*/b_kernel_trace.csv gpurun_out/r03/etraf_*/b_kernel_trace.csv gpurun_out/r03/ekfsq_*/b_kernel_trace.csv
ls gpurun_out/r03; head -3 gpurun_out/r03/r03_sor_pmc.json; cat gpurun_out/r03/r03_iter_timeline.txt; cat gpurun_out/r03/r03_ekf_traffic.csv | head -12
