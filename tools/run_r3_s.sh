cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 ./build_exp/chol32_bench 2000 4 > gpurun_out/r3_chol32_bench.txt 2>&1; tail -2 gpurun_out/r3_chol32_bench.txt
timeout -k 10 200 python tools/stamp_chol.py > gpurun_out/r3s_stamp_chol.txt 2>&1; cat gpurun_out/r3s_stamp_chol.txt | tail -9
timeout -k 10 100 python tools/chol_flow_check.py > gpurun_out/r3r_cholcheck.log 2>&1; tail -3 gpurun_out/r3r_cholcheck.log
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3q_alone -o b -- python tools/ekf_pmc.py 3 > gpurun_out/r3q_alone.log 2>&1 || echo fail
python tools/iter_timeline.py gpurun_out/r3q_alone/b_kernel_trace.csv | grep -E "iterations|k_chol_flow" | cut -c1-200
rm -f gpurun_out/r3q_alone/b_kernel_trace.csv
