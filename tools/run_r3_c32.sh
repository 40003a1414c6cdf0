cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 ./build_exp/chol32_bench 2000 4 > gpurun_out/r3_chol32_bench.txt 2>&1; cat gpurun_out/r3_chol32_bench.txt
