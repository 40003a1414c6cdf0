cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/estats -o b -- python tools/ekf_pmc.py 3 > gpurun_out/estats.log 2>&1 || echo fail
cut -d, -f1-4,6,7 gpurun_out/estats/b_kernel_stats.csv | head -24
