cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3t_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3t_pytest.log
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3t_alone -o b -- python tools/ekf_pmc.py 4 > gpurun_out/r3t_alone.log 2>&1 || echo alonefail
python tools/iter_timeline.py gpurun_out/r3t_alone/b_kernel_trace.csv | tee gpurun_out/r3t_timeline_alone.txt
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3t_bench20_$i.log 2>&1 && tail -1 gpurun_out/r3t_bench20_$i.log | cut -c60-130
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r3t_bench64_$i.log 2>&1 && tail -1 gpurun_out/r3t_bench64_$i.log | cut -c60-130
done
