# where a frame goes now: host breakdown, iteration timeline and the gap between frames from the bench's kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/state
python tools/frame_breakdown.py > gpurun_out/state/frame_breakdown.txt 2>&1 || echo fb_fail
python tools/predict_breakdown.py > gpurun_out/state/predict_breakdown.txt 2>&1 || echo pb_fail
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/state/bp -o b -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/state/benchprof.log 2>&1 || echo benchprof_fail
python tools/iter_timeline.py gpurun_out/state/bp/b_kernel_trace.csv > gpurun_out/state/iter_timeline.txt
python tools/frame_gap_timeline.py gpurun_out/state/bp/b_kernel_trace.csv > gpurun_out/state/frame_gap.txt
rm -f gpurun_out/state/bp/b_kernel_trace.csv
cat gpurun_out/state/frame_breakdown.txt gpurun_out/state/predict_breakdown.txt gpurun_out/state/iter_timeline.txt gpurun_out/state/frame_gap.txt
