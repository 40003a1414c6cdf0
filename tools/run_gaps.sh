cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gaps
for cfg in "1024 1" "1024 6" "512 1"; do
  set -- $cfg
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps/t_$1_$2 -o b -- python tools/brox_series.py $1 $2 2 > gpurun_out/gaps/log_$1_$2.txt 2>&1 || echo fail_$1_$2
  python tools/series_gaps.py gpurun_out/gaps/t_$1_$2/b_kernel_trace.csv > gpurun_out/gaps/gaps_$1_$2.txt
  tail -1 gpurun_out/gaps/log_$1_$2.txt
  rm -f gpurun_out/gaps/t_$1_$2/b_kernel_trace.csv
done
python tools/brox_series.py 1024 1 10; python tools/brox_series.py 1024 6 5; python tools/brox_series.py 512 1 10
