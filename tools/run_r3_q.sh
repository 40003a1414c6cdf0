cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "5 2" "5 3" "5 4" "4 2" "6 2" "8 2" "6 4"; do
  set -- $cfg
  export HYDRA_TUNE_measure_split=$1 HYDRA_TUNE_edge_split=$2
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3q_$1_$2 -o b -- python tools/ekf_pmc.py 3 > gpurun_out/r3q_$1_$2.log 2>&1 || echo fail
  echo "measure_split $1 edge_split $2: $(python tools/iter_timeline.py gpurun_out/r3q_$1_$2/b_kernel_trace.csv | grep -E 'k_measure_vertex|k_measure_edge|iterations' | tr '\n' ' ' | cut -c1-260)"
  rm -f gpurun_out/r3q_$1_$2/b_kernel_trace.csv
done
