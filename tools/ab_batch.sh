# pairs per flow series against the quantisation of k_sor's workgroups into rounds of the chip: bash tools/ab_batch.sh 8 7 6 9 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab_batch; out=gpurun_out/ab_batch/out.txt; : > $out
for b in "$@"; do
  for steps in "--steps 20 --warmup 5" ""; do
    timeout -k 10 200 python bench.py $steps --flow-batch $b --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['breakdown_ms_per_step']; print('flow-batch $b | $steps | %.1f fps, steady %.1f, flow wait %.3f, compute %.3f, iterations %.2f' % (d['value'], d['steady_state']['value'], b['brox_flow'], b['ekf_compute'], b['iekf_iterations']))" >> $out || exit 1
  done
done
cat $out
