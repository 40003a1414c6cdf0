# who gets the chip when the filter and the flow want it at once: the filter's stream priority and the flow stream's CU mask
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pr1
out=gpurun_out/pr1/out.txt; : > $out
run() {  # label, env, extra args
  for steps in "--steps 20 --warmup 5" ""; do
    env $2 timeout -k 10 200 python bench.py $steps --no-cpu-baseline $3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['breakdown_ms_per_step']
print('$1 | $steps | %.1f fps, steady %.1f, flow wait %.3f, compute %.3f' % (d['value'], d['steady_state']['value'], b['brox_flow'], b['ekf_compute']))" >> $out || exit 1
  done
}
run "as committed (filter high priority, flow leaves 32 CUs)" "X=1" ""
run "filter at normal priority" "HYDRA_MI_EKF_PRIORITY=0" ""
run "flow unmasked" "X=1" "--cu-reserve 0"
run "filter normal + flow unmasked" "HYDRA_MI_EKF_PRIORITY=0" "--cu-reserve 0"
run "flow leaves 64 CUs" "X=1" "--cu-reserve 64"
run "as committed again" "X=1" ""
cat $out
