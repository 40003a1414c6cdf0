cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3u_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3u_pytest.log
timeout -k 10 200 python tools/frame_breakdown.py > gpurun_out/r3u_frame_breakdown.txt 2>&1; cat gpurun_out/r3u_frame_breakdown.txt | tail -14
run() {
  tag=$1; shift
  timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3u_$tag.log 2>&1 || echo fail
  echo "$tag: $(tail -1 gpurun_out/r3u_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3), round(d["steady_state"]["value"],1), {k: round(v,3) for k,v in d["breakdown_ms_per_step"].items()})')"
}
run base20a --steps 20 --warmup 5
run base20b --steps 20 --warmup 5
run base64
