cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_brox_gpu.py -m gpu -q -x -k "tuning" > gpurun_out/r3t_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r3t_pytest.log
timeout -k 10 200 python tools/frame_breakdown.py > gpurun_out/r3t_frame_breakdown.txt 2>&1; cat gpurun_out/r3t_frame_breakdown.txt | tail -16
run() {
  tag=$1; shift
  timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/r3t_$tag.log 2>&1 || echo fail
  echo "$tag: $(tail -1 gpurun_out/r3t_$tag.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"],3), round(d["steady_state"]["value"],1), {k: round(v,3) for k,v in d["breakdown_ms_per_step"].items()})')"
}
run base20a --steps 20 --warmup 5
run split20a --steps 20 --warmup 5 --split-start
run fs4_20 --steps 20 --warmup 5 --first-series 4
run fs5_20 --steps 20 --warmup 5 --first-series 5
run fs8_20 --steps 20 --warmup 5 --first-series 8
run base20b --steps 20 --warmup 5
run split20b --steps 20 --warmup 5 --split-start
run res48_20 --steps 20 --warmup 5 --cu-reserve 48
run res24_20 --steps 20 --warmup 5 --cu-reserve 24
run base64 
run split64 --split-start
