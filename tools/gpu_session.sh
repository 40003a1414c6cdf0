#!/bin/bash
# One gpurun call of round 4's sessions: `bash tools/gpu_session.sh <step> [...]` runs the named steps in order and stops
# at the first that fails (no GPU step is started behind a failed one).  Output under gpurun_out/<tag>/.
#   tests      pytest -m gpu
#   bench      bench.py --steps 20 --warmup 5 (what the driver runs) and the default 64 frames
#   prof       rocprofv3 --kernel-trace --stats around the 20-frame bench (must exit 0), iteration / frame-gap timelines
#   stamps     tools/stamp_chol.py with the instrumented build (build_exp/libhydra_mi_stamp.so)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${TAG:-s}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for step in "$@"; do
    echo "== $step" | tee -a "$OUT/steps.txt"
    case $step in
    tests)
        timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; rc=$?
        tail -5 "$OUT/pytest.log" ;;
    bench)
        timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench20.json" 2> "$OUT/bench20.err"; rc=$?
        cut -c1-400 "$OUT/bench20.json"
        if [ $rc -eq 0 ]; then
            timeout -k 10 300 python bench.py --no-cpu-baseline > "$OUT/bench64.json" 2> "$OUT/bench64.err"; rc=$?
            cut -c1-300 "$OUT/bench64.json"
        fi ;;
    prof)
        timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bp" -o b -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/benchprof.log" 2>&1; rc=$?
        echo "rocprofv3 bench exit code $rc" | tee "$OUT/benchprof_rc.txt"
        python tools/iter_timeline.py "$OUT/bp/b_kernel_trace.csv" > "$OUT/iter_timeline.txt" 2>&1
        python tools/frame_gap_timeline.py "$OUT/bp/b_kernel_trace.csv" > "$OUT/frame_gap.txt" 2>&1
        python tools/sor_by_series.py "$OUT/bp/b_kernel_trace.csv" > "$OUT/sor_by_series.csv" 2>&1
        rm -f "$OUT/bp/b_kernel_trace.csv"
        cat "$OUT/iter_timeline.txt"; head -40 "$OUT/frame_gap.txt" ;;
    stamps)
        timeout -k 10 200 python tools/stamp_chol.py > "$OUT/chol_chain_stamps.txt" 2>&1; rc=$?
        cat "$OUT/chol_chain_stamps.txt" ;;
    *)
        echo "unknown step $step"; rc=1 ;;
    esac
    echo "== $step exit $rc" | tee -a "$OUT/steps.txt"
    [ $rc -eq 0 ] || exit $rc
done
