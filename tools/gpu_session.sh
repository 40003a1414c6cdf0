#!/bin/bash
# One gpurun call of round 4's sessions: `bash tools/gpu_session.sh <step> [...]` runs the named steps in order and stops
# at the first that fails (no GPU step is started behind a failed one).  Output under gpurun_out/<tag>/.
#   tests      pytest -m gpu
#   bench      bench.py --steps 20 --warmup 5 (what the driver runs) and the default 64 frames
#   prof       rocprofv3 --kernel-trace --stats around the 20-frame bench (must exit 0), iteration / frame-gap timelines
#   b512       the same two runs at 512^2 (BASELINE config 3), the first with the CPU baseline
#   driver     the driver's command, CPU baseline included
#   pmc        the two rocprofv3 --pmc passes of tools/brox_pmc.py and tools/sor_pmc_json.py (profiles/rNN_sor_pmc.json)
#   series     kernel trace of a flow series of 8 pairs alone, by kernel
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
    b512)
        timeout -k 10 300 python bench.py --size 512 --steps 20 --warmup 5 > "$OUT/bench512_20.json" 2> "$OUT/bench512_20.err"; rc=$?
        cut -c1-300 "$OUT/bench512_20.json"
        if [ $rc -eq 0 ]; then
            timeout -k 10 300 python bench.py --size 512 --no-cpu-baseline > "$OUT/bench512_64.json" 2> "$OUT/bench512_64.err"; rc=$?
            cut -c1-300 "$OUT/bench512_64.json"
        fi ;;
    driver)
        timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"; rc=$?
        cut -c1-300 "$OUT/bench_driver.json" ;;
    pmc)
        timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o b -- python tools/brox_pmc.py > "$OUT/pmc_fetch.log" 2>&1; rc=$?
        if [ $rc -eq 0 ]; then
            timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o b -- python tools/brox_pmc.py > "$OUT/pmc_write.log" 2>&1; rc=$?
        fi
        if [ $rc -eq 0 ]; then
            python tools/sor_pmc_json.py "$OUT/pmc_fetch/b_counter_collection.csv" "$OUT/pmc_write/b_counter_collection.csv" > "$OUT/sor_pmc.json"; rc=$?
            head -c 600 "$OUT/sor_pmc.json"
            rm -f "$OUT"/pmc_*/b_kernel_trace.csv
        fi ;;
    series)
        timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/series" -o b -- python tools/brox_pmc.py > "$OUT/series.log" 2>&1; rc=$?
        python tools/series_breakdown.py "$OUT/series/b_kernel_trace.csv" > "$OUT/flow_series_kernels.csv" 2>&1
        rm -f "$OUT/series/b_kernel_trace.csv"
        cat "$OUT/flow_series_kernels.csv" ;;
    stamps)
        timeout -k 10 200 python tools/stamp_chol.py > "$OUT/chol_chain_stamps.txt" 2>&1; rc=$?
        cat "$OUT/chol_chain_stamps.txt" ;;
    *)
        echo "unknown step $step"; rc=1 ;;
    esac
    echo "== $step exit $rc" | tee -a "$OUT/steps.txt"
    [ $rc -eq 0 ] || exit $rc
done
