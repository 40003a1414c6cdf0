# cu_reserve sweep of the bench (64 and 20 frames)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 32 64 96 128; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --cu-reserve $r > gpurun_out/r3c_bench64_$r.log 2>&1 && echo "reserve $r: $(tail -1 gpurun_out/r3c_bench64_$r.log | cut -c60-120) $(tail -1 gpurun_out/r3c_bench64_$r.log | grep -o '"breakdown_ms_per_step[^}]*}')"
  timeout -k 10 200 python bench.py --no-cpu-baseline --cu-reserve $r --steps 20 --warmup 5 > gpurun_out/r3c_bench20_$r.log 2>&1 && echo "reserve $r (20): $(tail -1 gpurun_out/r3c_bench20_$r.log | cut -c60-120)"
done
