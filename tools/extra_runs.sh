# config 3 of BASELINE.json (512^2 video, ~40-vertex mesh) and the self-launching --gpus N path rehearsed with two ranks on one GPU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/x1
timeout -k 10 300 python bench.py --size 512 --h0 0.12 --steps 20 --warmup 5 > gpurun_out/x1/bench512_cfg3_20.json 2> gpurun_out/x1/bench512_cfg3_20.err || exit 1
cut -c1-200 gpurun_out/x1/bench512_cfg3_20.json
timeout -k 10 300 python bench.py --size 512 --h0 0.12 --no-cpu-baseline > gpurun_out/x1/bench512_cfg3_64.json 2> gpurun_out/x1/bench512_cfg3_64.err || exit 1
cut -c1-200 gpurun_out/x1/bench512_cfg3_64.json
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/x1/bench_2ranks_gloo.json 2> gpurun_out/x1/bench_2ranks_gloo.err; echo "two ranks exit $?"
cut -c1-300 gpurun_out/x1/bench_2ranks_gloo.json; tail -3 gpurun_out/x1/bench_2ranks_gloo.err
