cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/w1
timeout -k 10 300 python tools/brox_time.py wide > gpurun_out/w1/wide.txt 2>&1 || { tail -5 gpurun_out/w1/wide.txt; exit 1; }
cat gpurun_out/w1/wide.txt | cut -c1-260
HYDRA_MI_BROX_TUNE=sor_wide=400 timeout -k 10 600 python -m pytest tests/test_brox_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -3
