cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fin
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/fin/smoke.txt 2>&1 || { cat gpurun_out/fin/smoke.txt; exit 1; }
tail -1 gpurun_out/fin/smoke.txt
TAG=fin bash tools/gpu_session.sh tests driver
