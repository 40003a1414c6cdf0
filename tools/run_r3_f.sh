cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -k "not golden" > gpurun_out/r3f_pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r3f_pytest.log
