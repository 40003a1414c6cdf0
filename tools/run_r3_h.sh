cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/stamp_render.py > gpurun_out/r3h_stamp.txt 2>&1; cat gpurun_out/r3h_stamp.txt | tail -15
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "dense_mesh or project_mask or config1 or config3" > gpurun_out/r3h_pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3h_pytest.log
