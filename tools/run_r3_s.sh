cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x -s -k "true_mesh_states" > gpurun_out/r3s_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "true-state|passed|failed|Error" gpurun_out/r3s_pytest.log | cut -c1-400
