cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r5c/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/r5c/pytest.log; grep -n "^FAILED\|^ERROR" gpurun_out/r5c/pytest.log | head -20
