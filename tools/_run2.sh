cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
timeout 1200 python -m pytest tests/test_gpu_codegen.py tests/test_gpu_generic.py tests/test_gpu_tensorflow.py tests/test_gpu_training.py -x -q -m gpu > gpurun_out/r5c/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/r5c/pytest.log; grep -n "Error\|^E " gpurun_out/r5c/pytest.log | head -20
