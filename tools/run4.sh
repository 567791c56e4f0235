cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 300 python -m pytest tests/test_gpu_domain.py -q -m gpu -k "native_halo" > gpurun_out/r4/halo.log 2>&1; tail -3 gpurun_out/r4/halo.log
run() { env "$@" python3 tools/fused_ab.py $ARGS 2>&1 | grep median; }
ARGS="--relax 200"
for rep in 1 2; do
run X=base
run HTF_FUSED_LDS=1
run HTF_FUSED_LDS=1 HTF_FUSED_ROWS=1
done
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r4/pytest_gpu.log 2>&1; tail -3 gpurun_out/r4/pytest_gpu.log
grep -E "^FAILED|^ERROR" gpurun_out/r4/pytest_gpu.log | head -20
