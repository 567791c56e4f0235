cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r8
run() { env "$@" python3 tools/fused_ab.py $ARGS 2>&1 | grep median; }
for ARGS in "--relax 200" "--relax 200 --fused 1"; do
for rep in 1 2; do
run X=base
run HTF_FUSED_TAILS=2
run HTF_FUSED_TAILS=4
done
done
for P in 4 2; do
HTF_FUSED_TAILS=$P timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_standin.py tests/test_gpu_fuzz.py -q -m gpu -p no:cacheprovider > gpurun_out/r8/pytest_tails$P.log 2>&1; tail -3 gpurun_out/r8/pytest_tails$P.log; grep -E "^FAILED" gpurun_out/r8/pytest_tails$P.log | head -20
done
