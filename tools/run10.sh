cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { env "$@" python3 tools/fused_ab.py $ARGS 2>&1 | grep median; }
for ARGS in "--relax 200" "--relax 200 --fused 1"; do
for rep in 1 2; do
run X=base
run HTF_AMD_LIB=build_variants/libhtf_idxnt.so
run HTF_FUSED_TAILS=0
run HTF_FUSED_TAILS=0 HTF_AMD_LIB=build_variants/libhtf_idxnt.so
done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "liquid or fused_matches" 2>&1 | tail -3
