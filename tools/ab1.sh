# round-2 A/B batch 1: occupancy (SGPR), workgroup size (L1 locality of the gathers), particle order
cd $GRAFT_REPO_ROOT
V=build_variants
run() { env "$@" python3 tools/fused_ab.py $ARGS 2>&1 | grep median; }
for ARGS in "--relax 200" "--relax 200 --fused 1"; do
for rep in 1 2; do
run X=base
run HTF_AMD_LIB=$V/libhtf_w8.so
run HTF_AMD_LIB=$V/libhtf_t1024.so HTF_FUSED_BLOCK=512
run HTF_AMD_LIB=$V/libhtf_t1024.so HTF_FUSED_BLOCK=1024
run HTF_AMD_LIB=$V/libhtf_t1024w8.so HTF_FUSED_BLOCK=512
run HTF_AMD_LIB=$V/libhtf_t1024w8.so HTF_FUSED_BLOCK=1024
run HTF_AMD_LIB=$V/libhtf_w8.so HTF_FUSED_ROWS=4
run HTF_AMD_LIB=$V/libhtf_t1024w8.so HTF_FUSED_BLOCK=1024 HTF_FUSED_ROWS=4
done
done
ARGS="--relax 200 --order sorted"; run X=base; run HTF_AMD_LIB=$V/libhtf_t1024w8.so HTF_FUSED_BLOCK=1024
ARGS="--relax 200 --order shuffled"; run X=base; run HTF_AMD_LIB=$V/libhtf_t1024w8.so HTF_FUSED_BLOCK=1024
ARGS="--relax 0"; run X=base; run HTF_AMD_LIB=$V/libhtf_t1024w8.so HTF_FUSED_BLOCK=1024
