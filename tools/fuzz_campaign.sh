# Fuzz / parity campaign with each kernel FORM forced (the defaults pick a form by batch size, so the small random systems of
# tests/test_gpu_fuzz.py never reach the merged-tail kernels on their own): HTF_FUSED_TAILS = 2 | 3 | 4 over 60 random systems +
# the parity suite, HTF_FUSED2_ROWS = 4 | 0 over the C4 / EDS / RDF tests.  Expected: everything passes except, under
# HTF_FUSED_TAILS=2, five assertions of BIT-identity between forms (test_rows_per_wave_variants_are_bit_identical[1,2,8],
# test_fused_matches_two_kernel_path_and_oracle[128-*]): a call with check_count takes the plain two-row form, whose second
# row sums its tail in a different order (differences <= 1 ulp of the largest term, 6e-5 on |F| ~ 1e3; measured round 3).
# Round 4: the forms and their switches exist in the variants build only (tools/build_variant.sh ab -DHTF_AB_VARIANTS, made by
# __graft_entry__.build()): the campaign loads it through HTF_AMD_LIB.
#   gpurun -- 'bash tools/fuzz_campaign.sh'   -> gpurun_out/fuzz_campaign.txt
cd $GRAFT_REPO_ROOT
export HTF_AMD_LIB=$GRAFT_REPO_ROOT/build_variants/libhtf_ab.so
[ -f $HTF_AMD_LIB ] || { echo "no variants build: tools/build_variant.sh ab -DHTF_AB_VARIANTS"; exit 1; }
O=gpurun_out/fuzz_campaign.txt; : > $O
for t in 2 3 4; do
  echo "== HTF_FUSED_TAILS=$t" >> $O
  HTF_FUSED_TAILS=$t HTF_FUZZ_SEEDS=60 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -15 >> $O
done
for r in 4 0; do
  echo "== HTF_FUSED2_ROWS=$r" >> $O
  HTF_FUSED2_ROWS=$r timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tensorflow.py -q -m gpu -k "eds or c4 or forces2 or rdf or sweep or two" 2>&1 | tail -8 >> $O
done
cat $O
