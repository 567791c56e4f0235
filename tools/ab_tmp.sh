cd $GRAFT_REPO_ROOT
O=gpurun_out/mlp_pad.txt; : > $O
for lib in hoomd_tf_amd/libhtf_amd.so build_variants/libhtf_mpad64.so build_variants/libhtf_mpad128.so hoomd_tf_amd/libhtf_amd.so; do
  echo $lib >> $O
  HTF_AMD_LIB=$lib timeout 200 python tools/mlp_ab.py split16 fp32 2>&1 | tail -3 >> $O
done
cat $O
