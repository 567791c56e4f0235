cd $GRAFT_REPO_ROOT
O=gpurun_out/valu_pad.txt; : > $O
for lib in hoomd_tf_amd/libhtf_amd.so build_variants/libhtf_pad8.so build_variants/libhtf_pad16.so hoomd_tf_amd/libhtf_amd.so; do
  HTF_AMD_LIB=$lib timeout 200 python tools/fused_ab.py --tag $(basename $lib) 2>&1 | tail -1 | cut -c1-28,90- >> $O
  HTF_AMD_LIB=$lib timeout 200 python tools/fused_ab.py --f64 --tag $(basename $lib)-f64 2>&1 | tail -1 | cut -c1-28,90- >> $O
  HTF_AMD_LIB=$lib timeout 200 python tools/fused_ab.py --cells 20 --tag $(basename $lib)-c20 2>&1 | tail -1 | cut -c1-28,90- >> $O
  HTF_AMD_LIB=$lib TAG=$(basename $lib) timeout 120 python tools/fused2_ab.py 2>&1 | grep "tensor=1 rdf=1\|tensor=0 rdf=0" | tr '\n' ' ' >> $O; echo >> $O
done
cat $O
