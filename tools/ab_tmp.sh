cd $GRAFT_REPO_ROOT
O=gpurun_out/c4_pipe.txt; : > $O
HTF_AMD_LIB=build_variants/libhtf_pipe2.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_tensorflow.py -q -m gpu -x -k "eds or c4 or forces2 or rdf or sweep or two" 2>&1 | tail -3 >> $O
for lib in build_variants/libhtf_pipe2.so hoomd_tf_amd/libhtf_amd.so build_variants/libhtf_pipe2.so hoomd_tf_amd/libhtf_amd.so; do
  HTF_AMD_LIB=$lib TAG=$(basename $lib) timeout 120 python tools/fused2_ab.py 2>&1 | grep "tensor=" | tr '\n' ' ' >> $O; echo >> $O
done
cat $O
