for rep in 1 2; do
for lib in base w8 w7 w6; do
  export HTF_AMD_LIB=build_variants/libhtf_$lib.so
  python tools/fused_ab.py --relax 100 --tag f32_$lib 2>&1 | tail -1
  HTF_FUSED_TAILS=2 python tools/fused_ab.py --f64 --relax 100 --tag f64_t2_$lib 2>&1 | tail -1
  HTF_FUSED_TAILS=4 python tools/fused_ab.py --f64 --relax 100 --tag f64_t4_$lib 2>&1 | tail -1
done; done
