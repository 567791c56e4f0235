#!/bin/bash
# Rebuild ONE object with extra flags and link it with the tree's other objects into build_variants/libhtf_<name>.so
# usage: tools/build_obj_variant.sh <name> <file-stem> "<flags>"      (same-box A/B: HTF_AMD_LIB=build_variants/libhtf_<name>.so)
set -e
cd "$(dirname "$0")/../hoomd_tf_amd/csrc"
NAME=$1; STEM=$2; EXTRA=$3
mkdir -p ../../build_variants
FP=""; [ "$STEM" = fused_eval ] && FP="-ffp-contract=on"; [ "$STEM" = pair_vectors ] && FP="-ffp-contract=off"; [ "$STEM" = pair_mlp ] && FP="-fno-slp-vectorize"; [ "$STEM" = mlp_train16 ] && FP="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form"
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I../../include -I. -w -fvisibility=hidden -DHTF_BUILD $EXTRA $FP \
  -c $STEM.hip -o ../../build_variants/${STEM}_$NAME.o
OBJS=$(ls *.o | grep -v "^$STEM.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/libhtf_$NAME.so $OBJS ../../build_variants/${STEM}_$NAME.o -ldl
echo built build_variants/libhtf_$NAME.so
