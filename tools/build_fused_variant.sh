#!/bin/bash
# Rebuild ONLY fused_eval.o with extra flags and link it with the tree's other objects into build_variants/libhtf_<name>.so
# (same-box A/B of one kernel file: HTF_AMD_LIB=build_variants/libhtf_<name>.so).  usage: tools/build_fused_variant.sh <name> "<flags>"
set -e
cd "$(dirname "$0")/../hoomd_tf_amd/csrc"
NAME=$1; EXTRA=$2
mkdir -p ../../build_variants
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I../../include -I. -w -fvisibility=hidden -DHTF_BUILD $EXTRA \
  -ffp-contract=on -c fused_eval.hip -o ../../build_variants/fused_eval_$NAME.o
OBJS=$(ls *.o | grep -v fused_eval.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_variants/libhtf_$NAME.so $OBJS ../../build_variants/fused_eval_$NAME.o -ldl
echo built build_variants/libhtf_$NAME.so
