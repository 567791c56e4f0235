#!/bin/bash
# Build another copy of libhtf_amd.so with extra compile flags, for same-box A/B runs
# (HTF_AMD_LIB=build_variants/libhtf_<name>.so).   usage: tools/build_variant.sh <name> "<extra flags>"
set -e
cd "$(dirname "$0")/.."
NAME=$1; EXTRA=$2
OUT=build_variants/$NAME
mkdir -p $OUT
cp -p -u hoomd_tf_amd/csrc/*.hip hoomd_tf_amd/csrc/*.h hoomd_tf_amd/csrc/Makefile $OUT/   # (-p -u: an unchanged source keeps its mtime, make rebuilds only what changed)
make -s -j8 -C $OUT ROOT=$(pwd) CXXFLAGS_EXTRA="$EXTRA" LIB=../libhtf_$NAME.so
echo built build_variants/libhtf_$NAME.so
