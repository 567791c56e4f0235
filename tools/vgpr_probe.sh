#!/bin/bash
# Register counts of the one-kernel step's instantiations in SECONDS instead of the library's four-minute build of fused_eval.hip:
# a unit that includes csrc/fused_eval.hip as the generated units do (HTF_JIT_UNIT: templates only) and instantiates just the
# kernels asked for, compiled with --save-temps; prints .vgpr_count / .sgpr_spill_count / scratch / LDS per kernel.
#   tools/vgpr_probe.sh [float|double] [rows: 2|4] [extra hipcc flags]        (round 6: how the step epilogue's 77 -> 55 VGPRs were found)
set -e
PT=${1:-float}; R=${2:-4}; shift 2 2>/dev/null || true
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
T=$(mktemp -d)
V4=$([ "$PT" = double ] && echo double4 || echo float4)
cat > $T/probe.hip <<SRC
#define HTF_JIT_UNIT 1
#include "fused_eval.hip"
using namespace htf;
#define INST(EP) template __global__ void htf::fused_forces_tails_kernel<HTF_POT_LJ, true, $R, $PT, EP>( \\
    const $V4 *__restrict__, unsigned, unsigned, unsigned, unsigned, BoxT<$PT>, const unsigned *__restrict__, const unsigned *__restrict__, \\
    const unsigned *__restrict__, $PT, void *__restrict__, int, PotParams, unsigned *__restrict__, float4 *__restrict__, float4 *__restrict__, \\
    unsigned *__restrict__, const StepEpilogue<$PT> *__restrict__);
INST(0)
INST(1)
INST(2)
SRC
cd $T
/opt/rocm/bin/hipcc -std=c++17 -O3 -ffp-contract=on --offload-arch=gfx950 -I$ROOT/include -I$ROOT/hoomd_tf_amd/csrc -DHTF_BUILD -w "$@" -c probe.hip --save-temps -o probe.o
echo "fused_forces_tails_kernel<LJ, store, $R rows, $PT, epilogue level 0 / 1 / 2>:"
grep -E "^\s+\.vgpr_count|sgpr_spill_count|group_segment_fixed_size:|private_segment_fixed_size:" probe-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - -
rm -rf $T
