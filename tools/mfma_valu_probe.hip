// Does VALU work run in the shadow of MFMA work on gfx950 -- inside one wave, and across the waves of a SIMD?
//   build: hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_probe.hip -o tools/mfma_valu_probe
// Every wave runs ITER rounds of {NM dependent-chain MFMAs, NV independent VALU fmas}.
//   mode 0: MFMA only          mode 1: VALU only
//   mode 2: both, MFMAs first then the VALU block (as a compiler clusters them)
//   mode 3: both, one MFMA then NV/NM fmas, repeated (hand interleave, sched_group_barrier)
//   mode 4: 512-thread workgroups, waves 0-3 MFMA only, waves 4-7 VALU only (cross-wave overlap:
//           waves w and w + 4 share a SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

constexpr int NM = 12, NV = 96, ITER = 2000;

// VK: 0 = v_fma_f32, 1 = integer (v_xad/v_and), 2 = v_exp_f32 (transcendental)
template <int VK>
__device__ __forceinline__ float vop(float x, float c1, float c2) {
    if constexpr (VK == 0) return fmaf(x, c1, c2);
    else if constexpr (VK == 1) return __uint_as_float((__float_as_uint(x) ^ __float_as_uint(c1)) + __float_as_uint(c2));
    else if constexpr (VK == 2) return __builtin_amdgcn_exp2f(x);
    else if constexpr (VK == 3) return x * c1;           // v_mul_f32 (VOP2)
    else if constexpr (VK == 4) return x + c1;           // v_add_f32 (VOP2)
    else return fmaf(c1, c2, x);                         // v_fmac_f32 (VOP2 form of the fma)
}

template <int MODE, bool F32 = false, int VK = 0>
__global__ __launch_bounds__(512) void k(float *out, float seed) {
    const unsigned lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // scalar: real branches
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = seed * i;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + lane + i;
    const float c1 = seed * 0.5f, c2 = seed * 0.25f;
    const bool do_m = MODE == 0 || MODE == 2 || MODE == 3 || (MODE == 4 && (wave & 4) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || MODE == 3 || (MODE == 4 && (wave & 4) != 0);
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (F32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, c2, acc, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV / NM; ++j) v[(m * (NV / NM) + j) & 15] = vop<VK>(v[(m * (NV / NM) + j) & 15], c1, c2);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NV / NM, 0);
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    if (F32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, c2, acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                }
            }
            if (do_v) {
#pragma unroll
                for (int j = 0; j < NV; ++j) v[j & 15] = vop<VK>(v[j & 15], c1, c2);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + wave;
}

template <int MODE, bool F32 = false, int VK = 0>
static void run(const char *name, int blocks_per_cu, float *out, int threads = 256) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, F32, VK>), dim3(grid), dim3(threads), 0, 0, out, 1e-3f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // cycles per round per wave at 2.4 GHz
    printf("%-64s %d wave(s)/SIMD  %8.3f ms  %7.0f cycles/round\n", name, blocks_per_cu, best, best * 1e-3 * 2.4e9 / ITER);
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    printf("round = %d MFMA 32x32x16 bf16 (dependent chain, 32 cycles each = %d) + %d VALU fma (4 cycles each = %d)\n", NM, NM * 32,
           NV, NV * 4);
    for (int w = 1; w <= 2; ++w) {
        run<0>("MFMA only", w, out);
        run<1>("VALU only", w, out);
        run<2>("MFMA block then VALU block (one wave does both)", w, out);
        run<3>("interleaved 1 MFMA : 8 VALU (one wave does both)", w, out);
    }
    run<0>("MFMA only, 512-thread workgroups (2 waves/SIMD)", 1, out, 512);
    run<1>("VALU only, 512-thread workgroups (2 waves/SIMD)", 1, out, 512);
    run<4>("waves 0-3 MFMA only, waves 4-7 VALU only (one of each per SIMD)", 1, out, 512);
    printf("-- the same with v_mfma_f32_32x32x2_f32 (64 cycles each = %d)\n", NM * 64);
    run<0, true>("MFMA only", 1, out);
    run<2, true>("MFMA block then VALU block (one wave does both)", 1, out);
    run<3, true>("interleaved 1 MFMA : 8 VALU (one wave does both)", 1, out);
    run<0, true>("MFMA only, 512-thread workgroups (2 waves/SIMD)", 1, out, 512);
    run<4, true>("waves 0-3 MFMA only, waves 4-7 VALU only (one of each per SIMD)", 1, out, 512);
    printf("-- bf16 MFMA beside integer VALU ops, then beside v_exp_f32\n");
    run<1, false, 1>("integer VALU only", 1, out);
    run<2, false, 1>("MFMA block then integer VALU block (one wave)", 1, out);
    run<3, false, 1>("interleaved 1 MFMA : 8 integer VALU (one wave)", 1, out);
    run<4, false, 1>("waves 0-3 MFMA only, waves 4-7 integer VALU only", 1, out, 512);
    run<1, false, 2>("v_exp only", 1, out);
    run<3, false, 2>("interleaved 1 MFMA : 8 v_exp (one wave)", 1, out);
    run<4, false, 2>("waves 0-3 MFMA only, waves 4-7 v_exp only", 1, out, 512);
    printf("-- bf16 MFMA beside other fp32 VALU ops\n");
    run<1, false, 3>("v_mul_f32 only", 1, out);
    run<3, false, 3>("interleaved 1 MFMA : 8 v_mul_f32 (one wave)", 1, out);
    run<1, false, 4>("v_add_f32 only", 1, out);
    run<3, false, 4>("interleaved 1 MFMA : 8 v_add_f32 (one wave)", 1, out);
    run<1, false, 5>("v_fmac_f32 only", 1, out);
    run<3, false, 5>("interleaved 1 MFMA : 8 v_fmac_f32 (one wave)", 1, out);
    return 0;
}
