// Which VALU encodings / op classes run in the shadow of a bf16 MFMA chain on gfx950?
//   build: hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_probe2.hip -o tools/mfma_valu_probe2
// One wave per SIMD; per round 12 dependent v_mfma_f32_32x32x16_bf16 (32 cycles each) interleaved
// 1 : 8 with 96 VALU instructions of ONE kind, written in inline asm so that the encoding is known.
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr int NM = 12, PER = 8, ITER = 2000;

#define OP_LIST(X)                                                                                   \
    X(0, "v_mul_f32_e32  (VOP2)", "v_mul_f32_e32 %0, %1, %0", 0)                                      \
    X(1, "v_mul_f32_e64  (VOP3)", "v_mul_f32_e64 %0, %1, %0", 0)                                      \
    X(2, "v_add_f32_e32  (VOP2)", "v_add_f32_e32 %0, %1, %0", 0)                                      \
    X(3, "v_fmac_f32_e32 (VOP2)", "v_fmac_f32_e32 %0, %1, %1", 0)                                     \
    X(4, "v_fma_f32      (VOP3)", "v_fma_f32 %0, %0, %1, %1", 0)                                      \
    X(5, "v_and_b32_e32  (VOP2)", "v_and_b32_e32 %0, %1, %0", 0)                                      \
    X(6, "v_and_b32_e64  (VOP3)", "v_and_b32_e64 %0, %1, %0", 0)                                      \
    X(7, "v_perm_b32     (VOP3)", "v_perm_b32 %0, %0, %1, %1", 0)                                     \
    X(8, "v_exp_f32_e32  (VOP1)", "v_exp_f32_e32 %0, %0", 0)                                          \
    X(9, "v_rcp_f32_e32  (VOP1)", "v_rcp_f32_e32 %0, %0", 0)                                          \
    X(10, "v_cvt_pk_bf16_f32 (VOP3)", "v_cvt_pk_bf16_f32 %0, %0, %1", 0)                              \
    X(11, "v_mov_b32_e32 (VOP1)", "v_mov_b32_e32 %0, %1", 0)

template <int OP>
__device__ __forceinline__ void one(float &x, float c) {
#define X(id, name, text, _) if constexpr (OP == id) asm volatile(text : "+v"(x) : "v"(c));
    OP_LIST(X)
#undef X
}
template <int OP>
__device__ __forceinline__ void one_pk(f32x2 &x, f32x2 c) {
    if constexpr (OP == 20) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(x) : "v"(c));
    if constexpr (OP == 21) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(x) : "v"(c));
    if constexpr (OP == 22) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
}

// MODE 0: MFMA only, 1: VALU only, 2: interleaved
template <int OP, int MODE, bool F32 = false>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = seed * i;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    float v[PER];
    f32x2 w[PER];
    for (int i = 0; i < PER; ++i) { v[i] = seed + threadIdx.x + i; w[i] = f32x2{v[i], v[i] + 1.f}; }
    const float c = 1.0f + seed;
    const f32x2 c2 = {c, c};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (MODE != 1) {
                if constexpr (F32) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(c));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            }
            if (MODE != 0) {
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    if constexpr (OP >= 20) one_pk<OP>(w[j], c2); else one<OP>(v[j], c);
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < PER; ++i) s += v[i] + w[i][0] + w[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static int g_blocks = 256; // 256 = one wave per SIMD, 512 = two
template <int OP, int MODE, bool F32 = false>
static float run(float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP, MODE, F32>), dim3(g_blocks), dim3(256), 0, 0, out, 1e-3f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1e-3f * 2.4e9f / ITER; // cycles per round at 2.4 GHz
}

template <int OP, bool F32 = false>
static void report(const char *name, float mfma, float *out) {
    const float v = run<OP, 1, F32>(out), both = run<OP, 2, F32>(out);
    printf("%-28s alone %6.0f   with MFMA %6.0f   hidden %5.0f of %5.0f VALU cycles (%3.0f %%)\n", name, v, both, mfma + v - both, v,
           100.f * (mfma + v - both) / v);
}

int main() {
    float *out;
    hipMalloc(&out, 512 * 256 * sizeof(float));
    const float mfma = run<0, 0>(out);
    printf("per round: %d dependent MFMAs alone = %.0f cycles; %d VALU instructions of one kind, 8 after every MFMA\n", NM, mfma, NM * PER);
#define X(id, name, text, _) report<id>(name, mfma, out);
    OP_LIST(X)
#undef X
    report<20>("v_pk_mul_f32  (VOP3P)", mfma, out);
    report<21>("v_pk_add_f32  (VOP3P)", mfma, out);
    report<22>("v_pk_fma_f32  (VOP3P)", mfma, out);
    g_blocks = 512;
    const float mfma2 = run<0, 0>(out);
    printf("-- TWO waves per SIMD, bf16 MFMA: %d dependent MFMAs per wave alone = %.0f cycles per round\n", NM, mfma2);
    report<0>("v_mul_f32_e32  (VOP2)", mfma2, out);
    report<4>("v_fma_f32      (VOP3)", mfma2, out);
    report<5>("v_and_b32_e32  (VOP2)", mfma2, out);
    report<7>("v_perm_b32     (VOP3)", mfma2, out);
    report<8>("v_exp_f32_e32  (VOP1)", mfma2, out);
    g_blocks = 256;
    const float mfma32 = run<0, 0, true>(out);
    printf("-- beside %d dependent v_mfma_f32_32x32x2_f32 (64 cycles each): alone = %.0f cycles\n", NM, mfma32);
    report<0, true>("v_mul_f32_e32  (VOP2)", mfma32, out);
    report<4, true>("v_fma_f32      (VOP3)", mfma32, out);
    report<5, true>("v_and_b32_e32  (VOP2)", mfma32, out);
    report<8, true>("v_exp_f32_e32  (VOP1)", mfma32, out);
    report<22, true>("v_pk_fma_f32  (VOP3P)", mfma32, out);
    return 0;
}
