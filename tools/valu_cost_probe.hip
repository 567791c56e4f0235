// Issue cost of single vector instructions of the pair-MLP tile, W waves per SIMD on one CU, 1152 independent copies per wave.
//   build: hipcc -O3 --offload-arch=gfx950 tools/valu_cost_probe.hip -o tools/valu_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K>
__global__ __launch_bounds__(768) void cost_kernel(unsigned long long *out, int iters, float seed) {
    float f[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = seed + 0.001f * (float)(threadIdx.x + i);
    unsigned pk[4] = {1u, 2u, 3u, 4u};
    unsigned long long smask = __builtin_amdgcn_ballot_w64(seed > 0.5f), sm[4] = {0, 0, 0, 0}, sbase = (unsigned long long)(size_t)out;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 72; ++g) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float &x = f[j];
                const float y = f[(j + 5) & 15], z = f[(j + 11) & 15];
                if (K == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (K == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                if (K == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (K == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                if (K == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[j & 3]) : "v"(x), "v"(y));
                if (K == 5) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(pk[j & 3]) : "v"(pk[(j + 1) & 3]), "v"(x));
                if (K == 6) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(x) : "v"(pk[j & 3]), "v"(y));
                if (K == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                if (K == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
                if (K == 9) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x) : "v"(y));
                if (K == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double *)&f[(2 * j) & 14]) : "v"(*(double *)&f[(2 * j + 4) & 14]), "v"(*(double *)&f[(2 * j + 8) & 14]));
                if (K == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&f[(2 * j) & 14]) : "v"(*(double *)&f[(2 * j + 4) & 14]));
                if (K == 12) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(seed), "v"(z));
                if (K == 13) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x) : "s"(seed));
                if (K == 14) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&f[(2 * j) & 14]) : "v"(*(double *)&f[(2 * j + 4) & 14]));
                if (K == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "s"(smask));
                if (K == 16) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(y) : "vcc");
                if (K == 17) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sm[j & 3]) : "v"(x), "v"(y));
                if (K == 18) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(x), "v"(y) : "vcc");
                if (K == 19) asm volatile("v_mul_f32 %0, 0x40490fdb, %0" : "+v"(x));
                if (K == 20) asm volatile("v_mul_f32 %0, 2.0, %0" : "+v"(x));
                if (K == 21) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x) : "s"(seed));
                if (K == 22) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "s"(seed));
                if (K == 23) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
                if (K == 24) asm volatile("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(*(unsigned long long *)&f[(2 * j) & 14]) : "v"(*(unsigned long long *)&f[(2 * j + 4) & 14]), "s"(sbase));
                if (K == 25) asm volatile("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(*(unsigned long long *)&f[(2 * j) & 14]) : "v"(*(unsigned long long *)&f[(2 * j + 4) & 14]), "v"(*(unsigned long long *)&f[(2 * j + 8) & 14]));
                if (K == 26) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(pk[j & 3]) : "s"((unsigned)smask));
                if (K == 27) asm volatile("v_add_u32 %0, %1, %0" : "+v"(pk[j & 3]) : "s"((unsigned)smask));
                if (K == 28) asm volatile("v_add_u32 %0, %1, %0" : "+v"(pk[j & 3]) : "v"(pk[(j + 1) & 3]));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)(pk[0] + pk[1] + pk[2] + pk[3]) + (float)(sm[0] + sm[1] + sm[2] + sm[3]);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += f[i];
    if (s == 1.2345f) out[63] = 0;
    if ((threadIdx.x & 63) == 0) {
        out[2 * (threadIdx.x >> 6)] = t0;
        out[2 * (threadIdx.x >> 6) + 1] = t1;
    }
}

template <int K>
static void run(const char *what, unsigned long long *d) {
    const int iters = 50;
    printf("%-28s", what);
    for (int W = 1; W <= 3; ++W) {
        unsigned long long h[64];
        for (int r = 0; r < 2; ++r) {
            hipLaunchKernelGGL(cost_kernel<K>, dim3(1), dim3(256 * W), 0, 0, d, iters, 1.25f);
            (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        }
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < 4 * W; ++w) {
            lo = h[2 * w] < lo ? h[2 * w] : lo;
            hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi;
        }
        printf("  W=%d %5.2f", W, (double)(hi - lo) / (double)(iters * W * 72 * 16));
    }
    printf("   ticks per instruction per SIMD\n");
}

int main() {
    unsigned long long *d;
    (void)hipMalloc(&d, 64 * 8);
    run<0>("v_fma_f32 (3 vgpr)", d);
    run<9>("v_fma_f32 (2 vgpr + const)", d);
    run<12>("v_fma_f32 (sgpr operand)", d);
    run<1>("v_fmac_f32", d);
    run<2>("v_mul_f32", d);
    run<3>("v_add_f32", d);
    run<13>("v_sub_f32 (sgpr)", d);
    run<4>("v_cvt_pk_f16_f32", d);
    run<5>("v_fma_mixlo_f16", d);
    run<6>("v_fma_mix_f32", d);
    run<7>("v_exp_f32", d);
    run<8>("v_rcp_f32", d);
    run<10>("v_pk_fma_f32", d);
    run<11>("v_pk_mul_f32", d);
    run<14>("v_pk_add_f32", d);
    run<15>("v_cndmask_b32 (sgpr mask)", d);
    run<16>("v_cndmask_b32 (vcc)", d);
    run<17>("v_cmp_lt_f32 -> sgpr pair", d);
    run<18>("v_cmp_lt_f32 -> vcc", d);
    run<19>("v_mul_f32 (literal)", d);
    run<20>("v_mul_f32 (inline const)", d);
    run<21>("v_mul_f32 (sgpr)", d);
    run<22>("v_mov_b32 v, s", d);
    run<23>("v_mov_b32 v, v", d);
    run<24>("v_lshl_add_u64 (sgpr base)", d);
    run<25>("v_lshl_add_u64 (vgpr base)", d);
    run<26>("v_mbcnt_lo (sgpr mask)", d);
    run<27>("v_add_u32 (sgpr)", d);
    run<28>("v_add_u32 (vgpr)", d);
    return 0;
}
