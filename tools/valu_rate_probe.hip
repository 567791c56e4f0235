// Issue cost (cycles per wave64 instruction on one SIMD, independent instructions, one wave per SIMD) of the VALU
// instructions the fp64-wire force kernel is made of, against their fp32 twins.  s_memtime around an unrolled block.
//   build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define BLOCK(name, body)                                                                          \
    __global__ void name(unsigned long long *out, double a, double b) {                           \
        double d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = a + 2, d5 = b + 2, d6 = a + 3, d7 = b + 3; \
        float f0 = (float)a, f1 = (float)b, f2 = f0 + 1, f3 = f1 + 1, f4 = f0 + 2, f5 = f1 + 2, f6 = f0 + 3, f7 = f1 + 3; \
        unsigned long long t0, t1;                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
        for (int i = 0; i < 16; ++i) { REP8(body) }                                                \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");                \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                           \
        if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 == 1.2345) out[1] = 0; \
    }
#define D8(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" \
    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b));
#define F8(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" \
    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(f1 * 0.f + 1.5f));
#define D8U(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" \
    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
#define F8U(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" \
    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
#define CVT8 asm volatile("v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15" \
    : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3), "=v"(f4), "=v"(f5), "=v"(f6), "=v"(f7) : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));
#define FMA8D asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8" \
    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b));
#define CMP8D asm volatile("v_cmp_gt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %1, %2\n v_cmp_gt_f64 vcc, %2, %3\n v_cmp_gt_f64 vcc, %3, %4\n v_cmp_gt_f64 vcc, %4, %5\n v_cmp_gt_f64 vcc, %5, %6\n v_cmp_gt_f64 vcc, %6, %7\n v_cmp_gt_f64 vcc, %7, %0" \
    :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7) : "vcc");

BLOCK(k_add_f32, F8("v_add_f32"))
BLOCK(k_mul_f32, F8("v_mul_f32"))
BLOCK(k_rndne_f32, F8U("v_rndne_f32"))
BLOCK(k_add_f64, D8("v_add_f64"))
BLOCK(k_mul_f64, D8("v_mul_f64"))
BLOCK(k_fma_f64, FMA8D)
BLOCK(k_rndne_f64, D8U("v_rndne_f64"))
BLOCK(k_cvt_f32_f64, CVT8)
BLOCK(k_cmp_gt_f64, CMP8D)

int main() {
    unsigned long long *d, h[4];
    hipMalloc(&d, 64);
#define RUN(k)                                                             \
    for (int r = 0; r < 2; ++r) {                                          \
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1.25, 0.75);     \
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);                         \
    }                                                                      \
    printf("%-16s %6.2f s_memtime ticks per instruction (128 instructions; 100 MHz ticks x core clock ratio)\n", #k, (double)h[0] / 128.0);
    RUN(k_add_f32) RUN(k_mul_f32) RUN(k_rndne_f32) RUN(k_add_f64) RUN(k_mul_f64) RUN(k_fma_f64) RUN(k_rndne_f64) RUN(k_cvt_f32_f64) RUN(k_cmp_gt_f64)
    return 0;
}
