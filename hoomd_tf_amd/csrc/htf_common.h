// Internal helpers shared by the HIP translation units (not part of the ABI).
#pragma once
// (__HIPCC_RTC__: a generated unit under hipRTC, csrc/jit.hip -- the device runtime is built in, there is no host side)
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#endif

#include "htf_amd.h"

namespace htf {

#ifndef __HIPCC_RTC__
void set_error(const char *fmt, ...);

#define HTF_CHECK_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            htf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return HTF_ERR_DEVICE;                                                       \
        }                                                                                \
    } while (0)

#define HTF_REQUIRE(cond, ...)                                                           \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            htf::set_error(__VA_ARGS__);                                                 \
            return HTF_ERR_INVALID;                                                      \
        }                                                                                \
    } while (0)

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return HTF_ERR_DEVICE;
    }
    return HTF_OK;
}
#endif // __HIPCC_RTC__

// a scalar lane mask read back as this lane's predicate (no vector instruction where the compiler has the builtin; the hipRTC that
// ships inside a PyTorch wheel may be a release older than the hipcc the library was built with and lack it)
__device__ __forceinline__ bool inverse_ballot64(unsigned long long m) {
#if __has_builtin(__builtin_amdgcn_inverse_ballot_w64)
    return __builtin_amdgcn_inverse_ballot_w64(m);
#else
    return ((m >> __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))) & 1ull) != 0ull;
#endif
}

// reference constants: simmodel.py:627-628 (nlist_rinv), :581 (safe_norm default)
constexpr float kRinvDelta = 3e-6f;
constexpr float kNormDelta = 1e-7f;

template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

// -------- wave64 lane-group reductions (G consecutive lanes, G a power of two) --------
// DPP row operations cover groups up to 16 lanes without touching LDS; 32/64 need
// the cross-row bpermute path.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// v + (v of lane ^ 32) and v + (v of lane ^ 16) without the LDS crossbar: gfx950's
// v_permlane32_swap / v_permlane16_swap exchange the upper half (odd rows) of one register with the
// lower half (even rows) of another; fed the same value twice they return (own half | own half) and
// (other half | other half), whose sum is the butterfly step.  Same bits as v + __shfl_xor(v, 32 / 16)
// (fp addition commutes), one VALU instruction instead of a ds_bpermute round trip.  A/B on one box
// (the LJ step, the C4 sweep, the pair-MLP evaluators): no measurable time difference either way;
// run-to-run spread between GPU boxes is ~5 %.
__device__ __forceinline__ float sum_xor32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// a + b reduced one butterfly step each, in ONE swap and ONE add: the lower half of the result holds a(l) + a(l + 32), the upper
// half b(l - 32) + b(l) (v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second);
// the 16-lane form leaves (a.r0 + a.r1 | b.r0 + b.r1 | a.r2 + a.r3 | b.r2 + b.r3) in rows 0..3.  Two values per step instead of
// one: the kernels that finish a row with several wave-wide sums (fx, fy, fz, e) are VALU-issue bound.
__device__ __forceinline__ float pair_sum_xor32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pair_sum_xor16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// wave-wide sums of four values in 10 instructions (32 one by one): rows 0..3 of the result hold sum(a), sum(c), sum(b), sum(d)
__device__ __forceinline__ float wave_sum4(float a, float b, float c, float d) {
    float v = pair_sum_xor16(pair_sum_xor32(a, b), pair_sum_xor32(c, d));
    v += dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v); // row_half_mirror
    v += dpp_mov<0x140>(v); // row_mirror
    return v;
}
// the wave ballot of a bool as the compare mask itself (__ballot(int) re-materialises the predicate: two VALU instructions)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// A wave-uniform value held in a VECTOR register by every lane.  hipcc keeps uniform values in scalar registers, and a vector
// instruction with a scalar source -- a constant multiplier, a mask, a base address -- issues at 4.1-4.4 cycles with two or more
// waves on the SIMD where one with vector (or inline-constant) sources takes 2.2-2.5: tools/valu_cost_probe.hip.
__device__ __forceinline__ float in_vgpr(float s) {
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}
__device__ __forceinline__ double in_vgpr(double s) {
    const unsigned long long q = (unsigned long long)__double_as_longlong(s);
    const float lo = in_vgpr(__uint_as_float((unsigned)q)), hi = in_vgpr(__uint_as_float((unsigned)(q >> 32)));
    return __longlong_as_double((long long)(((unsigned long long)__float_as_uint(hi) << 32) | __float_as_uint(lo)));
}

template <int G>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(G >= 1 && G <= 64 && (G & (G - 1)) == 0, "G must be a power of two <= 64");
    if constexpr (G >= 2) v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += dpp_mov<0x141>(v);  // row_half_mirror
    if constexpr (G >= 16) v += dpp_mov<0x140>(v); // row_mirror
    if constexpr (G >= 32) v = sum_xor16(v);
    if constexpr (G >= 64) v = sum_xor32(v);
    return v;
}

#ifndef __HIPCC_RTC__
// compute units of the current device (host side; 256 on MI355X)
inline int device_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        return 256;
    return n;
}
#endif

// Streaming (nontemporal) 16/32-byte store for the pair-vector tensor.  The tensor is 268 MB per step
// at C3 and is written once; with ordinary (write-back, allocating) stores it sweeps the L2 and the
// Infinity Cache clean of the index rows and the 2 MB position table the same kernel is gathering
// from, and the kernel's load and store phases add up instead of overlapping: 93 us.  With `nt`
// stores: 63 us.  The compiler merges the component stores into one global_store_dwordx4 ... nt.
__device__ __forceinline__ void store_stream(float4 *p, const float4 &v) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z);
    __builtin_nontemporal_store(v.w, &p->w);
}
__device__ __forceinline__ void store_stream(double4 *p, const double4 &v) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z);
    __builtin_nontemporal_store(v.w, &p->w);
}

// The pair-vector tensor is read ONCE per step by whatever evaluates it and rewritten by the next step's builder: read with
// ordinary loads its 268 MB stay in the cache hierarchy, and the builder's streaming stores then take 78-82 us at C3 where they
// take 52-53 behind non-temporal loads (bench.py --workload mlp, round 4).  Every consumer of the tensor loads it through these.
__device__ __forceinline__ float4 load_stream(const float4 *p) {
    return make_float4(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y), __builtin_nontemporal_load(&p->z),
                       __builtin_nontemporal_load(&p->w));
}
__device__ __forceinline__ double4 load_stream(const double4 *p) {
    return make_double4(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y), __builtin_nontemporal_load(&p->z),
                        __builtin_nontemporal_load(&p->w));
}

// tf.norm(nlist[:, :, :3], axis=2) as compute_rdf takes it (simmodel.py:661): sqrt(reduce_sum(square)), every product
// and sum rounded on its own (no fused multiply-add), correctly rounded square root.  The histogram bin of a pair
// sitting within an ulp of a bin edge depends on these roundings: at 33 M slots (C4) a contracted x*x + y*y + z*z
// moves a handful of pairs across an edge relative to TensorFlow's (and the oracle's) arithmetic.
__device__ __forceinline__ float plain_sq3(float x, float y, float z) {
#pragma clang fp contract(off)
    const float xx = x * x, yy = y * y, zz = z * z;
    return (xx + yy) + zz;
}
__device__ __forceinline__ float plain_norm3(float x, float y, float z) { return sqrtf(plain_sq3(x, y, z)); }

// number of set bits of a wave ballot below this lane: v_mbcnt_lo/hi, two instructions
// (the portable popcount(m & ((1 << lane) - 1)) costs a 64-bit shift, mask and two bit counts)
__device__ __forceinline__ unsigned ballot_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

template <int G>
__device__ __forceinline__ unsigned group_sum_u(unsigned v) {
#pragma unroll
    for (int m = 1; m < G; m <<= 1) v += (unsigned)__shfl_xor((int)v, m);
    return v;
}

} // namespace htf
