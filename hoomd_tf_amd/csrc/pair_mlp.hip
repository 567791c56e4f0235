// Fused pair-MLP evaluator on the matrix cores:
//   nlist [B, NN, 4] -> force [B] Scalar4,   per slot:
//   r = safe_norm(x) (simmodel.py:581-594) -> phi = RBFExpansion(r) (layers.py:46-49)
//   -> Dense(H1) -> act -> Dense(H2) -> act -> Dense(1) = u      (Keras Dense, a18)
//   E_i = 1/2 sum_j [r > 3e-6] u_ij ;  F_i = 2 sum_j dE_i/dx_ij   (simmodel.py:526-555)
// with the analytic backward pass (du/dr) fused in, so nothing but the 16-B slot is read
// and one Scalar4 per particle is written.  TF would materialise [N,NN,K] and
// [N,NN,H] tensors (2-4 GiB at N = 131072) forward and backward.
//
// MI355X mapping (MFMA-bound: 24.6 kflop per 16-B slot).  One wave owns a tile of 32
// pairs.  Every layer is computed TRANSPOSED, Z^T[feature][pair] = W^T X^T, with
// v_mfma_f32_32x32x2_f32 (exact fp32): features run over the accumulator rows
// (registers), pairs over the lanes.  A 32x32 accumulator register v of lane (p, h)
// holds feature f0(v) + 4h, f0(v) = (v&3) + 8(v>>2), of pair p -- which is exactly a B
// operand (k on the lane half, column on lane&31) of the next layer's MFMA if the
// k-steps are taken in that permuted order.  The k order of a sum is free, so the
// weight (A) operands are stored pre-permuted and the activations never leave
// registers: no LDS round trip, no transposes, for the forward AND the backward chain.
// The weights live in LDS as four operand-ordered images (48 KiB, read with
// conflict-free ds_read_b128, 4 k-steps per read); blocks are persistent so the images
// are loaded once.  ~150 VGPRs -> 3 waves/SIMD, 3 blocks/CU (148 KiB LDS), so one
// wave's tanh/exp VALU phase overlaps another's MFMA phase.  Fully padded tiles (the
// tail of every row) are skipped with a wave-uniform ballot.
#include <cmath>
#include <cstring>
#include <vector>

#include "htf_common.h"
#include "htf_internal.h"
#include "pair_mlp.h"
#include "pair_math.h"

namespace htf {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// VALU work beside the matrix pipe (tools/mfma_valu_probe2.hip, one wave per SIMD, 8 VALU
// instructions after each of 12 dependent MFMAs):
//  * beside v_mfma_f32_32x32x16_bf16 ordinary VALU instructions -- VOP1/VOP2/VOP3, fp32 or
//    integer -- run in the MFMA's shadow (about two thirds of the matrix time can be filled),
//    v_exp/v_rcp half as well; the PACKED fp32 instructions (v_pk_mul/add/fma_f32) do not
//    overlap at all: 12 MFMAs + 96 v_pk_fma take the sum of their times;
//  * beside v_mfma_f32_32x32x2_f32 NOTHING overlaps: the fp32 matrix instruction and the vector
//    ALU exclude each other, so the fp32 evaluator's time is its MFMA time plus its VALU time.
// Hence two styles in this file (compiled with -fno-slp-vectorize so that hipcc does not form
// v_pk_* on its own): the fp32 path (PK = true) halves its VALU instruction count with packed
// arithmetic; the bf16 and split paths keep element-wise arithmetic and interleave it with
// their MFMAs (HTF_PIPE in the kernel).
using f32x2 = __attribute__((ext_vector_type(2))) float;

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_exp2(f32x2 x) { return f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])}; }
__device__ __forceinline__ f32x2 pk_rcp(f32x2 x) { return f32x2{__builtin_amdgcn_rcpf(x[0]), __builtin_amdgcn_rcpf(x[1])}; }

// tanh(z) = 1 - 2 / (1 + exp(2z)) on a whole accumulator tile: v_mul (2 log2(e) folded into one
// constant), v_exp_f32, v_add, v_rcp_f32, v_fma per element
// U (round 6, VERDICT r5 item 5 lever (ii); MEASURED AND NOT SHIPPED -- compiled with -DHTF_MLP_UFORM=1 only:
// tools/build_obj_variant.sh uform pair_mlp "-DHTF_MLP_UFORM=1"): the split16 tile keeps u = 1 / (1 + exp(2z)) instead of
// tanh(z) = 1 - 2u -- the fma is gone (16 of a block's vector instructions, 64 of a tile's ~790).  What consumes a tanh output is
// linear in it, so the images absorb the rest (mlp_refresh_kernel<P, true>): W t + b = (b + sum_k W_k) - 2 W u for the next layer's
// block and bias table and for the output layer's w3 and b3; backward, 1 - t^2 = 4 u (1 - u) costs the same two instructions as
// before, its constants folded into the backward blocks (B2 x -2 with w3 x -2, B1 x 4).  Same box, C3, 100 steps, twice each:
// evaluator 953-955 us against 979-982 (-2.7 %; tools/mlp_mix_probe.hip's slope said -5), 977-979 against 949-955 MD steps/s --
// and b + sum W - 2 W u CANCELS where W t does not: the energy column's error doubles (eps sum|w3| instead of eps sum|w3 t|) and
// test_pair_mlp_split_operands[split16-128-tanh] exceeds its strict bound 1.16 x.  Parity is the first gate: off.
#ifndef HTF_MLP_UFORM
#define HTF_MLP_UFORM 0
#endif
template <bool TANH, int P> struct UForm { static constexpr bool value = HTF_MLP_UFORM && TANH && P == HTF_MLP_SPLIT16; };

template <bool TANH, bool PK, bool U = false>
__device__ __forceinline__ void act_tile(f32x16 &a) {
    if constexpr (U) {
        static_assert(TANH && !PK, "u-form: the split16 tanh evaluator");
#pragma unroll
        for (int v = 0; v < 16; ++v) a[v] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a[v]) + 1.0f);
    } else if constexpr (TANH && PK) {
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
            const f32x2 e = pk_exp2(f32x2{a[v], a[v + 1]} * 2.8853900817779268f);
            const f32x2 o = pk_fma(pk_rcp(e + 1.0f), f32x2{-2.0f, -2.0f}, f32x2{1.0f, 1.0f});
            a[v] = o[0];
            a[v + 1] = o[1];
        }
    } else if constexpr (TANH) {
        // bf16-typed images carry 2 log2(e) W and 2 log2(e) b in their forward blocks (mlp_refresh_kernel):
        // the accumulator IS the exponent, one multiply per element less
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            a[v] = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a[v]) + 1.0f), -2.0f, 1.0f);
            // (an empty asm the value passes through: hipcc otherwise folds this fma into the fp16 conversion of the split16
            //  operand split -- v_fma_mixlo / mixhi_f16, 8.5 cycles each, next to the fp32 fma it still needs for the residual:
            //  tools/valu_cost_probe.hip.  No instruction, so nothing for the hazard bookkeeping to miss.)
            asm("" : "+v"(a[v]));
        }
    }
}

// d <- g * act'(z) given h = act(z):  g * (1 - h^2)
template <bool TANH, bool PK, bool U = false>
__device__ __forceinline__ void act_bwd_tile(f32x16 &h_inout, const f32x16 &g) {
    if constexpr (U) { // g u (1 - u): a quarter of g (1 - t^2); the 4 lives in the next backward block's image
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float u = h_inout[v];
            h_inout[v] = g[v] * fmaf(-u, u, u);
            asm("" : "+v"(h_inout[v])); // (as below: keeps the multiply out of the fp16 conversion)
        }
    } else if constexpr (TANH && PK) {
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
            const f32x2 hv = {h_inout[v], h_inout[v + 1]};
            const f32x2 o = f32x2{g[v], g[v + 1]} * pk_fma(-hv, hv, f32x2{1.0f, 1.0f});
            h_inout[v] = o[0];
            h_inout[v + 1] = o[1];
        }
    } else {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const float hv = h_inout[v];
            h_inout[v] = TANH ? g[v] * fmaf(-hv, hv, 1.0f) : g[v];
            if constexpr (TANH) asm("" : "+v"(h_inout[v])); // (as in act_tile: keeps the multiply out of the conversion)
        }
    }
}

// sum_v a[v] * b[v] into two partial sums (even / odd v)
template <bool PK>
__device__ __forceinline__ void dot_tile(float (&acc)[2], const f32x16 &a, const f32x16 &b) {
    if constexpr (PK) {
        f32x2 s = {acc[0], acc[1]};
#pragma unroll
        for (int v = 0; v < 16; v += 2) s = pk_fma(f32x2{a[v], a[v + 1]}, f32x2{b[v], b[v + 1]}, s);
        acc[0] = s[0];
        acc[1] = s[1];
    } else {
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v & 1] = fmaf(a[v], b[v], acc[v & 1]);
    }
}

// B operand of one 32-feature block of the previous layer (an accumulator tile, see above), in the
// form the precision's MFMA takes.  Prepared once per tile and used by every output block.
template <int P> struct BOp;
template <> struct BOp<HTF_MLP_FP32> { f32x16 v; };
template <> struct BOp<HTF_MLP_BF16> { bf16x8 b[2]; };
template <> struct BOp<HTF_MLP_SPLIT> { bf16x8 hi[2], mid[2], lo[2]; };
template <> struct BOp<HTF_MLP_SPLIT16> { f16x8 hi[2], lo[2]; };

// bf16 operands: accumulator registers 8s..8s+7, converted pairwise (v_cvt_pk_bf16_f32), ARE the B
// fragment of k-step s of v_mfma_f32_32x32x16_bf16: element j of lane half h is feature
// 16s + 8(j>>2) + 4h + (j&3) -- the same feature set f0(8s+j) + 4h as in the fp32 path, so tables
// and RBF centres are shared and only the weight images differ.
//
// Split operands: x = hi + mid + lo EXACTLY, each part 8 significand bits (bf16's), by masking --
// hi = top 16 bits of x, mid = top 16 bits of x - hi, lo = x - hi - mid (at most 8 bits are left,
// so its low 16 bits are zero).  v_perm_b32 packs two upper halves into one register.
template <int P>
__device__ __forceinline__ BOp<P> prep(const f32x16 &x) {
    BOp<P> o;
    if constexpr (P == HTF_MLP_FP32) {
        o.v = x;
    } else if constexpr (P == HTF_MLP_BF16) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) o.b[s][j] = (__bf16)x[8 * s + j];
    } else if constexpr (P == HTF_MLP_SPLIT16) {
        // x = hi + lo + (<= 2^-22 |x|): hi = fp16(x) and lo = fp16(x - hi), both rounded to nearest, a PAIR of elements per
        // v_cvt_pk_f16_f32; the residual x - hi is ONE v_fma_mix_f32 that reads hi straight from its half of the packed
        // register (hi * -1 + x, exact: the difference of an fp32 and its 11-bit rounding fits fp32).  Four instructions per
        // pair of elements where the three-part bf16 split needs eleven.  (v_fma_mixlo / mixhi_f16 would write the rounded
        // residual directly, three instructions per pair -- but they issue at 8.5 cycles each where v_fma_mix_f32 takes 4.3
        // and v_cvt_pk_f16_f32 4.0 with two waves on the SIMD: tools/valu_cost_probe.hip.)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 ph, pl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = x[8 * s + 2 * j], b = x[8 * s + 2 * j + 1];
                // (compiler-visible on purpose: the packed halves are MFMA operands, and hipcc counts the wait states between a
                //  vector write and the MFMA that reads it for its own instructions only, not for inline asm)
                const unsigned hp = __builtin_bit_cast(unsigned, f16x2{(_Float16)a, (_Float16)b});
                float ra, rb;
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hp), "v"(a));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hp), "v"(b));
                ph[j] = hp;
                pl[j] = __builtin_bit_cast(unsigned, f16x2{(_Float16)ra, (_Float16)rb});
            }
            o.hi[s] = __builtin_bit_cast(f16x8, ph);
            o.lo[s] = __builtin_bit_cast(f16x8, pl);
        }
    } else {
        constexpr unsigned kTop = 0xFFFF0000u, kSel = 0x07060302u; // {hi16(second), hi16(first)}
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 ph, pm, pl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = x[8 * s + 2 * j], b = x[8 * s + 2 * j + 1];
                const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
                const float ra = a - __uint_as_float(ua & kTop), rb = b - __uint_as_float(ub & kTop);
                const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
                const float la = ra - __uint_as_float(va & kTop), lb = rb - __uint_as_float(vb & kTop);
                ph[j] = __builtin_amdgcn_perm(ub, ua, kSel);
                pm[j] = __builtin_amdgcn_perm(vb, va, kSel);
                pl[j] = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), kSel);
            }
            o.hi[s] = __builtin_bit_cast(bf16x8, ph);
            o.mid[s] = __builtin_bit_cast(bf16x8, pm);
            o.lo[s] = __builtin_bit_cast(bf16x8, pl);
        }
    }
    return o;
}

#define HTF_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define HTF_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// acc += A(image) * B(prev), over one 32-feature block of the previous layer.
// fp32: 16 k-steps, the image supplies 4 steps per ds_read_b128.  bf16: two k-steps of 16.
// split: per k-step the six partial products of (a_hi + a_mid + a_lo)(b_hi + b_mid + b_lo) that are
// >= 2^-16 of the product, smallest first; mid*lo, lo*mid and lo*lo (<= 2^-24) are dropped.
template <int P>
__device__ __forceinline__ void mfma_blk(f32x16 &acc, const float *img, unsigned lane, const BOp<P> &prev) {
    if constexpr (P == HTF_MLP_FP32) {
        const float4 *p = reinterpret_cast<const float4 *>(img) + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 w = p[g * 64];
            acc = HTF_MFMA(w.x, prev.v[4 * g + 0], acc);
            acc = HTF_MFMA(w.y, prev.v[4 * g + 1], acc);
            acc = HTF_MFMA(w.z, prev.v[4 * g + 2], acc);
            acc = HTF_MFMA(w.w, prev.v[4 * g + 3], acc);
        }
    } else if constexpr (P == HTF_MLP_BF16) {
        const bf16x8 *p = reinterpret_cast<const bf16x8 *>(img) + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s) acc = HTF_MFMA_BF16(p[s * 64], prev.b[s], acc);
    } else if constexpr (P == HTF_MLP_SPLIT16) {
        const f16x8 *p = reinterpret_cast<const f16x8 *>(img) + lane; // [part 2: hi, lo][s 2][lane 64]
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f16x8 ah = p[s * 64], al = p[(2 + s) * 64];
            acc = HTF_MFMA_F16(al, prev.hi[s], acc); // smallest first; lo * lo (<= 2^-22) is dropped
            acc = HTF_MFMA_F16(ah, prev.lo[s], acc);
            acc = HTF_MFMA_F16(ah, prev.hi[s], acc);
        }
    } else {
        const bf16x8 *p = reinterpret_cast<const bf16x8 *>(img) + lane; // [part 3][s 2][lane 64]
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 ah = p[s * 64], am = p[(2 + s) * 64], al = p[(4 + s) * 64];
            acc = HTF_MFMA_BF16(al, prev.hi[s], acc);
            acc = HTF_MFMA_BF16(ah, prev.lo[s], acc);
            acc = HTF_MFMA_BF16(am, prev.mid[s], acc);
            acc = HTF_MFMA_BF16(am, prev.hi[s], acc);
            acc = HTF_MFMA_BF16(ah, prev.mid[s], acc);
            acc = HTF_MFMA_BF16(ah, prev.hi[s], acc);
        }
    }
}

// hipcc's second launch bound is waves per SIMD: two for every precision (LDS images of 49 / 25 / 73 KiB).  At three
// waves per SIMD (168 VGPRs) the fp32 kernel spills a handful of registers whichever way its arithmetic is written and is no
// faster (3.00 ms either way at C3); split with six waves per workgroup spills 50-60 registers: 2.87 ms against 1.92 ms;
// split16 with twelve waves per CU: 1.14 against 1.15 ms (DESIGN A.0).
__device__ __forceinline__ float bcast_lane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

constexpr int pipe_per(int valu, int mfma) { return (valu + mfma - 1) / mfma > 0 ? (valu + mfma - 1) / mfma : 1; }

// ONE workgroup of eight waves per CU (round 4).  The two waves of a SIMD then share LDS, which is what lets them keep pace
// with each other (see the priority rule at the top of a tile); the image is staged once per CU instead of twice.  Every
// precision: split16 1.18 -> 1.10 ms, fp32 MFMA 2.73 -> 2.65, bf16 x 3 1.72 -> 1.66, plain bf16 0.74 -> 0.70 (tools/mlp_ab.py).
#ifndef HTF_MLP_WAVES
#define HTF_MLP_WAVES 8
#endif

template <int P> struct MlpLaunch {
    static constexpr int kWaves = HTF_MLP_WAVES;                               // per workgroup
    static constexpr int kPerCU = kWaves >= 8 ? 1 : 8 / kWaves;               // workgroups per CU
    static constexpr int kPerSimd = kWaves * kPerCU / 4;
};

template <bool TANH, typename IT, int P, bool VIRIAL>
__global__ __launch_bounds__(64 * MlpLaunch<P>::kWaves, MlpLaunch<P>::kPerSimd) void pair_mlp_kernel(const typename Vec4<IT>::type *__restrict__ nlist,
                                                          unsigned B, unsigned NN, void *__restrict__ force,
                                                          int out_f64, const float *__restrict__ images, float gap,
                                                          void *__restrict__ virial9) {
    using I = Img<P>;
    constexpr bool PK = P == HTF_MLP_FP32; // packed VALU arithmetic where nothing overlaps with the MFMAs anyway
    constexpr bool kU = UForm<TANH, P>::value; // activations travel as u = 1 / (1 + exp(2z)); `images` are then the u-form set
    __shared__ __attribute__((aligned(16))) float lds[I::Floats];
    {
        const float4 *src = reinterpret_cast<const float4 *>(images);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < I::Floats / 4; i += blockDim.x) dst[i] = src[i];
    }
    __shared__ unsigned progress[12]; // tiles done, per wave
    if (threadIdx.x < 12) progress[threadIdx.x] = 0u;
    __syncthreads();

    const unsigned lane = threadIdx.x & 63u;
    const unsigned p = lane & 31u, h = lane >> 5;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    const float ginv = 1.0f / gap;
    const float nginv_l2e = -1.4426950408889634f * ginv;
    const float b3 = lds[I::TabB3];
    const f32x16 cen = load_tab(lds + I::TabC, 0, h); // centers of this lane's 16 RBF indices

    // Live slots are COMPACTED across a wave's rows before they become 32-pair tiles (round 4): the rows of a liquid hold ~95
    // live slots of 128, so their third tile carries 31 pairs and a quarter of the rows add a fourth with 1-6 -- 424 k tiles for
    // the 389 k the C3 box's 12.45 M pairs fill.  The wave stages its rows 64 slots at a time (pair vector and row index of the
    // live slots into a wave-private LDS ring, ballot + mbcnt ranks) and pops 32 pairs at a time; a tile's per-pair results go
    // back to their rows through one masked wave sum per row present in the tile (one or two), carried in scalar registers
    // until the row's last pair has been seen.  The order of a wave's pairs is fixed: deterministic.
    constexpr unsigned kRing = 96; // >= 31 left over + one chunk of 64
    constexpr int kWaves = MlpLaunch<P>::kWaves;
    const unsigned wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float ring_all[kWaves][4][kRing];
    float(&ring)[4][kRing] = ring_all[wid];
    unsigned head = 0, count = 0; // head in [0, kRing), wave-uniform
    unsigned tiles_done = 0;
#ifdef HTF_EVAL_STAMPS // experiment (tools/build_obj_variant.sh): per-phase cycle totals of one wave, printed at the end
    unsigned long long stamps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, n_tiles = 0;
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long t_begin = t_prev, r_begin = __builtin_amdgcn_s_memrealtime();
#define HTF_ESTAMP(i)                                                                                                  \
    do {                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        stamps[i] += t_ - t_prev;                                                                                      \
        t_prev = t_;                                                                                                   \
    } while (0)
#else
#define HTF_ESTAMP(i)
#endif
    const unsigned nchunks = (NN + 63) / 64;
    auto write_row = [&](unsigned wr, float ofx, float ofy, float ofz, float oen, const float (&v6)[6]) {
        if (lane == 0) {
            if (out_f64)
                ((double4 *)force)[wr] = make_double4(ofx, ofy, ofz, oen);
            else
                ((float4 *)force)[wr] = make_float4(ofx, ofy, ofz, oen);
            if constexpr (VIRIAL) {
                const float v9[9] = {v6[0], v6[1], v6[2], v6[1], v6[3], v6[4], v6[2], v6[4], v6[5]};
#pragma unroll
                for (int c9 = 0; c9 < 9; ++c9) {
                    if (out_f64)
                        ((double *)virial9)[(size_t)wr * 9 + c9] = v9[c9];
                    else
                        ((float *)virial9)[(size_t)wr * 9 + c9] = v9[c9];
                }
            }
        }
    };
    // the row whose pairs are being summed: its partial sums live in scalar registers (wave-uniform)
    unsigned open_row = 0xFFFFFFFFu;
    float cfx = 0.f, cfy = 0.f, cfz = 0.f, cfe = 0.f, cv6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float zero6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned row = wave, chunk = 0;
    bool row_any = false; // a row without a live slot gets its zeros when its last chunk has been staged
    // The chunk that will be staged next is ALREADY on its way: its load is issued when the chunk before it is taken, one or
    // two tile bodies ahead of its use, so a wave never sits through an HBM round trip with the SIMD's other wave left alone
    // (0.67 chunks per tile at C3, each ~2 k cycles of exposed latency before).  It travels global -> LDS directly
    // (global_load_lds_dwordx4): through registers the loop-carried values are copied at the loop header, and hipcc waits for
    // the load there, i.e. at once.
    using V4 = typename Vec4<IT>::type;
    constexpr int kHalves = sizeof(V4) / 16; // a slot is one (fp32) or two (fp64) 16-byte pieces
    __shared__ __attribute__((aligned(16))) float4 stage_all[kWaves][kHalves][64];
    auto &stage = stage_all[wid];
    auto in_range = [&](unsigned r_, unsigned c_) { return r_ < B && c_ * 64 + lane < NN; };
    auto request = [&](unsigned r_, unsigned c_) { // global -> LDS without passing through registers: nothing to keep live
        if (in_range(r_, c_)) {
            const char *src = (const char *)(nlist + ((size_t)r_ * NN + c_ * 64 + lane));
#pragma unroll
            for (int q = 0; q < kHalves; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 16 * q),
                                                 (__attribute__((address_space(3))) void *)&stage[q][0], 16, 0, 2 /* nt: load_stream, htf_common.h */);
        }
    };
    request(row, chunk);
    while (true) {
        if (count < 32u && row < B) {
            // ---- stage 64 slots of the current row; request the chunk after it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float sx = 0.f, sy = 0.f, sz = 0.f;
            if (in_range(row, chunk)) {
                if constexpr (kHalves == 1) {
                    const float4 v = stage[0][lane];
                    sx = v.x; sy = v.y; sz = v.z;
                } else {
                    const double2 xy = reinterpret_cast<const double2 *>(&stage[0][0])[lane];
                    const double2 zw = reinterpret_cast<const double2 *>(&stage[1][0])[lane];
                    sx = (float)xy.x; sy = (float)xy.y; sz = (float)zw.x;
                }
            }
            const unsigned cur = row;
            const bool last = ++chunk == nchunks;
            if (last) {
                chunk = 0;
                row += nwaves;
            }
            const float ax = sx + kNormDelta, ay = sy + kNormDelta, az = sz + kNormDelta;
            const bool live = sqrtf(ax * ax + ay * ay + az * az) > kRinvDelta;
            const unsigned long long mask = __ballot(live);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the staged values are in registers: the buffer may be overwritten
            request(row, chunk);
            if (mask != 0ull) {
                row_any = true;
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (live) {
                    unsigned at = head + count + rank;
                    at = at >= 2 * kRing ? at - 2 * kRing : (at >= kRing ? at - kRing : at);
                    ring[0][at] = sx;
                    ring[1][at] = sy;
                    ring[2][at] = sz;
                    ring[3][at] = __uint_as_float(cur);
                }
                count += (unsigned)__builtin_popcountll(mask);
            }
            if (last) {
                if (!row_any) write_row(cur, 0.f, 0.f, 0.f, 0.f, zero6);
                row_any = false;
            }
            asm volatile("" ::: "memory");
            HTF_ESTAMP(0); // staging
            continue;
        }
        if (count == 0u) break;
        if constexpr (kWaves >= 8) {
            // Keep pace with the SIMD's other wave.  Between two waves of equal priority the issue arbiter prefers the OLDER
            // one: it runs at the speed it would have alone (9.4 k cycles per tile, latency-bound) and the younger one gets
            // what is left (17 k) -- with equal shares of the rows the older wave finished at 0.70 of the kernel and the
            // younger one ran the rest alone, a third slower per tile than the pair.  Each wave publishes its tile count; the
            // one behind takes the higher priority for its next tile.  Timing only: what a wave computes does not change.
            // (Waves w and w + 4 of a workgroup sit on the same SIMD -- read back from HW_ID with -DHTF_EVAL_STAMPS; were a
            //  dispatcher to place them otherwise, the rule would pace the wrong pairs and cost nothing but its own effect.)
            tiles_done = __builtin_amdgcn_readfirstlane(tiles_done + 1u);
            // (inline asm: a compiler-visible LDS write would be ordered behind the chunk that is in flight to LDS)
            const unsigned pr0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)&progress[0];
            const unsigned w1 = wid + 4u >= (unsigned)kWaves ? wid + 4u - kWaves : wid + 4u;
            const unsigned w2 = wid + 8u >= (unsigned)kWaves ? wid + 8u - kWaves : wid + 8u;
            unsigned seen, seen2;
            asm volatile("ds_write_b32 %2, %3\n\tds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(seen), "=&v"(seen2)
                         : "v"(pr0 + 4u * wid), "v"(tiles_done), "v"(pr0 + 4u * w1), "v"(pr0 + 4u * w2)
                         : "memory");
            const unsigned o1 = __builtin_amdgcn_readfirstlane(seen), o2 = __builtin_amdgcn_readfirstlane(seen2);
            const unsigned ahead = (tiles_done < o1 ? 1u : 0u) + (kWaves == 12 && tiles_done < o2 ? 1u : 0u); // partners ahead of me
            if (ahead == 0u) // (scalar: s_setprio is not predicated by exec)
                __builtin_amdgcn_s_setprio(0);
            else if (ahead == 1u)
                __builtin_amdgcn_s_setprio(1);
            else
                __builtin_amdgcn_s_setprio(2);
        }
        // ---- pop a tile: 32 pairs (the wave's last one may be partial: its empty lanes are padding)
        const unsigned avail = count < 32u ? count : 32u;
        const bool have = p < avail;
        unsigned at = head + p;
        at = at >= kRing ? at - kRing : at;
        const float x = have ? ring[0][at] : 0.f, y = have ? ring[1][at] : 0.f, z = have ? ring[2][at] : 0.f;
        const unsigned my_row = have ? __float_as_uint(ring[3][at]) : 0xFFFFFFFFu;
        head += avail;
        head = head >= kRing ? head - kRing : head;
        count -= avail;
        asm volatile("" ::: "memory");
        {
            const float tx = x + kNormDelta, ty = y + kNormDelta, tz = z + kNormDelta;
            const float r = have ? sqrtf(tx * tx + ty * ty + tz * tz) : 1.0f; // (an empty lane: any finite distance)
            const bool m = have;

#ifdef HTF_MLP_VALU_PAD // experiment: HTF_MLP_VALU_PAD dummy VALU instructions per 32-slot tile (does the instruction count bind?)
            {
                float pad = r;
#pragma unroll
                for (int i = 0; i < HTF_MLP_VALU_PAD; ++i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(pad));
            }
#endif
            // RBF expansion: lane (p, h) evaluates centres k = f0(v) + 4h, v = 0..15
            f32x16 phi;
#pragma unroll
            for (int v = 0; v < 16; v += 2) { // exp(-(r - c)^2 / gap) = exp2(d^2 * (-log2(e) / gap))
                if constexpr (PK) {
                    const f32x2 d = f32x2{r, r} - f32x2{cen[v], cen[v + 1]};
                    const f32x2 e = pk_exp2((d * d) * nginv_l2e);
                    phi[v] = e[0];
                    phi[v + 1] = e[1];
                } else {
                    const float d0 = r - cen[v], d1 = r - cen[v + 1];
                    phi[v] = __builtin_amdgcn_exp2f((d0 * d0) * nginv_l2e);
                    phi[v + 1] = __builtin_amdgcn_exp2f((d1 * d1) * nginv_l2e);
                }
            }

            HTF_ESTAMP(1); // pop + RBF
            // The chain phi -> L1 -> act -> L2 -> act -> L3 -> backward 2 -> backward 1 is software-pipelined by
            // hand: while the matrix pipe works on one 32-feature block, the wave's VALU turns the PREVIOUS block
            // into the next operand (activation, derivative, operand split).  hipcc clusters each kind of work
            // into its own phase otherwise, and with two or three waves per SIMD the phases of different waves
            // rarely complement each other.  HTF_PIPE pins the interleave (one MFMA, then `per` VALU
            // instructions, `n` times) inside a region closed by a full scheduling barrier.
            // (split16 since round 4: with the SIMD's two waves keeping pace -- see the priority rule above -- hipcc's own order
            //  inside a block is as good: 1.101 ms against 1.109 with the pinned interleave on tools/mlp_ab.py; the regions stay
            //  for the bf16-typed precisions)
#define HTF_PIPE(n, per)                                                                                               \
    if constexpr (!PK && P != HTF_MLP_SPLIT16) {                                                                                               \
        _Pragma("unroll") for (int q_ = 0; q_ < (n); ++q_) {                                                           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
            __builtin_amdgcn_sched_group_barrier(0x002, (per), 0);                                                     \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
            constexpr int kM = P == HTF_MLP_FP32 ? 16 : (P == HTF_MLP_BF16 ? 2 : (P == HTF_MLP_SPLIT ? 12 : 6)); // MFMAs per block
            constexpr int kAct = TANH ? 56 : 0;
            // (split16: 16 v_cvt_pk; its 16 v_fma_mix are inline asm, which the scheduler places by their dependences)
            constexpr int kPrep = P == HTF_MLP_SPLIT ? 76 : (P == HTF_MLP_BF16 ? 8 : (P == HTF_MLP_SPLIT16 ? 16 : 0));
            constexpr int kBwd = TANH ? 16 : 0, kDot = 8;

            f32x16 a1[2], a2[2], dphi;
            float up2[2] = {0.f, 0.f};
            if constexpr (PK) {
                // fp32 MFMA: nothing runs beside it, so the plain layer-by-layer order with the fewest live registers
                // (three waves per SIMD at 168 VGPRs) and packed arithmetic
                {
                    const BOp<P> phi_b = prep<P>(phi);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        a1[nb] = load_tab(lds + I::TabB1, nb, h);
                        mfma_blk<P>(a1[nb], lds + I::L1 + nb * I::BS, lane, phi_b);
                        act_tile<TANH, PK>(a1[nb]);
                    }
                }
                {
                    const BOp<P> a1_b[2] = {prep<P>(a1[0]), prep<P>(a1[1])};
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        a2[nb] = load_tab(lds + I::TabB2, nb, h);
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
                            mfma_blk<P>(a2[nb], lds + I::L2 + (nb * 2 + kb) * I::BS, lane, a1_b[kb]);
                        act_tile<TANH, PK>(a2[nb]);
                    }
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const f32x16 w3 = load_tab(lds + I::TabW3, b, h);
                    dot_tile<PK>(up2, a2[b], w3);
                    act_bwd_tile<TANH, PK>(a2[b], w3);
                }
                {
                    const BOp<P> dz2_b[2] = {prep<P>(a2[0]), prep<P>(a2[1])};
#pragma unroll
                    for (int fb = 0; fb < 2; ++fb) {
                        f32x16 d1;
#pragma unroll
                        for (int v = 0; v < 16; ++v) d1[v] = 0.f;
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
                            mfma_blk<P>(d1, lds + I::B2 + (fb * 2 + kb) * I::BS, lane, dz2_b[kb]);
                        act_bwd_tile<TANH, PK>(a1[fb], d1);
                    }
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) dphi[v] = 0.f;
                {
                    const BOp<P> dz1_b[2] = {prep<P>(a1[0]), prep<P>(a1[1])};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) mfma_blk<P>(dphi, lds + I::B1 + kb * I::BS, lane, dz1_b[kb]);
                }
            } else {
            // ---- layer 1: a1^T[f][p] = b1 + W1^T phi^T
            const BOp<P> phi_b = prep<P>(phi);
            a1[0] = load_tab(lds + I::TabB1, 0, h);
            mfma_blk<P>(a1[0], lds + I::L1, lane, phi_b);
            if constexpr (!PK) __builtin_amdgcn_sched_barrier(0);
            a1[1] = load_tab(lds + I::TabB1, 1, h);
            mfma_blk<P>(a1[1], lds + I::L1 + I::BS, lane, phi_b);
            act_tile<TANH, PK, kU>(a1[0]);
            const BOp<P> a1_b0 = prep<P>(a1[0]);
            HTF_PIPE(kM, pipe_per(kAct + kPrep, kM));
            HTF_ESTAMP(2);
            // ---- layer 2
            a2[0] = load_tab(lds + I::TabB2, 0, h);
            a2[1] = load_tab(lds + I::TabB2, 1, h);
            mfma_blk<P>(a2[0], lds + I::L2 + (0 * 2 + 0) * I::BS, lane, a1_b0);
            mfma_blk<P>(a2[1], lds + I::L2 + (1 * 2 + 0) * I::BS, lane, a1_b0);
            act_tile<TANH, PK, kU>(a1[1]);
            const BOp<P> a1_b1 = prep<P>(a1[1]);
            HTF_PIPE(2 * kM, pipe_per(kAct + kPrep, 2 * kM));
            HTF_ESTAMP(3);
            mfma_blk<P>(a2[0], lds + I::L2 + (0 * 2 + 1) * I::BS, lane, a1_b1);
            if constexpr (!PK) __builtin_amdgcn_sched_barrier(0);
            // ---- layer 3 (dot with w3) and dz2 = w3 * act'(z2), in place, block 0 under the last L2 block
            mfma_blk<P>(a2[1], lds + I::L2 + (1 * 2 + 1) * I::BS, lane, a1_b1);
            act_tile<TANH, PK, kU>(a2[0]);
            {
                const f32x16 w3 = load_tab(lds + I::TabW3, 0, h);
                dot_tile<PK>(up2, a2[0], w3);
                act_bwd_tile<TANH, PK, kU>(a2[0], w3);
            }
            const BOp<P> dz2_b0 = prep<P>(a2[0]);
            HTF_PIPE(kM, pipe_per(kAct + kDot + kBwd + kPrep, kM));
            HTF_ESTAMP(4);
            // ---- backward 2: dh1^T = W2 dz2^T, then dz1 = dh1 * act'(z1) into a1
            f32x16 d1[2];
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int v = 0; v < 16; ++v) d1[fb][v] = 0.f;
            mfma_blk<P>(d1[0], lds + I::B2 + (0 * 2 + 0) * I::BS, lane, dz2_b0);
            mfma_blk<P>(d1[1], lds + I::B2 + (1 * 2 + 0) * I::BS, lane, dz2_b0);
            act_tile<TANH, PK, kU>(a2[1]);
            {
                const f32x16 w3 = load_tab(lds + I::TabW3, 1, h);
                dot_tile<PK>(up2, a2[1], w3);
                act_bwd_tile<TANH, PK, kU>(a2[1], w3);
            }
            const BOp<P> dz2_b1 = prep<P>(a2[1]);
            HTF_PIPE(2 * kM, pipe_per(kAct + kDot + kBwd + kPrep, 2 * kM));
            HTF_ESTAMP(5);
            mfma_blk<P>(d1[0], lds + I::B2 + (0 * 2 + 1) * I::BS, lane, dz2_b1);
            if constexpr (!PK) __builtin_amdgcn_sched_barrier(0);
            mfma_blk<P>(d1[1], lds + I::B2 + (1 * 2 + 1) * I::BS, lane, dz2_b1);
            act_bwd_tile<TANH, PK, kU>(a1[0], d1[0]);
            const BOp<P> dz1_b0 = prep<P>(a1[0]);
            HTF_PIPE(kM, pipe_per(kBwd + kPrep, kM));
            HTF_ESTAMP(6);
            // ---- backward 1: dphi^T = W1 dz1^T
#pragma unroll
            for (int v = 0; v < 16; ++v) dphi[v] = 0.f;
            mfma_blk<P>(dphi, lds + I::B1, lane, dz1_b0);
            act_bwd_tile<TANH, PK, kU>(a1[1], d1[1]);
            const BOp<P> dz1_b1 = prep<P>(a1[1]);
            HTF_PIPE(kM, pipe_per(kBwd + kPrep, kM));
            mfma_blk<P>(dphi, lds + I::B1 + I::BS, lane, dz1_b1);
            if constexpr (!PK) __builtin_amdgcn_sched_barrier(0);
            HTF_ESTAMP(7);
            }
#undef HTF_PIPE
            const float upart = up2[0] + up2[1];
            const float u = sum_xor32(upart) + b3;

            // du/dr = sum_k dphi_k * (-2 (r - c_k) / gap) * phi_k
            float dp2[2] = {0.f, 0.f};
            const float m2ginv = -2.0f * ginv;
#pragma unroll
            for (int v = 0; v < 16; v += 2) {
                if constexpr (PK) {
                    const f32x2 d = f32x2{r, r} - f32x2{cen[v], cen[v + 1]};
                    const f32x2 t = pk_fma((f32x2{dphi[v], dphi[v + 1]} * f32x2{phi[v], phi[v + 1]}) * m2ginv, d, f32x2{dp2[0], dp2[1]});
                    dp2[0] = t[0];
                    dp2[1] = t[1];
                } else {
                    dp2[0] = fmaf((dphi[v] * phi[v]) * m2ginv, r - cen[v], dp2[0]);
                    dp2[1] = fmaf((dphi[v + 1] * phi[v + 1]) * m2ginv, r - cen[v + 1], dp2[1]);
                }
            }
            const float dpart = dp2[0] + dp2[1];
            const float dudr = sum_xor32(dpart);

            // E_i += 1/2 u ; F_i += 2 * (1/2) du/dr * t / r   (upper half duplicates the lower one's pairs)
            const bool mine = m && h == 0;
            const float c = dudr / r;
            const float px = mine ? c * tx : 0.f, py = mine ? c * ty : 0.f, pz = mine ? c * tz : 0.f, pe = mine ? 0.5f * u : 0.f;
            Virial6 vir; // _compute_virial (simmodel.py:509-523) works for any energy: -(|nf| / (2 |x|)) x (x) x per slot
            if constexpr (VIRIAL)
                if (mine) vir.add(x, y, z, c * tx, c * ty, c * tz);
            // one masked wave sum per row present in the tile, in row order; a row is written when the next one shows up
            unsigned long long todo = __ballot(mine);
            while (todo != 0ull) {
                const unsigned first = (unsigned)__builtin_ctzll(todo);
                const unsigned rid = (unsigned)__builtin_amdgcn_readlane((int)my_row, (int)first);
                const bool sel = mine && my_row == rid;
                const float tot = wave_sum4(sel ? px : 0.f, sel ? py : 0.f, sel ? pz : 0.f, sel ? pe : 0.f); // rows 0..3: x, z, y, e
                const float sx = bcast_lane(tot, 0), sz = bcast_lane(tot, 16), sy = bcast_lane(tot, 32), se = bcast_lane(tot, 48);
                if (rid != open_row) {
                    if (open_row != 0xFFFFFFFFu) write_row(open_row, cfx, cfy, cfz, cfe, cv6);
                    open_row = rid;
                    cfx = cfy = cfz = cfe = 0.f;
#pragma unroll
                    for (int i = 0; i < 6; ++i) cv6[i] = 0.f;
                }
                cfx += sx; cfy += sy; cfz += sz; cfe += se;
                if constexpr (VIRIAL) {
                    const float t1 = wave_sum4(sel ? vir.xx : 0.f, sel ? vir.xy : 0.f, sel ? vir.xz : 0.f, sel ? vir.yy : 0.f);
                    const float t2 = wave_sum4(sel ? vir.yz : 0.f, sel ? vir.zz : 0.f, 0.f, 0.f);
                    cv6[0] += bcast_lane(t1, 0); cv6[2] += bcast_lane(t1, 16); cv6[1] += bcast_lane(t1, 32); cv6[3] += bcast_lane(t1, 48);
                    cv6[4] += bcast_lane(t2, 0); cv6[5] += bcast_lane(t2, 32);
                }
                todo &= ~__ballot(sel);
            }
            HTF_ESTAMP(8); // du/dr, tile sums, row carries
#ifdef HTF_EVAL_STAMPS
            ++n_tiles;
#endif
        }
    }
#ifdef HTF_EVAL_STAMPS
    if (blockIdx.x == 7 && threadIdx.x == 64) {
        printf("stamps (cycles per tile, wave 1 of block 7, %llu tiles):", n_tiles);
        for (int i = 0; i <= 8; ++i) printf(" %d:%llu", i, stamps[i] / (n_tiles ? n_tiles : 1));
        printf("\n");
    }
    if (blockIdx.x == 9 && lane == 0)
        printf("block 9 wave %u: SIMD %u, %llu tiles, %llu ticks\n", wid, __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4), n_tiles,
               (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin));
    if ((blockIdx.x % 73 == 7 || blockIdx.x == gridDim.x - 1) && threadIdx.x == 64) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime(), r_end = __builtin_amdgcn_s_memrealtime();
        printf("block %u: %llu tiles, memtime %llu ticks, realtime (100 MHz) start %llu + %llu\n", blockIdx.x, n_tiles, t_end - t_begin,
               r_begin % 100000000ull, r_end - r_begin);
    }
#endif
    if (open_row != 0xFFFFFFFFu) write_row(open_row, cfx, cfy, cfz, cfe, cv6);
}

// ------------------------------------------------------------------------------ host side
// theta index feeding each image element (see pair_mlp.h): mirrors the operand order the
// kernel reads -- accumulator register r of lane half hh holds feature f0(r) + 4 hh.
template <bool BF16>
static void build_map(const MlpDevice *m, std::vector<int> &map) {
    using I = Img<BF16>;
    const int K = m->K, H1 = m->H1, H2 = m->H2;
    auto W1 = [&](int k, int f) { return (k < K && f < H1) ? k * H1 + f : -1; };
    auto W2 = [&](int a, int b) { return (a < H1 && b < H2) ? m->off_W2() + a * H2 + b : -1; };
    map.assign(kMapN, -1);
    // element of a block: k-step-local index r (fp32: 16 steps of 1; bf16: 2 steps of 8) and lane
    auto put = [&](int block_off, int r, int lane, int idx) {
        if constexpr (BF16)
            map[(size_t)block_off * 2 + (((r >> 3) * 64 + lane) * 8 + (r & 7))] = idx;
        else
            map[block_off + ((r >> 2) * 64 + lane) * 4 + (r & 3)] = idx;
    };
    for (int r = 0; r < 16; ++r)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31, hh = lane >> 5, kk = f0(r) + 4 * hh;
            for (int nb = 0; nb < 2; ++nb) put(I::L1 + nb * I::BS, r, lane, W1(kk, 32 * nb + i));
            for (int kb = 0; kb < 2; ++kb) put(I::B1 + kb * I::BS, r, lane, W1(i, 32 * kb + kk));
            for (int nb = 0; nb < 2; ++nb)
                for (int kb = 0; kb < 2; ++kb) {
                    put(I::L2 + (nb * 2 + kb) * I::BS, r, lane, W2(32 * kb + kk, 32 * nb + i));
                    put(I::B2 + (nb * 2 + kb) * I::BS, r, lane, W2(32 * nb + i, 32 * kb + kk));
                }
        }
    for (int b = 0; b < 2; ++b)
        for (int hh = 0; hh < 2; ++hh)
            for (int v = 0; v < 16; ++v) {
                const int f = 32 * b + f0(v) + 4 * hh, o = (b * 2 + hh) * 16 + v;
                map[kMapW + o] = f < H1 ? m->off_b1() + f : -1;
                map[kMapW + 64 + o] = f < H2 ? m->off_b2() + f : -1;
                map[kMapW + 128 + o] = f < H2 ? m->off_W3() + f : -1;
            }
    map[kMapW + kMapT] = m->off_b3();
}

// images <- theta (device side, so a training step never visits the host)
// fwd_scale: factor on the FORWARD operand blocks (L1, L2: the first six) and on the two bias tables --
// 2 log2(e) for a tanh network evaluated from bf16-typed images (see act_tile), 1 otherwise.  The
// backward blocks (B2, B1), w3 and b3 are never scaled.
// U: the u-form image set of the split16 tanh evaluator (act_tile): L2 x -2, bias table 2 <- b2 + column sums of W2, w3 x -2,
// b3 <- b3 + sum of w3, B2 x -2, B1 x 4 (L1 and bias table 1 as they are).
template <int P, bool U = false>
__global__ void mlp_refresh_kernel(float *__restrict__ images, const int *__restrict__ map,
                                   const float *__restrict__ theta, float fwd_scale, int *__restrict__ range_flag, MlpDims dm = MlpDims{}) {
    using I = Img<P>;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= kMapN) return;
    const int idx = map[e];
    const bool fwd = e < 6 * 1024 || (e >= kMapW && e < kMapW + 128);
    float v = (idx >= 0 ? theta[idx] : 0.f) * (fwd ? fwd_scale : 1.0f);
    if constexpr (U) {
        const int blk = e >> 10;
        if (e < kMapW) {
            v *= blk < 2 ? 1.0f : (blk < 10 ? -2.0f : 4.0f);
        } else if (e >= kMapW + 64 && e < kMapW + 128) {
            if (idx >= 0) {
                const int f = idx - dm.oB2;
                float sum = 0.f;
                for (int a = 0; a < dm.H1; ++a) sum += theta[dm.oW2 + a * dm.H2 + f];
                v += fwd_scale * sum;
            }
        } else if (e >= kMapW + 128 && e < kMapW + 192) {
            v *= -2.0f;
        } else if (e >= kMapW + kMapT) {
            float sum = 0.f;
            for (int f = 0; f < dm.H2; ++f) sum += theta[dm.oW3 + f];
            v += sum;
        }
    }
    if (e < kMapW) {
        if constexpr (P == HTF_MLP_BF16) { // round to nearest even (finite weights)
            const unsigned u = __float_as_uint(v);
            reinterpret_cast<unsigned short *>(images)[e] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
        } else if constexpr (P == HTF_MLP_SPLIT16) { // hi = fp16(v), lo = fp16(v - hi), round to nearest: see prep<>
            // a TRAINED weight (device parameter vector: mlp_create cannot see it) that left fp16's range, or is not a number:
            // read by the host at the next evaluation / training call, judged afresh by every image build
            if (range_flag != nullptr && !(fabsf(v) < 6.0e4f)) *range_flag = 1;
            const _Float16 hh = (_Float16)v;
            const _Float16 ll = (_Float16)(v - (float)hh);
            unsigned short *blk = reinterpret_cast<unsigned short *>(images) + (size_t)(e >> 10) * 2048 + (e & 1023);
            blk[0] = __builtin_bit_cast(unsigned short, hh);
            blk[1024] = __builtin_bit_cast(unsigned short, ll);
        } else if constexpr (P == HTF_MLP_SPLIT) { // exact three-way split, see prep<>
            const unsigned u = __float_as_uint(v);
            const float r1 = v - __uint_as_float(u & 0xFFFF0000u);
            const unsigned u1 = __float_as_uint(r1);
            const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);
            unsigned short *blk = reinterpret_cast<unsigned short *>(images) + (size_t)(e >> 10) * 3072 + (e & 1023);
            blk[0] = (unsigned short)(u >> 16);
            blk[1024] = (unsigned short)(u1 >> 16);
            blk[2048] = (unsigned short)(__float_as_uint(r2) >> 16);
        } else {
            images[e] = v;
        }
    } else if (e < kMapW + kMapT) {
        images[I::TabB1 + (e - kMapW)] = v;
    } else {
        images[I::TabB3] = v;
    }
}

__global__ void mlp_flag_clear_kernel(int *__restrict__ range_flag) { *range_flag = 0; }

// (read without a synchronisation: a violation written by an image build that has not run yet is seen by the call after)
bool mlp_out_of_range(const MlpDevice *m) { return m->range_flag != nullptr && *(volatile int *)m->range_flag != 0; }

int mlp_refresh(const MlpDevice *m, hipStream_t stream) {
    HTF_REQUIRE(m, "pair-MLP: null potential");
    const unsigned grid = (kMapN + 255) / 256;
    const float fwd_scale = m->act == HTF_ACT_TANH ? 2.8853900817779268f : 1.0f; // bf16-typed images only
    if (m->precision == HTF_MLP_BF16)
        hipLaunchKernelGGL(mlp_refresh_kernel<HTF_MLP_BF16>, dim3(grid), dim3(256), 0, stream, m->images, m->map, m->theta, fwd_scale, (int *)nullptr, MlpDims{});
    else if (m->precision == HTF_MLP_SPLIT)
        hipLaunchKernelGGL(mlp_refresh_kernel<HTF_MLP_SPLIT>, dim3(grid), dim3(256), 0, stream, m->images, m->map, m->theta, fwd_scale, (int *)nullptr, MlpDims{});
    else if (m->precision == HTF_MLP_SPLIT16) {
        // every image build judges the range afresh (the word used to be sticky -- one bad weight, and every later call
        // failed even after the caller had repaired theta and refreshed): cleared in stream order ahead of the build that may set it
        if (m->range_flag != nullptr) hipLaunchKernelGGL(mlp_flag_clear_kernel, dim3(1), dim3(1), 0, stream, m->range_flag);
        hipLaunchKernelGGL(mlp_refresh_kernel<HTF_MLP_SPLIT16>, dim3(grid), dim3(256), 0, stream, m->images, m->map, m->theta, fwd_scale, m->range_flag, MlpDims{});
        if (m->eval_images != m->images) { // the evaluator's u-form set (tanh): see act_tile
            const MlpDims dm{m->K, m->H1, m->H2, m->off_b1(), m->off_W2(), m->off_b2(), m->off_W3(), m->off_b3()};
            hipLaunchKernelGGL((mlp_refresh_kernel<HTF_MLP_SPLIT16, true>), dim3(grid), dim3(256), 0, stream, m->eval_images, m->map, m->theta, fwd_scale,
                               m->range_flag, dm);
        }
    } else
        hipLaunchKernelGGL(mlp_refresh_kernel<HTF_MLP_FP32>, dim3(grid), dim3(256), 0, stream, m->images, m->map, m->theta, 1.0f, (int *)nullptr, MlpDims{});
    if (m->train_images != m->images) // bf16 / split evaluator images: the training sweep reads its own fp32 set
        hipLaunchKernelGGL(mlp_refresh_kernel<HTF_MLP_FP32>, dim3(grid), dim3(256), 0, stream, m->train_images, m->train_map, m->theta, 1.0f, (int *)nullptr, MlpDims{});
    return check_launch("mlp_refresh_kernel");
}

int mlp_create(const htf_potential_desc *d, MlpDevice **out) {
    HTF_REQUIRE(d->d_theta || (d->W1 && d->b1 && d->W2 && d->b2 && d->W3 && d->b3), "pair-MLP: null weight pointer");
    HTF_REQUIRE(d->K >= 2 && d->K <= kK, "pair-MLP: K=%d outside [2, %d]", d->K, kK);
    HTF_REQUIRE(d->H1 >= 1 && d->H1 <= kH && d->H2 >= 1 && d->H2 <= kH, "pair-MLP: hidden widths (%d, %d) must be <= %d", d->H1, d->H2, kH);
    HTF_REQUIRE(d->rbf_high > d->rbf_low, "pair-MLP: rbf_high must exceed rbf_low");
    HTF_REQUIRE(d->activation == HTF_ACT_LINEAR || d->activation == HTF_ACT_TANH, "pair-MLP: unknown activation %d", d->activation);
    HTF_REQUIRE(d->mlp_precision == HTF_MLP_FP32 || d->mlp_precision == HTF_MLP_BF16 || d->mlp_precision == HTF_MLP_SPLIT ||
                    d->mlp_precision == HTF_MLP_SPLIT16, "pair-MLP: unknown precision %d", d->mlp_precision);
    if (d->mlp_precision == HTF_MLP_SPLIT16 && !d->d_theta) {
        // fp16 operands: host-supplied weights are checked against the format's range here (times the 2 log2(e) folded into
        // the forward images); a device parameter vector (training) is the caller's to keep there
        float wmax = 0.f;
        for (int i = 0; i < d->K * d->H1; ++i) wmax = fmaxf(wmax, fabsf(d->W1[i]));
        for (int i = 0; i < d->H1 * d->H2; ++i) wmax = fmaxf(wmax, fabsf(d->W2[i]));
        for (int i = 0; i < d->H1; ++i) wmax = fmaxf(wmax, fabsf(d->b1[i]));
        for (int i = 0; i < d->H2; ++i) wmax = fmaxf(wmax, fabsf(d->b2[i]));
        HTF_REQUIRE(wmax * 2.8853900817779268f < 6.0e4f, "pair-MLP: weights up to %g leave fp16's range; use precision 'split' (bf16 parts) or 'fp32'", wmax);
        HTF_REQUIRE(d->activation == HTF_ACT_TANH || wmax * (float)(d->K > d->H1 ? d->K : d->H1) < 6.0e4f,
                    "pair-MLP: a linear network with weights up to %g can leave fp16's range; use precision 'split' or 'fp32'", wmax);
    }
    MlpDevice *m = new (std::nothrow) MlpDevice();
    if (!m) {
        set_error("pair-MLP: out of host memory");
        return HTF_ERR_NOMEM;
    }
    const bool bf16 = d->mlp_precision != HTF_MLP_FP32; // bf16-typed images (one part, or three for the split)
    const bool split = d->mlp_precision == HTF_MLP_SPLIT;
    const bool split16 = d->mlp_precision == HTF_MLP_SPLIT16;
    m->K = d->K; m->H1 = d->H1; m->H2 = d->H2;
    m->act = d->activation;
    m->precision = d->mlp_precision;
    // RBF centres: float32 linspace, gap = c[1] - c[0]  (layers.py:31-34)
    for (int k = 0; k < kK; ++k) m->centers[k] = 0.f;
    for (int k = 0; k < m->K; ++k) {
        double step = (d->rbf_high - d->rbf_low) / (double)(m->K - 1);
        m->centers[k] = (float)(k == m->K - 1 ? d->rbf_high : d->rbf_low + k * step);
    }
    m->gap = m->centers[1] - m->centers[0];
    const int n_img = split ? Img<2>::Floats : (split16 ? Img<3>::Floats : (bf16 ? Img<1>::Floats : Img<0>::Floats));
    const int tabc = split ? Img<2>::TabC : (split16 ? Img<3>::TabC : (bf16 ? Img<1>::TabC : Img<0>::TabC));
    std::vector<float> img(n_img, 0.f);
    for (int hh = 0; hh < 2; ++hh)
        for (int v = 0; v < 16; ++v) img[tabc + hh * 16 + v] = m->centers[f0(v) + 4 * hh];
    std::vector<int> map;
    if (bf16) build_map<true>(m, map); else build_map<false>(m, map);
    const int P = m->num_params();
    hipError_t e = hipMalloc((void **)&m->images, img.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(m->images, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&m->map, map.size() * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(m->map, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice);
    m->train_images = m->images;
    m->train_map = m->map;
    m->eval_images = m->images;
    if (e == hipSuccess && split16 && m->act == HTF_ACT_TANH && HTF_MLP_UFORM) {
        m->eval_images = nullptr;
        e = hipMalloc((void **)&m->eval_images, img.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(m->eval_images, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice); // (the RBF centres)
    }
    if (e == hipSuccess && bf16) {
        std::vector<float> img32(Img<false>::Floats, 0.f);
        for (int hh = 0; hh < 2; ++hh)
            for (int v = 0; v < 16; ++v) img32[Img<false>::TabC + hh * 16 + v] = m->centers[f0(v) + 4 * hh];
        std::vector<int> map32;
        build_map<false>(m, map32);
        m->train_images = nullptr;
        m->train_map = nullptr;
        e = hipMalloc((void **)&m->train_images, img32.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(m->train_images, img32.data(), img32.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void **)&m->train_map, map32.size() * sizeof(int));
        if (e == hipSuccess) e = hipMemcpy(m->train_map, map32.data(), map32.size() * sizeof(int), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && d->d_theta) {
        m->theta = d->d_theta; // caller-owned, trainable: htf_potential_refresh after every update
    } else if (e == hipSuccess) {
        std::vector<float> th((size_t)P);
        std::memcpy(&th[0], d->W1, sizeof(float) * m->K * m->H1);
        std::memcpy(&th[m->off_b1()], d->b1, sizeof(float) * m->H1);
        std::memcpy(&th[m->off_W2()], d->W2, sizeof(float) * m->H1 * m->H2);
        std::memcpy(&th[m->off_b2()], d->b2, sizeof(float) * m->H2);
        std::memcpy(&th[m->off_W3()], d->W3, sizeof(float) * m->H2);
        th[m->off_b3()] = d->b3[0];
        e = hipMalloc((void **)&m->own_theta, th.size() * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(m->own_theta, th.data(), th.size() * sizeof(float), hipMemcpyHostToDevice);
        m->theta = m->own_theta;
    }
    if (e == hipSuccess && split16) { // host-mapped: the image build writes it, the host reads it without a synchronisation
        e = hipHostMalloc((void **)&m->range_flag, sizeof(int), hipHostMallocMapped);
        if (e == hipSuccess) *m->range_flag = 0;
    }
    if (e != hipSuccess) {
        set_error("pair-MLP: device upload failed: %s", hipGetErrorString(e));
        mlp_destroy(m);
        return HTF_ERR_DEVICE;
    }
    int rc = mlp_refresh(m, nullptr);
    if (rc == HTF_OK && hipStreamSynchronize(nullptr) != hipSuccess) {
        set_error("pair-MLP: image build failed");
        rc = HTF_ERR_DEVICE;
    }
    if (rc != HTF_OK) {
        mlp_destroy(m);
        return rc;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        m->n_cu = prop.multiProcessorCount;
    *out = m;
    return HTF_OK;
}

void mlp_destroy(MlpDevice *m) {
    if (!m) return;
    if (m->eval_images && m->eval_images != m->images) (void)hipFree(m->eval_images);
    if (m->train_images && m->train_images != m->images) (void)hipFree(m->train_images);
    if (m->train_map && m->train_map != m->map) (void)hipFree(m->train_map);
    if (m->images) (void)hipFree(m->images);
    if (m->map) (void)hipFree(m->map);
    if (m->own_theta) (void)hipFree(m->own_theta);
    if (m->range_flag) (void)hipHostFree(m->range_flag);
    delete m;
}

template <bool TANH, int P>
static int launch_mlp(const MlpDevice *m, const void *nlist, int in_dtype, unsigned B, unsigned NN, void *force,
                      int out_f64, void *virial9, hipStream_t s) {
    // persistent blocks: one of eight waves per CU (MlpLaunch)
    unsigned grid = (unsigned)m->n_cu * (unsigned)MlpLaunch<P>::kPerCU;
    constexpr unsigned kW = (unsigned)MlpLaunch<P>::kWaves;
    unsigned need = (B + kW - 1) / kW;
    if (grid > need) grid = need;
#define HTF_MLP_LAUNCH(T, V4, VIR)                                                                                     \
    hipLaunchKernelGGL((pair_mlp_kernel<TANH, T, P, VIR>), dim3(grid), dim3(64 * kW), 0, s, (const V4 *)nlist, B, NN, force, \
                       out_f64, m->eval_images, m->gap, virial9)
    if (in_dtype == HTF_F32) {
        if (virial9) HTF_MLP_LAUNCH(float, float4, true); else HTF_MLP_LAUNCH(float, float4, false);
    } else {
        if (virial9) HTF_MLP_LAUNCH(double, double4, true); else HTF_MLP_LAUNCH(double, double4, false);
    }
#undef HTF_MLP_LAUNCH
    return check_launch("pair_mlp_kernel");
}

int mlp_eval(const MlpDevice *m, const void *nlist, int in_dtype, unsigned B, unsigned NN, void *force,
             int force_dtype, void *virial9, hipStream_t stream) {
    HTF_REQUIRE(m, "pair-MLP: null potential");
    HTF_REQUIRE(!mlp_out_of_range(m), "pair-MLP: a weight of the device parameter vector left fp16's range (|2.885 w| >= 6e4) or is not a "
                                      "number: precision 'split16' cannot carry it; use 'split' or 'fp32'");
    const int out_f64 = force_dtype == HTF_F64;
    if (m->precision == HTF_MLP_BF16)
        return m->act == HTF_ACT_TANH ? launch_mlp<true, HTF_MLP_BF16>(m, nlist, in_dtype, B, NN, force, out_f64, virial9, stream)
                                      : launch_mlp<false, HTF_MLP_BF16>(m, nlist, in_dtype, B, NN, force, out_f64, virial9, stream);
    if (m->precision == HTF_MLP_SPLIT16)
        return m->act == HTF_ACT_TANH ? launch_mlp<true, HTF_MLP_SPLIT16>(m, nlist, in_dtype, B, NN, force, out_f64, virial9, stream)
                                      : launch_mlp<false, HTF_MLP_SPLIT16>(m, nlist, in_dtype, B, NN, force, out_f64, virial9, stream);
    if (m->precision == HTF_MLP_SPLIT)
        return m->act == HTF_ACT_TANH ? launch_mlp<true, HTF_MLP_SPLIT>(m, nlist, in_dtype, B, NN, force, out_f64, virial9, stream)
                                      : launch_mlp<false, HTF_MLP_SPLIT>(m, nlist, in_dtype, B, NN, force, out_f64, virial9, stream);
    return m->act == HTF_ACT_TANH ? launch_mlp<true, HTF_MLP_FP32>(m, nlist, in_dtype, B, NN, force, out_f64, virial9, stream)
                                  : launch_mlp<false, HTF_MLP_FP32>(m, nlist, in_dtype, B, NN, force, out_f64, virial9, stream);
}

} // namespace htf
