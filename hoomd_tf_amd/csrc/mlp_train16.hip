// Online force matching for split16 pair-MLP potentials (FORCE_MODE::hoomd2tf, tensorflowcompute.py:347-370): the
// loss-gradient sweep of mlp_train.hip -- value chain, r-tangent chain, ONE reverse pass over both -- with every matrix product
// on the fp16 pipeline and fp32-level operands (x = hi + lo in fp16, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, as the
// evaluator pair_mlp.hip HTF_MLP_SPLIT16, whose images it reads: forward blocks and bias tables carry 2 log2(e) for tanh -- the
// accumulator is the exponent -- and the tangent chain is scaled back).
// Compiled -fno-slp-vectorize: packed fp32 instructions do not run beside 16-bit MFMA work on gfx950 (tools/mfma_valu_probe2.hip).
#include "htf_common.h"
#include "htf_internal.h"
#include "pair_mlp.h"

namespace htf {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using u32x4t = __attribute__((ext_vector_type(4))) unsigned;
#define HTF_MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct Op16 {
    f16x8 hi[2], lo[2]; // one 32-feature block: accumulator elements 8s .. 8s+7 are k-step s
};

// x = hi + lo, both fp16 and rounded to nearest: v_cvt_pk_f16_f32 per pair of elements, the residual by v_fma_mix_f32 reading
// hi from its half of the packed register (pair_mlp.hip prep<HTF_MLP_SPLIT16>)
__device__ __forceinline__ Op16 split16(const f32x16 &x) {
    Op16 o;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4t ph, pl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = x[8 * s + 2 * j], b = x[8 * s + 2 * j + 1];
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
    return o;
}

// tanh from an accumulator that already carries 2 log2(e) z (the split16 images fold it into the forward blocks)
template <bool TANH>
__device__ __forceinline__ float act_scaled(float a) {
    if constexpr (!TANH) return a;
    float t = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a) + 1.0f), -2.0f, 1.0f);
    // (an empty asm the value passes through: hipcc otherwise folds this fma into split16()'s conversion as v_fma_mixlo / mixhi_f16
    //  -- 9.4 cycles each at one wave per SIMD, next to the fp32 fma it still needs -- pair_mlp.hip act_tile)
    asm("" : "+v"(t));
    return t;
}

__device__ __forceinline__ void mfma_pair16(f32x16 &acc0, f32x16 &acc1, const float *img, unsigned lane, const Op16 &p0, const Op16 &p1) {
    const f16x8 *p = reinterpret_cast<const f16x8 *>(img) + lane; // [part 2: hi, lo][s 2][lane 64]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f16x8 ah = p[s * 64], al = p[(2 + s) * 64];
        acc0 = HTF_MFMA_H(al, p0.hi[s], acc0);
        acc1 = HTF_MFMA_H(al, p1.hi[s], acc1);
        acc0 = HTF_MFMA_H(ah, p0.lo[s], acc0);
        acc1 = HTF_MFMA_H(ah, p1.lo[s], acc1);
        acc0 = HTF_MFMA_H(ah, p0.hi[s], acc0);
        acc1 = HTF_MFMA_H(ah, p1.hi[s], acc1);
    }
}

// acc[i][n] += sum over the tile's pairs of A[i][pair] B[n][pair], both operands in the F layout (lane = feature, k = pair)
__device__ __forceinline__ void outer16_f16(f32x16 &acc, const Op16 &A, const Op16 &Bm) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        acc = HTF_MFMA_H(A.lo[s], Bm.hi[s], acc);
        acc = HTF_MFMA_H(A.hi[s], Bm.lo[s], acc);
        acc = HTF_MFMA_H(A.hi[s], Bm.hi[s], acc);
    }
}

// ---- a wave-private transpose through LDS (no barrier: one wave writes and reads its own scratch; DS operations of a wave
// complete in issue order).  A 32-feature block of a P-layout quantity (lane = pair, registers = features: an accumulator tile
// or its split form) is stored as two planes (hi, lo) of [pair 32][feature 32] fp16, 64-byte rows, and read back with
// ds_read_b64_tr_b16 as the F-layout MFMA operand (lane = feature, k = pair) of the pair-contracted weight gradients.
// Lane (p, h) owns features 8c + 4h + (0..3), c = 0..3 -- elements 4c..4c+3 of its split form -- i.e. the 8-byte chunk 2c + h of
// its row; chunk positions are XORed with (pair >> 1) & 7, which makes the ds_write_b64 (16 lanes x 32 banks) and the transposed
// reads (32 lanes x 64 banks) conflict-free at once.
constexpr int kTrPlane = 32 * 64;       // bytes: one plane
constexpr int kTrBlock = 2 * kTrPlane;  // hi plane, lo plane
constexpr int kTrBlocks = 6;            // per wave: phi, phid | h1[0], h1[1], hd1[0], hd1[1] (then zb2[0..1], zdb2[0..1] in their place)
using v4h = __fp16 __attribute__((__vector_size__(8)));
using lds_v4h = __attribute__((address_space(3))) v4h;

__device__ __forceinline__ void tr_write(unsigned char *blk, unsigned p, unsigned h, const Op16 &o) {
    unsigned char *row = blk + p * 64u;
    const unsigned sw = (p >> 1) & 7u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned off = 8u * ((2u * c + h) ^ sw);
        const u32x4t hi = __builtin_bit_cast(u32x4t, o.hi[c >> 1]), lo = __builtin_bit_cast(u32x4t, o.lo[c >> 1]);
        *reinterpret_cast<uint2 *>(row + off) = make_uint2(hi[2 * (c & 1)], hi[2 * (c & 1) + 1]);
        *reinterpret_cast<uint2 *>(row + kTrPlane + off) = make_uint2(lo[2 * (c & 1)], lo[2 * (c & 1) + 1]);
    }
}

// The F-layout operand of the block: lane l holds feature l & 31; element j of k-step s of lane half hh is pair
// 16 s + 8 (j >> 2) + 4 hh + (j & 3) = f0(8 s + j) + 4 hh -- the pair order of an F-layout ACCUMULATOR's registers, so operands
// that come from LDS and operands split from an accumulator contract over the same k.  ds_read_b64_tr_b16: lane 4q + pp of a
// 16-lane group addresses columns 4 pp .. + 3 of row q of a 4 x 16 block and lane i receives column i, row q in element q
// (tools/tr16_probe.hip checks the map with exact integers).  EXEC is all ones here (wave-uniform control flow only).
__device__ __forceinline__ Op16 tr_read(const unsigned char *blk, unsigned lane) {
    const unsigned g = lane >> 4, q = (lane >> 2) & 3u, pp = lane & 3u, hh = lane >> 5;
    const unsigned ch = 4u * (g & 1u) + pp;
    Op16 o;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4t wh, wl;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned row = 16u * s + 8u * t + 4u * hh + q;
            const unsigned off = row * 64u + 8u * (ch ^ ((row >> 1) & 7u));
            const v4h a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_v4h *)(blk + off));
            const v4h b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_v4h *)(blk + kTrPlane + off));
            const uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
            wh[2 * t] = ua.x; wh[2 * t + 1] = ua.y;
            wl[2 * t] = ub.x; wl[2 * t + 1] = ub.y;
        }
        o.hi[s] = __builtin_bit_cast(f16x8, wh);
        o.lo[s] = __builtin_bit_cast(f16x8, wl);
    }
    return o;
}

// compiler-level ordering of the scratch accesses of one wave (the hardware keeps a wave's DS operations in order)
__device__ __forceinline__ void tr_fence() { asm volatile("" ::: "memory"); }

// D[pair][feature] instead of D[feature][pair]: the SAME image fragment taken as the B operand and the activations as A
// transposes the product -- the reverse pass through layer 2 lands in the F layout its weight-gradient contraction wants.
__device__ __forceinline__ void mfma_pair16_t(f32x16 &acc0, f32x16 &acc1, const float *img, unsigned lane, const Op16 &p0, const Op16 &p1) {
    const f16x8 *p = reinterpret_cast<const f16x8 *>(img) + lane; // [part 2: hi, lo][s 2][lane 64]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f16x8 ah = p[s * 64], al = p[(2 + s) * 64];
        acc0 = HTF_MFMA_H(p0.hi[s], al, acc0);
        acc1 = HTF_MFMA_H(p1.hi[s], al, acc1);
        acc0 = HTF_MFMA_H(p0.lo[s], ah, acc0);
        acc1 = HTF_MFMA_H(p1.lo[s], ah, acc1);
        acc0 = HTF_MFMA_H(p0.hi[s], ah, acc0);
        acc1 = HTF_MFMA_H(p1.hi[s], ah, acc1);
    }
}

// element e (0..15 = 8 s + j) of an operand as fp32: hi + lo, one v_fma_mix_f32 reading both halves from their packed registers
__device__ __forceinline__ float op16_elem(const Op16 &o, int e) {
    const u32x4t ph = __builtin_bit_cast(u32x4t, o.hi[e >> 3]), pl = __builtin_bit_cast(u32x4t, o.lo[e >> 3]);
    const unsigned a = ph[(e & 7) >> 1], b = pl[(e & 7) >> 1];
    float r;
    if (e & 1)
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
    else
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__global__ void mlp_resid_max_kernel(const float4 *__restrict__ pred, const void *__restrict__ labels, int lab_f64, unsigned B,
                                     unsigned *__restrict__ out_bits) {
    float m = 0.f;
    for (unsigned row = blockIdx.x * blockDim.x + threadIdx.x; row < B; row += gridDim.x * blockDim.x) {
        const float4 rs = residual(pred, labels, lab_f64, row);
        const float a = 2.0f * sqrtf(rs.x * rs.x + rs.y * rs.y + rs.z * rs.z), b = fabsf(rs.w);
        const float v = a > b ? a : b;
        m = (v > m || v != v) ? v : m; // a NaN residual wins: the sweep then runs unscaled and the NaN shows in the gradient
    }
    // non-negative floats (and NaN above them) order as their bit patterns
    unsigned u = __float_as_uint(m);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)u, d);
        u = o > u ? o : u;
    }
    if ((threadIdx.x & 63u) == 0) atomicMax(out_bits, u);
}

// The sweep, round 4: every wave is on its own.  One wave owns whole rows -- staged chunk by chunk, their live pairs compacted
// into 32-pair tiles across row boundaries (see the ring below) -- and the COMPLETE weight-gradient accumulators -- dW2 64 + dW1 32
// (+ 32 for the layer-2 bias) registers -- for the whole launch: nothing is shared between waves inside the loop, so there is no
// workgroup barrier in it (the previous form split the accumulators over the four waves of a block and paid nine barriers per
// round at one wave per SIMD, with the wave of a mostly padded fourth tile idling), and the [feature][pair] operands of the
// pair-contracted gradients come from a wave-private LDS transpose (tr_write / tr_read: 8-byte writes, hardware-transposed
// reads; the previous form scattered 2-byte writes).  The reverse pass through layer 2 is taken with swapped operands so that it
// lands in that layout by itself.  The prediction comes from the evaluator (mlp_eval: a row's residual needs all of its pairs).
// Per 32-pair tile: 192 + 8 v_mfma_f32_32x32x16_f16 and ~1 750 vector instructions.
template <bool TANH, typename IT>
__global__ __launch_bounds__(256, 1) void mlp_grad_tr16_kernel(const typename Vec4<IT>::type *__restrict__ nlist, unsigned B, unsigned NN,
                                                               const void *__restrict__ labels, int lab_f64,
                                                               const float4 *__restrict__ pred, const float *__restrict__ images,
                                                               MlpDims dm, float gap, const float *__restrict__ resid_max,
                                                               float *__restrict__ partial, unsigned stride, int window) {
    using I = Img<HTF_MLP_SPLIT16>;
    __shared__ __attribute__((aligned(16))) float lds[I::Floats];
    __shared__ __attribute__((aligned(16))) unsigned char trs[4 * kTrBlocks * kTrBlock];
    constexpr float kCinv = TANH ? 1.0f / 2.8853900817779268f : 1.0f;
    {
        const float4 *src = reinterpret_cast<const float4 *>(images);
        float4 *dst = reinterpret_cast<float4 *>(lds);
        for (int i = threadIdx.x; i < I::Floats / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const unsigned lane = threadIdx.x & 63u;
    const unsigned p = lane & 31u, h = lane >> 5;
    const unsigned w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned gw = blockIdx.x * 4u + w, nw = gridDim.x * 4u;
    const float ginv = 1.0f / gap;
    // RESIDUAL WINDOWS (round 6).  Everything downstream of a pair's seeds travels as fp16 hi + lo, and the launch-wide power of
    // two that puts the LARGEST seed into [1, 2) leaves a seed 2^-k of it with an absolute resolution of 2^-24 -- k bits short of
    // fp32: one outlier row (a close contact under LJ labels: 10^5 x the median residual) cost the other million rows' gradient
    // 0.5 % of its largest component (tools/train_size_probe.py, profiles/r06_train_size_probe_outlier.txt).  So a sweep is
    // kTrainWindows launches: launch w takes the rows whose residual measure (mlp_resid_max_kernel's) lies in
    // (R 2^-kWindowBits (w + 1), R 2^-kWindowBits w] -- the last one everything below -- with its own scale S 2^(kWindowBits w); the
    // partials of a window are folded back with ITS scale (mlp_reduce_partials_kernel).  A row is swept exactly once.
    const float Rmax = *resid_max;
    const bool windowed = Rmax > 0.f && Rmax < 3.0e38f;     // (a NaN or zero maximum: one window takes every row, unscaled, as before)
    const int eR = windowed ? ilogbf(Rmax) : 0;
    const float S = windowed ? ldexpf(seed_scale(Rmax), kWindowBits * window) : 1.0f;
    auto window_of = [&](const float4 &q) -> int {
        if (!windowed) return 0;
        const float a = 2.0f * sqrtf(q.x * q.x + q.y * q.y + q.z * q.z), b = fabsf(q.w);
        const float v = a > b ? a : b;
        if (!(v > 0.f)) return kTrainWindows - 1;
        const int d = (eR - ilogbf(v)) / kWindowBits;
        return d < 0 ? 0 : (d > kTrainWindows - 1 ? kTrainWindows - 1 : d);
    };
    unsigned char *tr_phi = trs + w * (kTrBlocks * kTrBlock); // phi | phid
    unsigned char *tr_a = tr_phi + 2 * kTrBlock;               // h1[0] h1[1] hd1[0] hd1[1], then zb2[0] zb2[1] zdb2[0] zdb2[1]
    f16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;

    f32x16 acc2[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}}; // [f1b][f2b]: lane (f2, h) register v = dW2[32 f1b + f0(v) + 4h][32 f2b + f2]
    f32x16 acc1[2] = {zero16(), zero16()};                             // [fb]: lane (f1, h) register v = dW1[f0(v) + 4h][32 fb + f1]
    f32x16 accb[2] = {zero16(), zero16()};                             // [f2b]: every row = sum over pairs of zb2[32 f2b + lane]
    f32x16 gw3[2] = {zero16(), zero16()};                              // P layout: partial over this lane's pairs
    float gb1[2] = {0.f, 0.f}, gb3 = 0.f, loss = 0.f;
#ifdef HTF_TRAIN_STAMPS
    unsigned long long stamps[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0, n_live = 0;
#endif

    // Live pairs are COMPACTED across rows before they become tiles: a pair's share of the gradient depends on its row only
    // through the row's residual, which is folded into the pair's two reverse seeds when the pair is staged -- so a tile is any
    // 32 live pairs.  The wave stages its rows chunk by chunk (64 slots: pair distance r and the seeds a, b into a wave-private
    // LDS ring, live slots only, ballot + mbcnt ranks) and pops 32 at a time: 389 k full tiles at C3 instead of the 424 k
    // partly filled 32-slot tiles of the rows (a row's ~95 live slots end in a tile with 31, a quarter of the rows add one with
    // 1-6).  The order of a wave's pairs is fixed, so the sum is deterministic.
    constexpr unsigned kRing = 128; // >= 31 left over + one chunk of 64
    __shared__ float ring_all[4][3][kRing];
    float(&ring)[3][kRing] = ring_all[w];
    unsigned head = 0, tail = 0; // monotonic, wave-uniform
    const unsigned nchunks = (NN + 63) / 64;
    // One wave per SIMD: nothing hides a global load but the wave's own work, so a chunk's slots (and a row's residual) are read
    // one chunk -- up to two tiles -- ahead.
    auto read_chunk = [&](unsigned row, unsigned chunk, float &ox, float &oy, float &oz) {
        ox = oy = oz = 0.f;
        const unsigned sl = chunk * 64 + lane;
        if (row < B && sl < NN) {
            const auto v = load_stream(&nlist[(size_t)row * NN + sl]);
            ox = (float)v.x; oy = (float)v.y; oz = (float)v.z;
        }
    };
    // This wave's rows are gw, gw + nw, ...; the ones of THIS launch's window are found 64 candidates at a time (lane l looks at
    // row base + l nw: its residual stays in the lane, a ballot says which are in the window) and handed out lowest first -- the
    // residual by v_readlane, so that taking the next row costs no trip to memory.
    float4 cand_rs = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned long long cand_mask = 0ull, cand_base = gw;
    auto refill = [&](unsigned long long base) {
        const unsigned long long r64 = base + (unsigned long long)lane * nw;
        const bool valid = r64 < (unsigned long long)B;
        cand_rs = valid ? residual(pred, labels, lab_f64, (unsigned)r64) : make_float4(0.f, 0.f, 0.f, 0.f);
        cand_mask = __ballot(valid && window_of(cand_rs) == window);
        cand_base = base;
    };
    auto next_row = [&](float4 &out) -> unsigned {     // -> the next row of this wave in the window (B: none left), its residual
        while (cand_mask == 0ull) {
            const unsigned long long nb = cand_base + 64ull * nw;
            if (nb >= (unsigned long long)B) return B;
            refill(nb);
        }
        const int l = __builtin_ctzll(cand_mask);
        cand_mask &= cand_mask - 1ull;
        out.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand_rs.x), l));
        out.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand_rs.y), l));
        out.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand_rs.z), l));
        out.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cand_rs.w), l));
        return (unsigned)(cand_base + (unsigned long long)l * nw);
    };
    float4 rs = make_float4(0.f, 0.f, 0.f, 0.f), nrs = rs;
    unsigned row = B, chunk = 0;
    if ((unsigned long long)gw < (unsigned long long)B) {
        refill(gw);
        row = next_row(nrs);
    }
    float nx, ny, nz;
    read_chunk(row, 0, nx, ny, nz);
    while (true) {
        if (tail - head < 32u && row < B) {
            // ---- stage one chunk of the current row
            const float x = nx, y = ny, z = nz;
            const unsigned slot = chunk * 64 + lane;
            if (chunk == 0) {
                rs = nrs;
                if (lane == 0) loss += rs.x * rs.x + rs.y * rs.y + rs.z * rs.z + rs.w * rs.w;
            }
            unsigned nrow = row, nchunk = chunk + 1;
            if (nchunk == nchunks) {
                nchunk = 0;
                nrow = next_row(nrs);
            }
            read_chunk(nrow, nchunk, nx, ny, nz);
            row = nrow;
            chunk = nchunk;
            const float tx = x + kNormDelta, ty = y + kNormDelta, tz = z + kNormDelta;
            const float rr = sqrtf(tx * tx + ty * ty + tz * tz);
            const bool live = slot < NN && rr > kRinvDelta;
            const unsigned long long mask = __ballot(live);
            if (mask != 0ull) {
                const float sa = S * (2.0f * (rs.x * tx + rs.y * ty + rs.z * tz) / rr); // reverse seeds: S = a u' + b u
                const float sb = S * rs.w;
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                if (live) {
                    const unsigned at = (tail + rank) & (kRing - 1u);
                    ring[0][at] = rr;
                    ring[1][at] = sa;
                    ring[2][at] = sb;
                    gb3 += sb;
                }
                tail += (unsigned)__builtin_popcountll(mask);
            }
            tr_fence();
            continue;
        }
        const unsigned avail = tail - head;
        if (avail == 0u) break;
        // ---- pop a tile: 32 pairs (the last one of the wave may be partial: its empty lanes carry zero seeds)
        const bool m = p < avail;
        const unsigned at = (head + p) & (kRing - 1u);
        const float r = m ? ring[0][at] : 1.0f;
        const float aq = m ? ring[1][at] : 0.f;
        const float bq = m ? ring[2][at] : 0.f;
        const float aq2 = -2.0f * aq;
        head += avail < 32u ? avail : 32u;
        tr_fence();
#ifdef HTF_TRAIN_STAMPS
        ++n_live;
        __builtin_amdgcn_sched_barrier(0);
        t_prev = __builtin_amdgcn_s_memtime();
#endif
        // The phases of a tile are software-pipelined by hand (as the evaluator's, pair_mlp.hip HTF_PIPE): with one wave per SIMD
        // nothing but the wave's own instruction stream can put vector work under a matrix instruction, and hipcc clusters each
        // kind into its own phase (measured so: vector unit 52 % busy + matrix pipe 30 % busy + waits = the whole 4.25 ms).  Each
        // HTF_TPIPE region holds one group of MFMAs and the vector block of the PREVIOUS group's results -- activation, seeds or
        // zb1 arithmetic and the hi / lo splits -- interleaved one MFMA, then `per` vector instructions.
#ifdef HTF_TRAIN_STAMPS // experiment (tools/build_obj_variant.sh): per-phase cycle totals of one wave, printed at the end
#define HTF_STAMP(i)                                                                                                   \
    do {                                                                                                               \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        stamps[i] += t_ - t_prev;                                                                                      \
        t_prev = t_;                                                                                                   \
    } while (0)
#else
#define HTF_STAMP(i)
#endif
#define HTF_TPIPE(n, per, pt)                                                                                          \
    _Pragma("unroll") for (int q_ = 0; q_ < (n); ++q_) {                                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                             \
        if ((pt) < 0) __builtin_amdgcn_sched_group_barrier(0x400, -(pt), 0); /* v_exp / v_rcp: a group of their own, */ \
        __builtin_amdgcn_sched_group_barrier(0x002, (per), 0);                                                         \
        if ((pt) > 0) __builtin_amdgcn_sched_group_barrier(0x400, (pt), 0);  /* before or after the plain ones */       \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0)
        // activation of one 32-feature block of layer 1, in place: zz <- h1, zd <- hd1, then their split forms (kept for layer 2
        // and written to the scratch for the transposed reads)
        auto act1 = [&](f32x16 &zz, f32x16 &zd, int nb, Op16 &ho, Op16 &hdo) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float hv = act_scaled<TANH>(zz[v]);
                zd[v] = TANH ? fmaf(-hv, hv, 1.0f) * (zd[v] * kCinv) : zd[v];
                zz[v] = hv;
            }
            ho = split16(zz);
            hdo = split16(zd);
            tr_write(tr_a + nb * kTrBlock, p, h, ho);
            tr_write(tr_a + (2 + nb) * kTrBlock, p, h, hdo);
        };

        // ---- V0: RBF values and r-derivatives (P layout: lane = pair)
        Op16 phi_o, phid_o;
        {
            f32x16 phi, phid;
            const f32x16 cen = load_tab(lds + I::TabC, 0, h);
            const float k1 = -1.4426950408889634f * ginv, k2 = -2.0f * ginv;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float d = r - cen[v];
                phi[v] = __builtin_amdgcn_exp2f((d * d) * k1);
                phid[v] = (d * phi[v]) * k2;
            }
            phi_o = split16(phi);
            phid_o = split16(phid);
            tr_write(tr_phi, p, h, phi_o);
            tr_write(tr_phi + kTrBlock, p, h, phid_o);
        }
        __builtin_amdgcn_sched_barrier(0);
        HTF_STAMP(1); // V0
        // ---- M1a: layer 1, block 0 (value + r-tangent)
        Op16 h1o[2], hd1o[2];
        f32x16 z1a = load_tab(lds + I::TabB1, 0, h), d1a = zero16();
        mfma_pair16(z1a, d1a, lds + I::L1, lane, phi_o, phid_o);
        __builtin_amdgcn_sched_barrier(0);
        HTF_STAMP(2); // M1a
        // ---- M1b: layer 1, block 1  ||  V1a: activation of block 0
        f32x16 z1b = load_tab(lds + I::TabB1, 1, h), d1b = zero16();
        mfma_pair16(z1b, d1b, lds + I::L1 + I::BS, lane, phi_o, phid_o);
        act1(z1a, d1a, 0, h1o[0], hd1o[0]);
        HTF_TPIPE(12, 13, 3);
        HTF_STAMP(3); // M1b | V1a
        // ---- M2a: layer 2 from block 0 of layer 1 (both output blocks)  ||  V1b: activation of block 1
        f32x16 z2a = load_tab(lds + I::TabB2, 0, h), d2a = zero16(), z2b = load_tab(lds + I::TabB2, 1, h), d2b = zero16();
        mfma_pair16(z2a, d2a, lds + I::L2 + (0 * 2 + 0) * I::BS, lane, h1o[0], hd1o[0]);
        mfma_pair16(z2b, d2b, lds + I::L2 + (1 * 2 + 0) * I::BS, lane, h1o[0], hd1o[0]);
        act1(z1b, d1b, 1, h1o[1], hd1o[1]);
        HTF_TPIPE(24, 7, 2);
        HTF_STAMP(4); // M2a | V1b
        // ---- M2b: layer 2, output block 0, from block 1
        mfma_pair16(z2a, d2a, lds + I::L2 + (0 * 2 + 1) * I::BS, lane, h1o[1], hd1o[1]);
        __builtin_amdgcn_sched_barrier(0);
        HTF_STAMP(5); // M2b
        // layer 2's activation and the reverse seeds of one block, in place: zz <- zb2, zd <- zdb2 (u = w3 . h2 + b3, u' = w3 . hd2)
        Op16 zb2o[2], zdb2o[2];
        auto seeds = [&](f32x16 &zz, f32x16 &zd, int nb) {
            const f32x16 w3 = load_tab(lds + I::TabW3, nb, h);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float hv = act_scaled<TANH>(zz[v]);
                const float s2 = TANH ? fmaf(-hv, hv, 1.0f) : 1.0f;
                const float hdv = TANH ? s2 * (zd[v] * kCinv) : zd[v];
                gw3[nb][v] = fmaf(aq, hdv, fmaf(bq, hv, gw3[nb][v]));
                const float ws = w3[v] * s2;
                // zb2 = w3 (b s2 - 2 a h2 hd2) (tanh) | w3 b;  zdb2 = a w3 s2
                zz[v] = TANH ? w3[v] * fmaf(aq2, hv * hdv, bq * s2) : bq * w3[v];
                zd[v] = aq * ws;
            }
            zb2o[nb] = split16(zz);
            zdb2o[nb] = split16(zd);
        };
        // ---- M2c: layer 2, output block 1, from block 1  ||  V2a: seeds of block 0
        mfma_pair16(z2b, d2b, lds + I::L2 + (1 * 2 + 1) * I::BS, lane, h1o[1], hd1o[1]);
        seeds(z2a, d2a, 0);
        HTF_TPIPE(12, 23, -3);
        HTF_STAMP(6); // M2c | V2a
        // ---- M3a: reverse through layer 2 INTO THE F LAYOUT, (hb1, hdb1)[pair][f1] = (zb2, zdb2)^T W2^T, from block 0  ||  V2b
        f32x16 hb[2] = {zero16(), zero16()}, hdb[2] = {zero16(), zero16()};
        mfma_pair16_t(hb[0], hdb[0], lds + I::B2 + (0 * 2 + 0) * I::BS, lane, zb2o[0], zdb2o[0]);
        mfma_pair16_t(hb[1], hdb[1], lds + I::B2 + (1 * 2 + 0) * I::BS, lane, zb2o[0], zdb2o[0]);
        seeds(z2b, d2b, 1);
        HTF_TPIPE(24, 11, -2);
        HTF_STAMP(7); // M3a | V2b
        // ---- M3b: ... from block 1  ||  layer 1's activations as [feature][pair] operands (transposed reads)
        mfma_pair16_t(hb[0], hdb[0], lds + I::B2 + (0 * 2 + 1) * I::BS, lane, zb2o[1], zdb2o[1]);
        mfma_pair16_t(hb[1], hdb[1], lds + I::B2 + (1 * 2 + 1) * I::BS, lane, zb2o[1], zdb2o[1]);
        tr_fence();
        Op16 h1F[2], hd1F[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            h1F[b] = tr_read(tr_a + b * kTrBlock, lane);
            hd1F[b] = tr_read(tr_a + (2 + b) * kTrBlock, lane);
        }
        tr_fence();
        // zb2 / zdb2 take their place in the scratch
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            tr_write(tr_a + b * kTrBlock, p, h, zb2o[b]);
            tr_write(tr_a + (2 + b) * kTrBlock, p, h, zdb2o[b]);
        }
        tr_fence();
        __builtin_amdgcn_sched_barrier(0);
        HTF_STAMP(8); // M3b + transposes
        // through act at z1, in place: hb <- zb1, hdb <- zdb1 (F layout), the layer-1 bias gradient, their split forms
        Op16 q[2], qd[2];
        auto rev1 = [&](int fb) {
            float bsum = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const float hv = op16_elem(h1F[fb], v);
                const float s1 = TANH ? fmaf(-hv, hv, 1.0f) : 1.0f;
                const float zb1 = TANH ? fmaf(hdb[fb][v] * -2.0f, hv * op16_elem(hd1F[fb], v), hb[fb][v] * s1) : hb[fb][v];
                hdb[fb][v] = hdb[fb][v] * s1; // zdb1
                hb[fb][v] = zb1;
                bsum += zb1;
            }
            gb1[fb] += bsum;
            q[fb] = split16(hb[fb]);
            qd[fb] = split16(hdb[fb]);
        };
        // ---- M4 / M5: dW2 += h1 (x) zb2 + hd1 (x) zdb2 (k = the tile's pairs), layer-2 bias gradient as a product with ones,
        //      dW1 += phi (x) zb1 + phid (x) zdb1  ||  V3: zb1, zdb1 of one block at a time
        auto dw2 = [&](int f2b) {
            const Op16 zF = tr_read(tr_a + f2b * kTrBlock, lane), zdF = tr_read(tr_a + (2 + f2b) * kTrBlock, lane);
#pragma unroll
            for (int f1b = 0; f1b < 2; ++f1b) {
                outer16_f16(acc2[f1b][f2b], h1F[f1b], zF);
                outer16_f16(acc2[f1b][f2b], hd1F[f1b], zdF);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                accb[f2b] = HTF_MFMA_H(ones, zF.lo[s], accb[f2b]);
                accb[f2b] = HTF_MFMA_H(ones, zF.hi[s], accb[f2b]);
            }
        };
        dw2(0);
        rev1(0);
        HTF_TPIPE(28, 8, 0);
        HTF_STAMP(9); // M4a | V3a
        const Op16 phiF = tr_read(tr_phi, lane), phidF = tr_read(tr_phi + kTrBlock, lane);
        dw2(1);
        outer16_f16(acc1[0], phiF, q[0]);
        outer16_f16(acc1[0], phidF, qd[0]);
        rev1(1);
        HTF_TPIPE(40, 6, 0);
        outer16_f16(acc1[1], phiF, q[1]);
        outer16_f16(acc1[1], phidF, qd[1]);
        HTF_STAMP(10); // M5
#undef HTF_TPIPE
    }

#ifdef HTF_TRAIN_STAMPS
    if (blockIdx.x == 7 && threadIdx.x == 64) {
        printf("stamps (cycles per live tile, wave 1 of block 7, %llu tiles):", n_live);
        for (int i = 1; i <= 10; ++i) printf(" %d:%llu", i, stamps[i] / (n_live ? n_live : 1));
        printf("\n");
    }
#endif
    // ---- block partial in LDS, waves in a fixed order (the scratch is free now)
    __syncthreads();
    float *red = reinterpret_cast<float *>(trs); // 1 + P <= 6338 floats
    for (unsigned c = threadIdx.x; c < stride; c += blockDim.x) red[c] = 0.f;
    __syncthreads();
    // gw3: sum over the 32 pairs of each lane half
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float s = gw3[b][v];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 8); s += __shfl_xor(s, 16);
            gw3[b][v] = s;
        }
    gb1[0] = sum_xor32(gb1[0]); // the two lane halves hold the two halves of a feature's pairs
    gb1[1] = sum_xor32(gb1[1]);
    gb3 = group_sum<64>(gb3);
    for (unsigned turn = 0; turn < 4; ++turn) {
        if (w == turn) {
#pragma unroll
            for (int f1b = 0; f1b < 2; ++f1b)
#pragma unroll
                for (int f2b = 0; f2b < 2; ++f2b) {
                    const int f2 = 32 * f2b + (int)p;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int f1 = 32 * f1b + f0(v) + 4 * (int)h;
                        if (f1 < dm.H1 && f2 < dm.H2) red[1 + dm.oW2 + f1 * dm.H2 + f2] += acc2[f1b][f2b][v];
                    }
                }
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
                const int f1 = 32 * fb + (int)p;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int k = f0(v) + 4 * (int)h;
                    if (k < dm.K && f1 < dm.H1) red[1 + k * dm.H1 + f1] += acc1[fb][v];
                }
                if (h == 0 && f1 < dm.H1) red[1 + dm.oB1 + f1] += gb1[fb];
                if (h == 0 && f1 < dm.H2) red[1 + dm.oB2 + f1] += accb[fb][0]; // (f2b = fb, f2 = f1: row 0 of the ones product)
            }
            if (p == 0) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int f = 32 * b + f0(v) + 4 * (int)h;
                        if (f < dm.H2) red[1 + dm.oW3 + f] += gw3[b][v];
                    }
            }
            if (lane == 0) {
                red[1 + dm.oB3] += gb3;
                red[0] += loss;
            }
        }
        __syncthreads();
    }
    float *out = partial + (size_t)blockIdx.x * stride;
    for (unsigned c = threadIdx.x; c < stride; c += blockDim.x) out[c] = red[c];
}


int mlp_train_grad16(const MlpDevice *m, const void *nlist, int in_dtype, unsigned B, unsigned NN, const void *labels, int lab_f64,
                     float4 *predbuf, float *partial, unsigned stride, float *resid_max, unsigned *nblk_out, hipStream_t stream) {
    int rc = mlp_eval(m, nlist, in_dtype, B, NN, predbuf, HTF_F32, nullptr, stream);
    if (rc != HTF_OK) return rc;
    HTF_CHECK_HIP(hipMemsetAsync(resid_max, 0, sizeof(float), stream));
    unsigned rblk = (B + 255u) / 256u;
    if (rblk > 1024u) rblk = 1024u;
    hipLaunchKernelGGL(mlp_resid_max_kernel, dim3(rblk), dim3(256), 0, stream, predbuf, labels, lab_f64, B, (unsigned *)resid_max);
    rc = check_launch("mlp_resid_max_kernel");
    if (rc != HTF_OK) return rc;
    unsigned nblk = (unsigned)m->n_cu; // one 4-wave workgroup per CU (146 KB of LDS), a wave per row at a time
    // (a sweep on fewer CUs, the rest left to the MD kernels that run beside it: 9.0 k MD steps/s with 256 workgroups, 8.99 / 8.74 /
    //  8.69 / 8.35 k with 224 / 192 / 160 / 128 on bench.py --workload mlp-train -- the whole chip it is)
    if ((unsigned long long)nblk * 4 > B) nblk = (B + 3u) / 4u;
    MlpDims dm{m->K, m->H1, m->H2, m->off_b1(), m->off_W2(), m->off_b2(), m->off_W3(), m->off_b3()};
    const bool th = m->act == HTF_ACT_TANH;
#define HTF_LAUNCH_TR16(T, IT, V4)                                                                                     \
    hipLaunchKernelGGL((mlp_grad_tr16_kernel<T, IT>), dim3(nblk), dim3(256), 0, stream, (const V4 *)nlist, B, NN, labels, \
                       lab_f64, predbuf, m->images, dm, m->gap, resid_max, partial + (size_t)win * nblk * stride, stride, win)
    // one launch per residual window (see the kernel): window `win`'s block partials behind window win - 1's
    for (int win = 0; win < kTrainWindows; ++win) {
        if (in_dtype == HTF_F32) {
            if (th) HTF_LAUNCH_TR16(true, float, float4); else HTF_LAUNCH_TR16(false, float, float4);
        } else {
            if (th) HTF_LAUNCH_TR16(true, double, double4); else HTF_LAUNCH_TR16(false, double, double4);
        }
    }
#undef HTF_LAUNCH_TR16
    *nblk_out = nblk;
    return check_launch("mlp_grad_tr16_kernel");
}

} // namespace htf
