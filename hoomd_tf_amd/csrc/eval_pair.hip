// Fused streaming evaluator for the closed-form pair potentials:
//   nlist [B, NN, 4]  ->  force [B] Scalar4 (fx, fy, fz, E_i)  (+ virial [B, 9])
//
// Replaces the TF2 graph the reference runs per step (SURVEY 3.1 "HOT LOOP"):
//   nlist_rinv (simmodel.py:618-635) -> model energy (build_examples.py / layers.py)
//   -> tf.gradients -> *2 -> reduce_sum over neighbors (simmodel.py:526-555)
//   -> _add_energy (:558-578) [-> _compute_virial (:509-523)] -> TfToHoomd copy.
// TF streams [N,NN] / [N,NN,3] tensors through ~30 elementwise kernels; here each
// 16-B slot is read from HBM exactly once and everything else stays in registers.
//
// MI355X mapping (HBM-bound: ~40 flop per 16-B slot): a group of G consecutive
// lanes owns one particle row; lane g reads slots g, g+G, g+2G, ... as float4, so
// every wave-level load instruction fetches (64/G) fully-used contiguous segments
// of G*16 B (>= one 128-B line for G >= 8).  Eight independent 16-B loads per lane
// are issued before the first use (128 B in flight per lane).  The pair->particle
// sum is a DPP row reduction inside the group -- no LDS, no atomics -- and lane 0 of
// each group stores the Scalar4.  No LDS staging: there is no reuse to exploit.
#ifndef __HIPCC_RTC__
#include <cmath>
#endif

#include "htf_common.h"
#include "htf_internal.h"
#include "pair_math.h"

namespace htf {


constexpr int kUnroll = 8;

template <typename IT>
__device__ __forceinline__ float4 load_slot(const typename Vec4<IT>::type *p) {
    auto v = load_stream(p);
    return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}

template <int KIND, int G, bool VIRIAL, typename IT>
__device__ __forceinline__ void eval_pair_body(const typename Vec4<IT>::type *__restrict__ nlist,
                                               unsigned B, unsigned NN, void *__restrict__ force,
                                               void *__restrict__ virial9, int out_f64, PotParams pin,
                                               const unsigned *__restrict__ counts) {
    constexpr int RPW = 64 / G; // particle rows per wave
    const PotParams p = resolve_theta<KIND>(pin);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned g = lane % G, sub = lane / G;
    // Blocks walk the rows from the END of the tensor: in computeForces this kernel runs right
    // after the pair-vector build, whose most recently written rows are the ones still
    // resident in the 256 MiB Infinity Cache (the 268 MB tensor does not fit; reading in
    // write order would miss everywhere, LRU-style).
    const unsigned wave = ((gridDim.x - 1 - blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const unsigned row = wave * RPW + sub;
    const bool active = row < B;
    const typename Vec4<IT>::type *rp = nlist + (size_t)(active ? row : B - 1) * NN;

    float fx = 0.f, fy = 0.f, fz = 0.f, en = 0.f;
    float vxx = 0.f, vxy = 0.f, vxz = 0.f, vyy = 0.f, vyz = 0.f, vzz = 0.f;
    float own_type = 0.f; // positions[row, 3]: read by generated bodies only
    if constexpr (KIND == HTF_POT_JIT) {
        if (pin.own != nullptr && active)
            own_type = pin.own_f64 ? (float)reinterpret_cast<const double4 *>(pin.own)[row].w : reinterpret_cast<const float4 *>(pin.own)[row].w;
    }
    // live slots of this row when the producer recorded them (context path): the zero
    // padding behind them contributes nothing and is not fetched
    unsigned cnt = NN;
    if (counts != nullptr && active) {
        cnt = counts[row];
        cnt = cnt < NN ? cnt : NN;
    }

    for (unsigned j0 = 0; j0 < NN; j0 += kUnroll * G) {
        float4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            unsigned j = j0 + u * G + g;
            v[u] = (j < cnt) ? load_slot<IT>(rp + j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            float e, ax, ay, az;
            pair_eval<KIND>(v[u].x, v[u].y, v[u].z, p, e, ax, ay, az, v[u].w, own_type);
            fx += ax;
            fy += ay;
            fz += az;
            en += e;
            if constexpr (VIRIAL) {
                // simmodel.py:509-523: -(|nf| / (2 |x|)) x (x) x, divide_no_nan
                float x = v[u].x, y = v[u].y, z = v[u].z;
                float fmag = sqrtf(ax * ax + ay * ay + az * az);
                float den = 2.0f * sqrtf(x * x + y * y + z * z);
                float frs = (den == 0.0f) ? 0.0f : fmag / den;
                vxx -= frs * x * x;
                vxy -= frs * x * y;
                vxz -= frs * x * z;
                vyy -= frs * y * y;
                vyz -= frs * y * z;
                vzz -= frs * z * z;
            }
        }
    }

    fx = group_sum<G>(fx);
    fy = group_sum<G>(fy);
    fz = group_sum<G>(fz);
    en = group_sum<G>(en);
    if constexpr (VIRIAL) {
        vxx = group_sum<G>(vxx);
        vxy = group_sum<G>(vxy);
        vxz = group_sum<G>(vxz);
        vyy = group_sum<G>(vyy);
        vyz = group_sum<G>(vyz);
        vzz = group_sum<G>(vzz);
    }

    if constexpr (HasRowFn<KIND>::value) { // (generated units: energy_i = F(sum_j e_ij), pair_math.h row_function)
        float Fv, dF;
        row_function<KIND>(p, en, Fv, dF);
        fx *= dF;
        fy *= dF;
        fz *= dF;
        en = Fv;
        if constexpr (VIRIAL) {
            const float a = fabsf(dF);
            vxx *= a;
            vxy *= a;
            vxz *= a;
            vyy *= a;
            vyz *= a;
            vzz *= a;
        }
    }
    if (g == 0 && active) {
        if (out_f64) {
            ((double4 *)force)[row] = make_double4(fx, fy, fz, en);
        } else {
            ((float4 *)force)[row] = make_float4(fx, fy, fz, en);
        }
        if constexpr (VIRIAL) {
            const float v9[9] = {vxx, vxy, vxz, vxy, vyy, vyz, vxz, vyz, vzz};
            if (out_f64) {
                double *o = (double *)virial9 + (size_t)row * 9;
#pragma unroll
                for (int c = 0; c < 9; ++c) o[c] = v9[c];
            } else {
                float *o = (float *)virial9 + (size_t)row * 9;
#pragma unroll
                for (int c = 0; c < 9; ++c) o[c] = v9[c];
            }
        }
    }
}

// (the body is a device function so that a generated unit -- csrc/jit_unit.hip -- can give its instantiations C names)
template <int KIND, int G, bool VIRIAL, typename IT>
__global__ __launch_bounds__(256) void eval_pair_kernel(const typename Vec4<IT>::type *__restrict__ nlist,
                                                        unsigned B, unsigned NN, void *__restrict__ force,
                                                        void *__restrict__ virial9, int out_f64, PotParams pin,
                                                        const unsigned *__restrict__ counts) {
    eval_pair_body<KIND, G, VIRIAL, IT>(nlist, B, NN, force, virial9, out_f64, pin, counts);
}

#ifdef HTF_JIT_UNIT
} // namespace htf  (a generated unit takes the templates above and nothing else of this file)
#else
template <int KIND, int G, bool VIRIAL, typename IT>
static int launch_eval_g(const void *nlist, unsigned B, unsigned NN, void *force, void *virial9,
                         int out_f64, const PotParams &p, const unsigned *counts, hipStream_t stream) {
    constexpr unsigned rows_per_block = 4 * (64 / G);
    unsigned grid = (B + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL((eval_pair_kernel<KIND, G, VIRIAL, IT>), dim3(grid), dim3(256), 0, stream,
                       (const typename Vec4<IT>::type *)nlist, B, NN, force, virial9, out_f64, p, counts);
    return check_launch("eval_pair_kernel");
}

// lanes per particle row: 8 slots per lane where NN allows (NN=128 -> 16 lanes)
static int pick_group(unsigned NN) {
    if (NN >= 128) return 16;
    if (NN >= 64) return 8;
    return 4;
}

template <int KIND, bool VIRIAL, typename IT>
static int launch_eval_k(const void *nlist, unsigned B, unsigned NN, void *force, void *virial9,
                         int out_f64, const PotParams &p, const unsigned *counts, hipStream_t stream) {
    switch (pick_group(NN)) {
    case 16: return launch_eval_g<KIND, 16, VIRIAL, IT>(nlist, B, NN, force, virial9, out_f64, p, counts, stream);
    case 8: return launch_eval_g<KIND, 8, VIRIAL, IT>(nlist, B, NN, force, virial9, out_f64, p, counts, stream);
    default: return launch_eval_g<KIND, 4, VIRIAL, IT>(nlist, B, NN, force, virial9, out_f64, p, counts, stream);
    }
}

template <int KIND>
static int launch_eval(const void *nlist, int in_dtype, unsigned B, unsigned NN, void *force,
                       void *virial9, int out_f64, const PotParams &p, const unsigned *counts, hipStream_t stream) {
    if (in_dtype == HTF_F32) {
        return virial9 ? launch_eval_k<KIND, true, float>(nlist, B, NN, force, virial9, out_f64, p, counts, stream)
                       : launch_eval_k<KIND, false, float>(nlist, B, NN, force, virial9, out_f64, p, counts, stream);
    }
    return virial9 ? launch_eval_k<KIND, true, double>(nlist, B, NN, force, virial9, out_f64, p, counts, stream)
                   : launch_eval_k<KIND, false, double>(nlist, B, NN, force, virial9, out_f64, p, counts, stream);
}

int eval_pair_dispatch(const PotParams &p, const void *nlist, int in_dtype, unsigned B, unsigned NN,
                       void *force, int force_dtype, void *virial9, const unsigned *counts, hipStream_t stream) {
    const int out_f64 = force_dtype == HTF_F64;
    switch (p.kind) {
    case HTF_POT_LJ: return launch_eval<HTF_POT_LJ>(nlist, in_dtype, B, NN, force, virial9, out_f64, p, counts, stream);
    case HTF_POT_WCA: return launch_eval<HTF_POT_WCA>(nlist, in_dtype, B, NN, force, virial9, out_f64, p, counts, stream);
    case HTF_POT_RINV_POLY: return launch_eval<HTF_POT_RINV_POLY>(nlist, in_dtype, B, NN, force, virial9, out_f64, p, counts, stream);
    case HTF_POT_SIMPLE: return launch_eval<HTF_POT_SIMPLE>(nlist, in_dtype, B, NN, force, virial9, out_f64, p, counts, stream);
    case HTF_POT_GAUSS: return launch_eval<HTF_POT_GAUSS>(nlist, in_dtype, B, NN, force, virial9, out_f64, p, counts, stream);
    case HTF_POT_LJ_PARAM: return launch_eval<HTF_POT_LJ_PARAM>(nlist, in_dtype, B, NN, force, virial9, out_f64, p, counts, stream);
    case HTF_POT_JIT: return jit_launch_eval(p, nlist, in_dtype, B, NN, force, virial9, out_f64, counts, stream);
    default:
        set_error("eval_pair_dispatch: potential kind %d is not a closed-form pair potential", p.kind);
        return HTF_ERR_INVALID;
    }
}

// ---- two potentials in one pass (EDS-biased models: base potential + Gaussian CV channel) ----
struct RdfArgs { // optional compute_rdf histogram fused into the same sweep (hist == nullptr: off)
    float r0, r1;
    unsigned nb;
    unsigned *hist;
};
constexpr unsigned kRdfMaxBins = 1024;

template <int KA, int G, typename IT>
__global__ __launch_bounds__(256) void eval_pair2_kernel(const typename Vec4<IT>::type *__restrict__ nlist, unsigned B,
                                                         unsigned NN, void *__restrict__ forceA,
                                                         void *__restrict__ forceB, int out_f64, PotParams pa_in,
                                                         PotParams pb, float *__restrict__ partials, RdfArgs rdf) {
    constexpr int RPW = 64 / G;
    const PotParams pa = resolve_theta<KA>(pa_in);
    __shared__ float s_part[4];
    __shared__ unsigned s_hist[kRdfMaxBins];
    if (rdf.hist != nullptr) {
        for (unsigned i = threadIdx.x; i < rdf.nb; i += blockDim.x) s_hist[i] = 0;
        __syncthreads();
    }
    const unsigned lane = threadIdx.x & 63u;
    const unsigned g = lane % G, sub = lane / G;
    const unsigned ngroups = (B + 4 * RPW - 1) / (4 * RPW); // one group = the 4*RPW rows a block sweeps per trip
    unsigned n_lo = 0, n_hi = 0;
    // persistent blocks: the LDS histogram is flushed once per block, not once per 16 rows
    // (16 384 flushes x ~100 same-address global atomics cost 200 us at C4)
    for (unsigned grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        // walk the rows cache-first (see eval_pair_kernel)
        const unsigned rgrp = ngroups - 1 - grp;
        const unsigned row = (rgrp * 4 + (threadIdx.x >> 6)) * RPW + sub;
        const bool active = row < B;
        const typename Vec4<IT>::type *rp = nlist + (size_t)(active ? row : B - 1) * NN;
        float ax = 0.f, ay = 0.f, az = 0.f, ae = 0.f, bx = 0.f, by = 0.f, bz = 0.f, be = 0.f;
        for (unsigned j0 = 0; j0 < NN; j0 += kUnroll * G) {
            float4 v[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                unsigned j = j0 + u * G + g;
                v[u] = (j < NN) ? load_slot<IT>(rp + j) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                float e, fx, fy, fz;
                const RinvFwd f = rinv_fwd(v[u].x, v[u].y, v[u].z); // once for both potentials
                pair_eval_f<KA>(f, v[u].x, v[u].y, v[u].z, pa, e, fx, fy, fz);
                ax += fx; ay += fy; az += fz; ae += e;
                pair_eval_f<HTF_POT_GAUSS>(f, v[u].x, v[u].y, v[u].z, pb, e, fx, fy, fz);
                bx += fx; by += fy; bz += fz; be += e;
                if (rdf.hist != nullptr && active && j0 + u * G + g < NN) {
                    // compute_rdf (simmodel.py:661-662): plain norm, histogram_fixed_width clamping
                    const float r = plain_norm3(v[u].x, v[u].y, v[u].z);
                    const float fi = floorf((float)rdf.nb * ((r - rdf.r0) / (rdf.r1 - rdf.r0)));
                    const int idx = fi < 0.f ? 0 : (fi > (float)(rdf.nb - 1) ? (int)(rdf.nb - 1) : (int)fi);
                    // the clamped end bins take every padded slot (26 % of the tensor lands in bin 0):
                    // count those in a register, only interior bins go through LDS atomics
                    if (idx == 0) ++n_lo;
                    else if (idx == (int)rdf.nb - 1) ++n_hi;
                    else atomicAdd(&s_hist[idx], 1u);
                }
            }
        }
        ax = group_sum<G>(ax); ay = group_sum<G>(ay); az = group_sum<G>(az); ae = group_sum<G>(ae);
        bx = group_sum<G>(bx); by = group_sum<G>(by); bz = group_sum<G>(bz); be = group_sum<G>(be);
        if (g == 0 && active) {
            if (out_f64) {
                ((double4 *)forceA)[row] = make_double4(ax, ay, az, ae);
                ((double4 *)forceB)[row] = make_double4(bx, by, bz, be);
            } else {
                ((float4 *)forceA)[row] = make_float4(ax, ay, az, ae);
                ((float4 *)forceB)[row] = make_float4(bx, by, bz, be);
            }
        }
        if (partials != nullptr) { // per-group sum of the B energy column, fixed order -> deterministic
            float c = (g == 0 && active) ? be : 0.f;
            c = group_sum<64>(c);
            __syncthreads(); // s_part reuse across trips
            if (lane == 0) s_part[threadIdx.x >> 6] = c;
            __syncthreads();
            if (threadIdx.x == 0) partials[rgrp] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        }
    }
    if (rdf.hist != nullptr) {
        n_lo = group_sum_u<64>(n_lo);
        n_hi = group_sum_u<64>(n_hi);
        if (lane == 0) {
            if (n_lo) atomicAdd(&s_hist[0], n_lo);
            if (n_hi) atomicAdd(&s_hist[rdf.nb - 1], n_hi);
        }
        __syncthreads();
        for (unsigned i = threadIdx.x; i < rdf.nb; i += blockDim.x)
            if (s_hist[i]) atomicAdd(&rdf.hist[i], s_hist[i]);
    }
}

template <int KA, int G, typename IT>
static int launch_eval2_g(const void *nlist, unsigned B, unsigned NN, void *fa, void *fb, int out_f64,
                          const PotParams &pa, const PotParams &pb, float *partials, const RdfArgs &rdf, hipStream_t stream) {
    constexpr unsigned rows_per_block = 4 * (64 / G);
    unsigned grid = (B + rows_per_block - 1) / rows_per_block;
    if (grid > 2048u) grid = 2048u; // persistent: 8 blocks per CU
    hipLaunchKernelGGL((eval_pair2_kernel<KA, G, IT>), dim3(grid), dim3(256), 0, stream,
                       (const typename Vec4<IT>::type *)nlist, B, NN, fa, fb, out_f64, pa, pb, partials, rdf);
    return check_launch("eval_pair2_kernel");
}

template <int KA, typename IT>
static int launch_eval2_k(const void *nlist, unsigned B, unsigned NN, void *fa, void *fb, int out_f64,
                          const PotParams &pa, const PotParams &pb, float *partials, const RdfArgs &rdf,
                          hipStream_t stream) {
    // 16 lanes per row at every NN: this two-kernel form is the fallback of the one-kernel sweep (fused_eval.hip), which is what
    // config C4 runs; the 8- and 4-lane groups for short rows exist in variants builds only
#ifdef HTF_AB_VARIANTS
    switch (pick_group(NN)) {
    case 16: return launch_eval2_g<KA, 16, IT>(nlist, B, NN, fa, fb, out_f64, pa, pb, partials, rdf, stream);
    case 8: return launch_eval2_g<KA, 8, IT>(nlist, B, NN, fa, fb, out_f64, pa, pb, partials, rdf, stream);
    default: return launch_eval2_g<KA, 4, IT>(nlist, B, NN, fa, fb, out_f64, pa, pb, partials, rdf, stream);
    }
#else
    return launch_eval2_g<KA, 16, IT>(nlist, B, NN, fa, fb, out_f64, pa, pb, partials, rdf, stream);
#endif
}

unsigned eval_pair2_num_partials(unsigned B, unsigned NN) {
#ifdef HTF_AB_VARIANTS
    const unsigned rows_per_block = 4 * (64 / pick_group(NN));
#else
    (void)NN;
    const unsigned rows_per_block = 4 * (64 / 16);
#endif
    return (B + rows_per_block - 1) / rows_per_block;
}

int eval_pair2_dispatch(const PotParams &pa, const PotParams &pb, const void *nlist, int in_dtype, unsigned B,
                        unsigned NN, void *forceA, void *forceB, int force_dtype, float *partials, float rdf_r0,
                        float rdf_r1, unsigned rdf_nbins_total, unsigned *rdf_hist, hipStream_t stream) {
    const int out_f64 = force_dtype == HTF_F64;
    RdfArgs rdf{rdf_r0, rdf_r1, rdf_nbins_total, rdf_hist};
    if (rdf_hist != nullptr && (rdf_nbins_total < 3 || rdf_nbins_total > kRdfMaxBins || !(rdf_r1 > rdf_r0))) {
        set_error("htf_eval_forces2: bad fused rdf arguments (nbins+2 = %u, range [%g, %g])", rdf_nbins_total, rdf_r0, rdf_r1);
        return HTF_ERR_INVALID;
    }
#define HTF_E2(K) (in_dtype == HTF_F32 ? launch_eval2_k<K, float>(nlist, B, NN, forceA, forceB, out_f64, pa, pb, partials, rdf, stream) \
                                       : launch_eval2_k<K, double>(nlist, B, NN, forceA, forceB, out_f64, pa, pb, partials, rdf, stream))
    switch (pa.kind) {
    case HTF_POT_LJ: return HTF_E2(HTF_POT_LJ);
    case HTF_POT_WCA: return HTF_E2(HTF_POT_WCA);
    case HTF_POT_RINV_POLY: return HTF_E2(HTF_POT_RINV_POLY);
    default:
        set_error("htf_eval_forces2: base potential kind %d is not a closed-form rinv potential", pa.kind);
        return HTF_ERR_INVALID;
    }
#undef HTF_E2
}

__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float *__restrict__ partials, unsigned n, float scale,
                                                               float *__restrict__ out) {
    __shared__ double s[16];
    double acc = 0.0;
    for (unsigned i = threadIdx.x; i < n; i += 1024) acc += (double)partials[i];
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += s[i];
        *out = (float)(t * (double)scale);
    }
}

template <typename V>
__global__ __launch_bounds__(256) void bias_combine_kernel(V *__restrict__ force, const V *__restrict__ bias,
                                                           const float *__restrict__ alpha_p,
                                                           const float *__restrict__ cv_p, unsigned N) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float alpha = *alpha_p, cv = *cv_p;
    V f = force[i], b = bias[i];
    f.x += alpha * b.x;
    f.y += alpha * b.y;
    f.z += alpha * b.z;
    f.w += alpha * cv; // rank-0 energy term tiled into every particle (simmodel.py:567-572)
    force[i] = f;
}

// ---- small elementwise companions -------------------------------------------------

template <typename IT>
__global__ __launch_bounds__(256) void nlist_rinv_kernel(const typename Vec4<IT>::type *__restrict__ nlist,
                                                         size_t n, float *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float4 v = load_slot<IT>(nlist + i);
        out[i] = rinv_fwd(v.x, v.y, v.z).s;
    }
}

// simmodel.py:214-219: max over rows of the number of slots with dx > 0
template <typename IT, int G>
__global__ __launch_bounds__(256) void check_nlist_kernel(const typename Vec4<IT>::type *__restrict__ nlist,
                                                          unsigned B, unsigned NN, unsigned *__restrict__ out) {
    constexpr int RPW = 64 / G;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned g = lane % G, sub = lane / G;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned row = wave * RPW + sub;
    unsigned cnt = 0;
    if (row < B) {
        const typename Vec4<IT>::type *rp = nlist + (size_t)row * NN;
        for (unsigned j = g; j < NN; j += G) cnt += (__builtin_nontemporal_load(&rp[j].x) > (IT)0) ? 1u : 0u;
    }
    cnt = group_sum_u<G>(cnt);
    // wave max, then one atomic per wave
    for (int m = G; m < 64; m <<= 1) {
        unsigned o = (unsigned)__shfl_xor((int)cnt, m);
        cnt = o > cnt ? o : cnt;
    }
    if (lane == 0 && cnt > 0) atomicMax(out, cnt);
}

} // namespace htf

extern "C" int htf_nlist_rinv(const void *d_nlist, int nlist_dtype, unsigned B, unsigned NN, float *d_out,
                              htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_nlist && d_out, "htf_nlist_rinv: null pointer");
    HTF_REQUIRE(nlist_dtype == HTF_F32 || nlist_dtype == HTF_F64, "htf_nlist_rinv: bad dtype %d", nlist_dtype);
    size_t n = (size_t)B * NN;
    if (n == 0) return HTF_OK;
    unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (nlist_dtype == HTF_F32)
        hipLaunchKernelGGL((nlist_rinv_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4 *)d_nlist, n, d_out);
    else
        hipLaunchKernelGGL((nlist_rinv_kernel<double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const double4 *)d_nlist, n, d_out);
    return check_launch("nlist_rinv_kernel");
}

extern "C" int htf_check_nlist(const void *d_nlist, int nlist_dtype, unsigned B, unsigned NN, unsigned *d_out,
                               htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_nlist && d_out, "htf_check_nlist: null pointer");
    HTF_REQUIRE(nlist_dtype == HTF_F32 || nlist_dtype == HTF_F64, "htf_check_nlist: bad dtype %d", nlist_dtype);
    if (B == 0 || NN == 0) return HTF_OK;
    constexpr int G = 16;
    constexpr unsigned rows_per_block = 4 * (64 / G);
    unsigned grid = (B + rows_per_block - 1) / rows_per_block;
    if (nlist_dtype == HTF_F32)
        hipLaunchKernelGGL((check_nlist_kernel<float, G>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4 *)d_nlist, B, NN, d_out);
    else
        hipLaunchKernelGGL((check_nlist_kernel<double, G>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const double4 *)d_nlist, B, NN, d_out);
    return check_launch("check_nlist_kernel");
}

extern "C" int htf_reduce_partials(const float *d_partials, unsigned n, float scale, float *d_out, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_partials && d_out, "htf_reduce_partials: null pointer");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_partials, n, scale, d_out);
    return check_launch("reduce_partials_kernel");
}

extern "C" int htf_bias_combine(void *d_force, const void *d_bias, const float *d_alpha, const float *d_cv, int dtype,
                                unsigned N, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_force && d_bias && d_alpha && d_cv, "htf_bias_combine: null pointer");
    if (N == 0) return HTF_OK;
    unsigned grid = (N + 255) / 256;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((bias_combine_kernel<float4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (float4 *)d_force, (const float4 *)d_bias, d_alpha, d_cv, N);
    else if (dtype == HTF_F64)
        hipLaunchKernelGGL((bias_combine_kernel<double4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (double4 *)d_force, (const double4 *)d_bias, d_alpha, d_cv, N);
    else {
        set_error("htf_bias_combine: bad dtype %d", dtype);
        return HTF_ERR_INVALID;
    }
    return check_launch("bias_combine_kernel");
}
#endif // HTF_JIT_UNIT
