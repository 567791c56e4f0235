// Online training of the closed-form trainable potentials (FORCE_MODE::hoomd2tf).
//
// Reference: tfcompute._finish_update(train) -> model.train_on_batch(x=inputs, y=labels)
// (tensorflowcompute.py:347-370) with loss MeanSquaredError over the [B, 4] force/energy
// columns; Keras differentiates the loss through compute_nlist_forces, i.e. through a
// gradient (second order).  For the parametric pair potentials the parameter-derivative of
// the per-slot force is closed form, so one sweep over the pair vectors accumulates, per
// particle row, the prediction F_i (4 values) AND J_i = d F_i / d theta (4 x P values) in
// registers; after the row reduction  d(sum res^2)/d theta_k = 2 sum_c (F_ic - Y_ic) J_ick.
// Per-block partials are written in a fixed order and reduced by one block (deterministic);
// the optimizer (Keras SGD / Adam / Nadam rules + NonNeg + L1 regulariser) is a one-thread
// kernel on the device parameter vector that every evaluation kernel reads at launch.
#ifndef __HIPCC_RTC__
#include <cmath>
#endif

#include "htf_common.h"
#include "htf_internal.h"
#include "box_math.h"
#include "pair_math.h"

namespace htf {

constexpr int kTrainG = 16; // lanes per row (as the evaluator at NN = 128); smaller NN just idle lanes

// (the body is a device function so that a generated unit -- csrc/jit_unit.hip -- can give its instantiation a C name)
template <int KIND, typename IT>
__device__ __forceinline__ void train_pair_body(const typename Vec4<IT>::type *__restrict__ nlist, unsigned B,
                                                unsigned NN, const void *__restrict__ labels, int lab_f64,
                                                void *__restrict__ pred, PotParams pin,
                                                float *__restrict__ partials) {
    constexpr int G = kTrainG, RPW = 64 / G, P = NumParams<KIND>::value;
    __shared__ float s_part[4][1 + P];
    const PotParams p = resolve_theta<KIND>(pin);
    const unsigned lane = threadIdx.x & 63u, g = lane % G, sub = lane / G;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned row = wave * RPW + sub;
    const bool active = row < B;
    const typename Vec4<IT>::type *rp = nlist + (size_t)(active ? row : B - 1) * NN;
    float F[4] = {0.f, 0.f, 0.f, 0.f};
    float4 J[P];
#pragma unroll
    for (int k = 0; k < P; ++k) J[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned j = g; j < NN; j += G) {
        auto v = load_stream(&rp[j]);
        float e, fx, fy, fz;
        float4 dd[P];
        pair_eval_grad<KIND>((float)v.x, (float)v.y, (float)v.z, p, e, fx, fy, fz, dd, (float)v.w);
        F[0] += fx; F[1] += fy; F[2] += fz; F[3] += e;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            J[k].x += dd[k].x; J[k].y += dd[k].y; J[k].z += dd[k].z; J[k].w += dd[k].w;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) F[c] = group_sum<G>(F[c]);
#pragma unroll
    for (int k = 0; k < P; ++k) {
        J[k].x = group_sum<G>(J[k].x); J[k].y = group_sum<G>(J[k].y);
        J[k].z = group_sum<G>(J[k].z); J[k].w = group_sum<G>(J[k].w);
    }
    float out[1 + P];
#pragma unroll
    for (int k = 0; k <= P; ++k) out[k] = 0.f;
    if (g == 0 && active) {
        float Y[4];
        if (lab_f64) {
            const double4 y = ((const double4 *)labels)[row];
            Y[0] = (float)y.x; Y[1] = (float)y.y; Y[2] = (float)y.z; Y[3] = (float)y.w;
        } else {
            const float4 y = ((const float4 *)labels)[row];
            Y[0] = y.x; Y[1] = y.y; Y[2] = y.z; Y[3] = y.w;
        }
        const float r0 = F[0] - Y[0], r1 = F[1] - Y[1], r2 = F[2] - Y[2], r3 = F[3] - Y[3];
        out[0] = r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3;
#pragma unroll
        for (int k = 0; k < P; ++k) out[1 + k] = 2.0f * (r0 * J[k].x + r1 * J[k].y + r2 * J[k].z + r3 * J[k].w);
        if (pred != nullptr) ((float4 *)pred)[row] = make_float4(F[0], F[1], F[2], F[3]);
    }
    // block partial: rows of a wave (lanes with g == 0), then the four waves, fixed order
#pragma unroll
    for (int k = 0; k <= P; ++k) {
        float v = out[k];
        for (int m = G; m < 64; m <<= 1) v += __shfl_xor(v, m);
        if (lane == 0) s_part[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x <= (unsigned)P)
        partials[(size_t)blockIdx.x * (1 + P) + threadIdx.x] =
            (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
}

// The same sweep FROM THE INDEX LIST (round 6): no [B, NN, 4] tensor is written for, or read by, a training step.  Sixteen lanes
// walk row i's list entries; a neighbor's position is gathered and its pair vector formed in registers exactly as
// prepareNeighbors forms it (box_math.h pair_vector: minimum image, then the r_cut mask), so a kept pair's (x, y, z, type) are the
// bits the tensor would hold; a row keeps its first NN neighbors within r_cut -- or, past NN, the LAST NN of them (the
// reference's slot wrap; the row is then redone with those bounds).  Only the order of a row's fp32 sums differs from the
// tensor sweep (a pair sits in lane (list position) % 16 instead of (slot) % 16).
// list entries per lane gathered ahead of their arithmetic.  ONE: the sweep hides its gathers with waves, not with loads in flight per
// wave -- at C3 (trainable LJ, tools/train_list_probe.py, same box) 1 / 2 / 3 / 4 / 6 entries: 71.7 / 80.8 / 83.5 / 94.6 / 120.9 us
// (the accumulators of 1 + P columns already fill the registers: 94 VGPRs at 4)
#ifndef HTF_TRAIN_LIST_CHUNK
#define HTF_TRAIN_LIST_CHUNK 1
#endif
constexpr int kTrainChunk = HTF_TRAIN_LIST_CHUNK;
// lanes per row of the list sweep
#ifndef HTF_TRAIN_LIST_G
#define HTF_TRAIN_LIST_G 16
#endif
constexpr int kTrainListG = HTF_TRAIN_LIST_G;

template <int KIND, typename PT>
__device__ __forceinline__ void train_list_body(const typename Vec4<PT>::type *__restrict__ pos, unsigned B, unsigned NN,
                                                BoxT<PT> box, const unsigned *__restrict__ n_neigh,
                                                const unsigned *__restrict__ nlist, const unsigned *__restrict__ head_list,
                                                PT rmaxsq, const void *__restrict__ labels, int lab_f64, void *__restrict__ pred,
                                                PotParams pin, float *__restrict__ partials) {
    using PV = typename Vec4<PT>::type;
    constexpr int G = kTrainListG, RPW = 64 / G, P = NumParams<KIND>::value;
    __shared__ float s_part[4][1 + P];
    const PotParams p = resolve_theta<KIND>(pin);
    const unsigned lane = threadIdx.x & 63u, g = lane % G, sub = lane / G;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned row = wave * RPW + sub;
    const bool active = row < B;
    const unsigned r_ = active ? row : B - 1;
    const unsigned nn = active ? n_neigh[r_] : 0u;
    const unsigned *nl = nlist + head_list[r_];
    const PV pi = pos[r_];
    // the longest list among the wave's four rows bounds the loop (ballots need every lane)
    unsigned nn_max = nn;
    for (int m = G; m < 64; m <<= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)nn_max, m);
        nn_max = o > nn_max ? o : nn_max;
    }
    float F[4];
    float4 J[P];
    unsigned q_lo = 0u, q_hi = NN;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        F[0] = F[1] = F[2] = F[3] = 0.f;
#pragma unroll
        for (int k = 0; k < P; ++k) J[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned Q = 0u;
        for (unsigned base = 0; base < nn_max; base += kTrainChunk * G) {
            unsigned idx[kTrainChunk];
            PV pk[kTrainChunk];
#pragma unroll
            for (int t = 0; t < kTrainChunk; ++t) {
                const unsigned j = base + t * G + g;
                idx[t] = nn ? nl[j < nn ? j : nn - 1] : 0u;
            }
#pragma unroll
            for (int t = 0; t < kTrainChunk; ++t) pk[t] = nn ? load_neighbor(pos, idx[t]) : pi;
#pragma unroll
            for (int t = 0; t < kTrainChunk; ++t) {
                const unsigned j = base + t * G + g;
                PT dx, dy, dz;
                const PT rsq = pair_vector<PT>(pk[t], pi, box, dx, dy, dz);
                const bool keep = j < nn && !(rsq > rmaxsq);
                const unsigned long long bits = (__ballot(keep) >> (G * sub)) & (G == 64 ? ~0ull : ((1ull << G) - 1ull));
                const unsigned q = Q + (unsigned)__popcll(bits & ((1ull << g) - 1ull));
                Q += (unsigned)__popcll(bits);
                const bool use = keep && q >= q_lo && q < q_hi;
                // (a slot that is not used is the tensor's zero padding: every trainable form vanishes on it)
                const float x = use ? (float)dx : 0.f, y = use ? (float)dy : 0.f, z = use ? (float)dz : 0.f;
                float e, fx, fy, fz;
                float4 dd[P];
                pair_eval_grad<KIND>(x, y, z, p, e, fx, fy, fz, dd, use ? (float)scalar_as_int(pk[t].w) : 0.f);
                F[0] += fx; F[1] += fy; F[2] += fz; F[3] += e;
#pragma unroll
                for (int k = 0; k < P; ++k) {
                    J[k].x += dd[k].x; J[k].y += dd[k].y; J[k].z += dd[k].z; J[k].w += dd[k].w;
                }
            }
        }
        // more than NN neighbors within r_cut (an error upstream): the tensor holds the last NN of them
        const bool over = Q > NN;
        q_lo = over ? Q - NN : 0u;
        q_hi = over ? Q : NN;
        if (pass == 1 || __ballot(over) == 0ull) break;
        if (!over) { q_lo = 0u; q_hi = NN; }   // (a row without overflow is simply redone with the same bounds)
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) F[c] = group_sum<G>(F[c]);
#pragma unroll
    for (int k = 0; k < P; ++k) {
        J[k].x = group_sum<G>(J[k].x); J[k].y = group_sum<G>(J[k].y);
        J[k].z = group_sum<G>(J[k].z); J[k].w = group_sum<G>(J[k].w);
    }
    float out[1 + P];
#pragma unroll
    for (int k = 0; k <= P; ++k) out[k] = 0.f;
    if (g == 0 && active) {
        float Y[4];
        if (lab_f64) {
            const double4 y = ((const double4 *)labels)[row];
            Y[0] = (float)y.x; Y[1] = (float)y.y; Y[2] = (float)y.z; Y[3] = (float)y.w;
        } else {
            const float4 y = ((const float4 *)labels)[row];
            Y[0] = y.x; Y[1] = y.y; Y[2] = y.z; Y[3] = y.w;
        }
        const float r0 = F[0] - Y[0], r1 = F[1] - Y[1], r2 = F[2] - Y[2], r3 = F[3] - Y[3];
        out[0] = r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3;
#pragma unroll
        for (int k = 0; k < P; ++k) out[1 + k] = 2.0f * (r0 * J[k].x + r1 * J[k].y + r2 * J[k].z + r3 * J[k].w);
        if (pred != nullptr) ((float4 *)pred)[row] = make_float4(F[0], F[1], F[2], F[3]);
    }
#pragma unroll
    for (int k = 0; k <= P; ++k) {
        float v = out[k];
        for (int m = G; m < 64; m <<= 1) v += __shfl_xor(v, m);
        if (lane == 0) s_part[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x <= (unsigned)P)
        partials[(size_t)blockIdx.x * (1 + P) + threadIdx.x] =
            (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
}

template <int KIND, typename PT>
__global__ __launch_bounds__(256) void train_list_kernel(const typename Vec4<PT>::type *__restrict__ pos, unsigned B, unsigned NN,
                                                         BoxT<PT> box, const unsigned *__restrict__ n_neigh,
                                                         const unsigned *__restrict__ nlist, const unsigned *__restrict__ head_list,
                                                         PT rmaxsq, const void *__restrict__ labels, int lab_f64,
                                                         void *__restrict__ pred, PotParams pin, float *__restrict__ partials) {
    train_list_body<KIND, PT>(pos, B, NN, box, n_neigh, nlist, head_list, rmaxsq, labels, lab_f64, pred, pin, partials);
}

template <int KIND, typename IT>
__global__ __launch_bounds__(256) void train_pair_kernel(const typename Vec4<IT>::type *__restrict__ nlist, unsigned B,
                                                         unsigned NN, const void *__restrict__ labels, int lab_f64,
                                                         void *__restrict__ pred, PotParams pin,
                                                         float *__restrict__ partials) {
    train_pair_body<KIND, IT>(nlist, B, NN, labels, lab_f64, pred, pin, partials);
}

#ifdef HTF_JIT_UNIT
} // namespace htf  (a generated unit takes the sweep above and nothing else of this file)
#else
// accum[w] = sum_b partials[b][w]  (one block, fixed order, fp64 accumulation)
__global__ __launch_bounds__(1024) void reduce_columns_kernel(const float *__restrict__ partials, unsigned nblocks,
                                                              unsigned width, float *__restrict__ accum) {
    __shared__ double s[16];
    for (unsigned w = 0; w < width; ++w) {
        double acc = 0.0;
        for (unsigned b = threadIdx.x; b < nblocks; b += 1024) acc += (double)partials[(size_t)b * width + w];
        for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int i = 0; i < 16; ++i) t += s[i];
            accum[w] = (float)t;
        }
    }
}

template <int KIND>
static int launch_train(const PotParams &p, const void *nlist, int in_dtype, unsigned B, unsigned NN,
                        const void *labels, int lab_f64, void *pred, float *accum, float *scratch, hipStream_t s) {
    constexpr unsigned rows_per_block = 4 * (64 / kTrainG);
    constexpr unsigned width = 1 + NumParams<KIND>::value;
    const unsigned grid = (B + rows_per_block - 1) / rows_per_block;
    if (in_dtype == HTF_F32)
        hipLaunchKernelGGL((train_pair_kernel<KIND, float>), dim3(grid), dim3(256), 0, s, (const float4 *)nlist, B, NN, labels, lab_f64, pred, p, scratch);
    else
        hipLaunchKernelGGL((train_pair_kernel<KIND, double>), dim3(grid), dim3(256), 0, s, (const double4 *)nlist, B, NN, labels, lab_f64, pred, p, scratch);
    int rc = check_launch("train_pair_kernel");
    if (rc != HTF_OK) return rc;
    hipLaunchKernelGGL(reduce_columns_kernel, dim3(1), dim3(1024), 0, s, scratch, grid, width, accum);
    return check_launch("reduce_columns_kernel");
}

template <int KIND>
static int launch_train_list(const PotParams &p, const void *pos, int pos_dtype, unsigned B, unsigned NN, const htf_box *hb,
                             const unsigned *n_neigh, const unsigned *nlist, const unsigned *head_list, double rmax,
                             const void *labels, int lab_f64, void *pred, float *accum, float *scratch, hipStream_t s) {
    constexpr unsigned rows_per_block = 4 * (64 / kTrainListG);
    constexpr unsigned width = 1 + NumParams<KIND>::value;
    const unsigned grid = (B + rows_per_block - 1) / rows_per_block;
    if (pos_dtype == HTF_F32) {
        const float rc = (float)rmax;
        hipLaunchKernelGGL((train_list_kernel<KIND, float>), dim3(grid), dim3(256), 0, s, (const float4 *)pos, B, NN, make_boxt<float>(hb),
                           n_neigh, nlist, head_list, rc * rc, labels, lab_f64, pred, p, scratch);
    } else {
        hipLaunchKernelGGL((train_list_kernel<KIND, double>), dim3(grid), dim3(256), 0, s, (const double4 *)pos, B, NN, make_boxt<double>(hb),
                           n_neigh, nlist, head_list, rmax * rmax, labels, lab_f64, pred, p, scratch);
    }
    int rc2 = check_launch("train_list_kernel");
    if (rc2 != HTF_OK) return rc2;
    hipLaunchKernelGGL(reduce_columns_kernel, dim3(1), dim3(1024), 0, s, scratch, grid, width, accum);
    return check_launch("reduce_columns_kernel");
}

int potential_num_params(const PotParams &p) {
    switch (p.kind) {
    case HTF_POT_LJ_PARAM: return 2;
    case HTF_POT_WCA: return 1;
    case HTF_POT_RINV_POLY: return p.n_terms;
    case HTF_POT_JIT: return p.n_terms; // (a traced energy's weights: how many the generated unit was compiled for)
    default: return 0;
    }
}

static unsigned train_width(const PotParams &p) {
    switch (p.kind) {
    case HTF_POT_LJ_PARAM: return 3;
    case HTF_POT_WCA: return 2;
    case HTF_POT_JIT: return 1u + (unsigned)p.n_terms;
    default: return 1 + HTF_MAX_POLY_TERMS;
    }
}

size_t train_scratch_floats(const PotParams &p, unsigned B, unsigned NN) {
    (void)NN;
    const unsigned rows_per_block = 4 * (64 / (kTrainG > kTrainListG ? kTrainG : kTrainListG)); // (either sweep's block partials fit)
    return (size_t)((B + rows_per_block - 1) / rows_per_block) * train_width(p);
}

int train_pair_dispatch(const PotParams &p, const void *nlist, int in_dtype, unsigned B, unsigned NN,
                        const void *labels, int label_dtype, void *pred, float *accum, float *scratch,
                        hipStream_t stream) {
    const int lab_f64 = label_dtype == HTF_F64;
    switch (p.kind) {
    case HTF_POT_LJ_PARAM: return launch_train<HTF_POT_LJ_PARAM>(p, nlist, in_dtype, B, NN, labels, lab_f64, pred, accum, scratch, stream);
    case HTF_POT_WCA: return launch_train<HTF_POT_WCA>(p, nlist, in_dtype, B, NN, labels, lab_f64, pred, accum, scratch, stream);
    case HTF_POT_RINV_POLY: return launch_train<HTF_POT_RINV_POLY>(p, nlist, in_dtype, B, NN, labels, lab_f64, pred, accum, scratch, stream);
    case HTF_POT_JIT: { // the generated unit's instantiation of the sweep above (csrc/jit.hip), then the library's own reduction
        constexpr unsigned rows_per_block = 4 * (64 / kTrainG);
        const unsigned grid = (B + rows_per_block - 1) / rows_per_block;
        int rc = jit_launch_train(p, nlist, in_dtype, B, NN, labels, lab_f64, pred, scratch, grid, stream);
        if (rc != HTF_OK) return rc;
        hipLaunchKernelGGL(reduce_columns_kernel, dim3(1), dim3(1024), 0, stream, scratch, grid, 1u + (unsigned)p.n_terms, accum);
        return check_launch("reduce_columns_kernel");
    }
    default:
        set_error("htf_train_pair_grad: potential kind %d has no trainable closed form (pair-MLP training: next round)", p.kind);
        return HTF_ERR_INVALID;
    }
}

int train_list_dispatch(const PotParams &p, const void *pos, int pos_dtype, unsigned B, unsigned NN, const htf_box *box,
                        const unsigned *n_neigh, const unsigned *nlist, const unsigned *head_list, double rmax, const void *labels,
                        int label_dtype, void *pred, float *accum, float *scratch, hipStream_t stream) {
    const int lab_f64 = label_dtype == HTF_F64;
#define HTF_TL(K) launch_train_list<K>(p, pos, pos_dtype, B, NN, box, n_neigh, nlist, head_list, rmax, labels, lab_f64, pred, accum, scratch, stream)
    switch (p.kind) {
    case HTF_POT_LJ_PARAM: return HTF_TL(HTF_POT_LJ_PARAM);
    case HTF_POT_WCA: return HTF_TL(HTF_POT_WCA);
    case HTF_POT_RINV_POLY: return HTF_TL(HTF_POT_RINV_POLY);
    case HTF_POT_JIT: {
        constexpr unsigned rows_per_block = 4 * (64 / kTrainListG);
        const unsigned grid = (B + rows_per_block - 1) / rows_per_block;
        int rc = jit_launch_train_list(p, pos, pos_dtype, B, NN, box, n_neigh, nlist, head_list, rmax, labels, lab_f64, pred, scratch, grid, stream);
        if (rc != HTF_OK) return rc;
        hipLaunchKernelGGL(reduce_columns_kernel, dim3(1), dim3(1024), 0, stream, scratch, grid, 1u + (unsigned)p.n_terms, accum);
        return check_launch("reduce_columns_kernel");
    }
    default:
        set_error("htf_train_pair_grad_list: potential kind %d has no list-form training sweep (the pair-MLP's reads the tensor)", p.kind);
        return HTF_ERR_INVALID;
    }
#undef HTF_TL
}

// tf.keras.optimizers.{SGD, Adam, Nadam} (TF 2.3/2.4 optimizer_v2 update rules)
__global__ void optimizer_kernel(float *__restrict__ theta, unsigned P, const float *__restrict__ accum, float scale,
                                 float *__restrict__ st, htf_optimizer_desc d) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float *m = st, *v = st + 8;
    float t = st[16] + 1.0f;
    float m_sched = st[17] == 0.0f ? 1.0f : st[17];
    const float loss = accum[0] * scale;
    st[18] += loss;
    st[19] += 1.0f;
    st[20] = loss;
    const float b1 = d.beta1, b2 = d.beta2;
    float u_t = 0.f, u_t1 = 0.f, m_sched_new = m_sched, m_sched_next = m_sched;
    if (d.kind == HTF_OPT_NADAM) { // Nadam._prepare_local: momentum schedule, decay base 0.96, 0.004
        u_t = b1 * (1.0f - 0.5f * powf(0.96f, 0.004f * t));
        u_t1 = b1 * (1.0f - 0.5f * powf(0.96f, 0.004f * (t + 1.0f)));
        m_sched_new = m_sched * u_t;
        m_sched_next = m_sched_new * u_t1;
    }
    for (unsigned k = 0; k < P; ++k) {
        const float g = accum[1 + k] * scale + d.l1_reg[k];
        float th = theta[k];
        if (d.kind == HTF_OPT_SGD) {
            th -= d.lr * g;
        } else if (d.kind == HTF_OPT_ADAM) {
            const float lr_t = d.lr * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));
            m[k] += (g - m[k]) * (1.0f - b1);
            v[k] += (g * g - v[k]) * (1.0f - b2);
            th -= lr_t * m[k] / (sqrtf(v[k]) + d.epsilon);
        } else {
            const float g_prime = g / (1.0f - m_sched_new);
            m[k] = b1 * m[k] + (1.0f - b1) * g;
            const float m_prime = m[k] / (1.0f - m_sched_next);
            v[k] = b2 * v[k] + (1.0f - b2) * g * g;
            const float v_prime = v[k] / (1.0f - powf(b2, t));
            const float m_bar = (1.0f - u_t) * g_prime + u_t1 * m_prime;
            th -= d.lr * m_bar / (sqrtf(v_prime) + d.epsilon);
        }
        if ((d.nonneg_mask >> k) & 1u) th = fmaxf(th, 0.0f); // tf.keras.constraints.NonNeg
        theta[k] = th;
    }
    st[16] = t;
    st[17] = m_sched_new;
}

// The same update rules for a parameter vector of any length (pair-MLP: 6337 weights): one
// thread per parameter, then one thread advances the shared scalars.  State: the 24-float
// header of htf_optimizer_step (scalars at [16..20]) followed by m[P], v[P].
__global__ void optimizer_vec_kernel(float *__restrict__ theta, unsigned P, const float *__restrict__ accum, float scale,
                                     float *__restrict__ st, htf_optimizer_desc d) {
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    float *m = st + HTF_OPT_STATE_FLOATS, *v = m + P;
    const float t = st[16] + 1.0f;
    const float m_sched = st[17] == 0.0f ? 1.0f : st[17];
    const float b1 = d.beta1, b2 = d.beta2;
    const float g = accum[1 + k] * scale;
    float th = theta[k];
    if (d.kind == HTF_OPT_SGD) {
        th -= d.lr * g;
    } else if (d.kind == HTF_OPT_ADAM) {
        const float lr_t = d.lr * sqrtf(1.0f - powf(b2, t)) / (1.0f - powf(b1, t));
        const float mk = m[k] + (g - m[k]) * (1.0f - b1);
        const float vk = v[k] + (g * g - v[k]) * (1.0f - b2);
        m[k] = mk;
        v[k] = vk;
        th -= lr_t * mk / (sqrtf(vk) + d.epsilon);
    } else {
        const float u_t = b1 * (1.0f - 0.5f * powf(0.96f, 0.004f * t));
        const float u_t1 = b1 * (1.0f - 0.5f * powf(0.96f, 0.004f * (t + 1.0f)));
        const float m_sched_new = m_sched * u_t, m_sched_next = m_sched_new * u_t1;
        const float g_prime = g / (1.0f - m_sched_new);
        const float mk = b1 * m[k] + (1.0f - b1) * g;
        const float vk = b2 * v[k] + (1.0f - b2) * g * g;
        m[k] = mk;
        v[k] = vk;
        const float m_bar = (1.0f - u_t) * g_prime + u_t1 * (mk / (1.0f - m_sched_next));
        th -= d.lr * m_bar / (sqrtf(vk / (1.0f - powf(b2, t))) + d.epsilon);
    }
    theta[k] = th;
}

__global__ void optimizer_vec_finish_kernel(const float *__restrict__ accum, float scale, float *__restrict__ st,
                                            htf_optimizer_desc d) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float t = st[16] + 1.0f;
    const float m_sched = st[17] == 0.0f ? 1.0f : st[17];
    const float loss = accum[0] * scale;
    st[18] += loss;
    st[19] += 1.0f;
    st[20] = loss;
    st[16] = t;
    st[17] = d.kind == HTF_OPT_NADAM ? m_sched * d.beta1 * (1.0f - 0.5f * powf(0.96f, 0.004f * t)) : m_sched;
}

} // namespace htf

extern "C" int htf_optimizer_step_n(float *d_theta, unsigned P, const float *d_accum, float scale, float *d_state,
                                    const htf_optimizer_desc *desc, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_theta && d_accum && d_state && desc, "htf_optimizer_step_n: null pointer");
    HTF_REQUIRE(P >= 1, "htf_optimizer_step_n: empty parameter vector");
    HTF_REQUIRE(desc->kind >= HTF_OPT_SGD && desc->kind <= HTF_OPT_NADAM, "htf_optimizer_step_n: unknown optimizer %d", desc->kind);
    hipLaunchKernelGGL(optimizer_vec_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_theta, P, d_accum,
                       scale, d_state, *desc);
    int rc = check_launch("optimizer_vec_kernel");
    if (rc != HTF_OK) return rc;
    hipLaunchKernelGGL(optimizer_vec_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_accum, scale, d_state, *desc);
    return check_launch("optimizer_vec_finish_kernel");
}

extern "C" int htf_optimizer_step(float *d_theta, unsigned P, const float *d_accum, float scale, float *d_state,
                                  const htf_optimizer_desc *desc, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_theta && d_accum && d_state && desc, "htf_optimizer_step: null pointer");
    HTF_REQUIRE(P >= 1 && P <= 8, "htf_optimizer_step: P=%u outside [1, 8]", P);
    HTF_REQUIRE(desc->kind >= HTF_OPT_SGD && desc->kind <= HTF_OPT_NADAM, "htf_optimizer_step: unknown optimizer %d", desc->kind);
    hipLaunchKernelGGL(optimizer_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_theta, P, d_accum, scale, d_state, *desc);
    return check_launch("optimizer_kernel");
}
#endif // HTF_JIT_UNIT
