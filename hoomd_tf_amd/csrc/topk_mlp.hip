// Example 08's model on the device (examples/08 "Learning Potentials" cell 3, build_examples.py:199-218
// NlistNN): per PARTICLE, the K largest 1/r of its neighbor row in descending order -> Dense(H1) ->
// Dense(H2) -> Dense(1) = E_i, forces by compute_nlist_forces (simmodel.py:526-555): F_i = 2 sum_j dE_i/dx_ij.
//
//   rinv  = nlist_rinv(nlist)                                        simmodel.py:618-635
//   top_n = tf.sort(rinv, axis=1, direction='DESCENDING')[:, :K]     (top_k underneath: ties -> lower index first)
//   E_i   = Dense(1)(act(Dense(H2)(act(Dense(H1)(top_n)))))          Keras Dense: x W + b, activation None by default
//
// One wave64 per particle row.  Each lane holds ceil(NN / 64) slots; the K winners are extracted by K wave-wide
// arg-max rounds on a 64-bit key (1/r bits, ~slot): largest value first, lowest slot among equals -- the order
// matters on a perfect lattice, where equidistant neighbors take DIFFERENT first-layer weights.  The tiny network
// (example 08: 8 -> 16 -> 16 -> 1, here K <= 16, H <= 64) runs with one hidden unit per lane, activations
// exchanged through a wave-private LDS line (broadcast reads), forward then backward; the gradient with respect
// to the k-th sorted value goes back to the slot that supplied it.  No sort of the whole row, no [N, NN]
// intermediate, no autograd graph: 16 B per slot read once.  HBM-bound like the closed-form evaluators.
#include <new>
#include <vector>

#include "htf_common.h"
#include "htf_internal.h"
#include "pair_math.h"

namespace htf {

constexpr int kTopMax = 16;  // top_neighs
constexpr int kTopH = 64;    // widest hidden layer
constexpr int kTopSlots = 4; // slots per lane: NN <= 256

struct TopkDevice {
    float *w = nullptr; // W1 [K][H1] | b1 [H1] | W2 [H1][H2] | b2 [H2] | W3 [H2] | b3, row-major Keras kernels
    int K = 0, H1 = 0, H2 = 0, act = 0;
    int n_floats() const { return K * H1 + H1 + H1 * H2 + H2 + H2 + 1; }
};

template <bool TANH>
__device__ __forceinline__ float tk_act(float z) {
    if constexpr (!TANH) return z;
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * z)), 1.0f);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)k, m);
        const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(k >> 32), m);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        k = o > k ? o : k;
    }
    return k;
}

template <bool TANH, bool VIRIAL, typename IT>
__global__ __launch_bounds__(256) void topk_mlp_kernel(const typename Vec4<IT>::type *__restrict__ nlist, unsigned B,
                                                       unsigned NN, void *__restrict__ force, void *__restrict__ virial9,
                                                       int out_f64, const float *__restrict__ weights, int K, int H1,
                                                       int H2) {
    extern __shared__ float s_mem[];
    const int nw = K * H1 + H1 + H1 * H2 + H2 + H2 + 1;
    float *s_w = s_mem;                                  // the weights, once per block
    float *s_x = s_mem + ((nw + 3) & ~3) + (threadIdx.x >> 6) * kTopH; // this wave's exchange line
    for (int i = threadIdx.x; i < nw; i += blockDim.x) s_w[i] = weights[i];
    __syncthreads();
    const float *W1 = s_w, *b1 = W1 + K * H1, *W2 = b1 + H1, *b2 = W2 + H1 * H2, *W3 = b2 + H2, *b3 = W3 + H2;

    const unsigned lane = threadIdx.x & 63u;
    const unsigned row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= B) return; // wave-uniform (no barrier below)
    const typename Vec4<IT>::type *rp = nlist + (size_t)row * NN;

    // 1. this lane's slots: pair vector, 1/r and d(1/r)/dx pieces
    float x[kTopSlots], y[kTopSlots], z[kTopSlots];
    RinvFwd f[kTopSlots];
    int rank[kTopSlots];
#pragma unroll
    for (int t = 0; t < kTopSlots; ++t) {
        const unsigned slot = t * 64 + lane;
        x[t] = y[t] = z[t] = 0.f;
        if (slot < NN) {
            const auto v = load_stream(&rp[slot]);
            x[t] = (float)v.x; y[t] = (float)v.y; z[t] = (float)v.z;
        }
        f[t] = rinv_fwd(x[t], y[t], z[t]);
        if (slot >= NN) f[t].s = -1.f; // not a slot of this row: below every real value (padding has s = 0)
        rank[t] = -1;
    }

    // 2. K rounds of wave arg-max: value descending, lower slot first among equals (tf.sort DESCENDING = top_k)
    float top[kTopMax];
#pragma unroll
    for (int k = 0; k < kTopMax; ++k) {
        if (k >= K) { top[k] = 0.f; continue; } // wave-uniform
        unsigned long long best = 0ull;
#pragma unroll
        for (int t = 0; t < kTopSlots; ++t) {
            if (rank[t] < 0 && f[t].s >= 0.f) {
                const unsigned slot = t * 64 + lane;
                const unsigned long long key = ((unsigned long long)__float_as_uint(f[t].s) << 32) | (0xFFFFFFFFu - slot);
                best = key > best ? key : best;
            }
        }
        // (every row has >= K slots: the host checks NN >= K)
        best = wave_max_u64(best);
        top[k] = __uint_as_float((unsigned)(best >> 32));
        const unsigned wslot = 0xFFFFFFFFu - (unsigned)best;
#pragma unroll
        for (int t = 0; t < kTopSlots; ++t)
            if ((unsigned)(t * 64) + lane == wslot) rank[t] = k;
    }

    // 3. forward: one hidden unit per lane
    float z1 = 0.f, h1 = 0.f;
    if ((int)lane < H1) {
        z1 = b1[lane];
#pragma unroll
        for (int k = 0; k < kTopMax; ++k)
            if (k < K) z1 = fmaf(top[k], W1[k * H1 + lane], z1);
        h1 = tk_act<TANH>(z1);
    }
    s_x[lane] = h1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float h2 = 0.f;
    if ((int)lane < H2) {
        float z2 = b2[lane];
        for (int a = 0; a < H1; ++a) z2 = fmaf(s_x[a], W2[a * H2 + lane], z2);
        h2 = tk_act<TANH>(z2);
    }
    const float w3 = (int)lane < H2 ? W3[lane] : 0.f;
    const float energy = group_sum<64>(h2 * w3) + b3[0];

    // 4. backward: g2 = dE/dz2, g1 = dE/dz1, gt[k] = dE/dtop_k
    const float g2 = TANH ? w3 * (1.0f - h2 * h2) : w3;
    __builtin_amdgcn_wave_barrier();
    s_x[lane] = g2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float g1 = 0.f;
    if ((int)lane < H1) {
        float acc = 0.f;
        for (int b = 0; b < H2; ++b) acc = fmaf(s_x[b], W2[lane * H2 + b], acc);
        g1 = TANH ? acc * (1.0f - h1 * h1) : acc;
    }
    __builtin_amdgcn_wave_barrier();
    s_x[lane] = g1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // every lane forms the gradient of the ranks its own slots hold (at most kTopSlots of them)
    float fx = 0.f, fy = 0.f, fz = 0.f;
    Virial6 vir;
#pragma unroll
    for (int t = 0; t < kTopSlots; ++t) {
        if (rank[t] >= 0 && f[t].cond) {
            float gt = 0.f;
            for (int a = 0; a < H1; ++a) gt = fmaf(s_x[a], W1[rank[t] * H1 + a], gt);
            // d s / d r' = -s^2, d r' / d x = t / r'; nlist_forces = 2 dE/dx  (simmodel.py:548)
            const float c = 2.0f * (gt * (-(f[t].s * f[t].s))) * f[t].irp;
            const float ax = c * f[t].tx, ay = c * f[t].ty, az = c * f[t].tz;
            fx += ax; fy += ay; fz += az;
            if constexpr (VIRIAL) vir.add(x[t], y[t], z[t], ax, ay, az);
        }
    }
    fx = group_sum<64>(fx);
    fy = group_sum<64>(fy);
    fz = group_sum<64>(fz);
    float v6[6];
    if constexpr (VIRIAL) {
        v6[0] = group_sum<64>(vir.xx); v6[1] = group_sum<64>(vir.xy); v6[2] = group_sum<64>(vir.xz);
        v6[3] = group_sum<64>(vir.yy); v6[4] = group_sum<64>(vir.yz); v6[5] = group_sum<64>(vir.zz);
    }
    if (lane == 0) {
        if (out_f64)
            ((double4 *)force)[row] = make_double4(fx, fy, fz, energy);
        else
            ((float4 *)force)[row] = make_float4(fx, fy, fz, energy);
        if constexpr (VIRIAL) {
            const float v9[9] = {v6[0], v6[1], v6[2], v6[1], v6[3], v6[4], v6[2], v6[4], v6[5]};
#pragma unroll
            for (int c9 = 0; c9 < 9; ++c9) {
                if (out_f64)
                    ((double *)virial9)[(size_t)row * 9 + c9] = v9[c9];
                else
                    ((float *)virial9)[(size_t)row * 9 + c9] = v9[c9];
            }
        }
    }
}

// the K largest values of every row, descending, and the slots they came from (tf.math.top_k semantics)
__global__ __launch_bounds__(256) void topk_values_kernel(const float *__restrict__ xin, unsigned B, unsigned n, int K,
                                                          float *__restrict__ vals, int *__restrict__ idx) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= B) return;
    float v[kTopSlots];
    bool taken[kTopSlots];
#pragma unroll
    for (int t = 0; t < kTopSlots; ++t) {
        const unsigned s = t * 64 + lane;
        v[t] = s < n ? xin[(size_t)row * n + s] : 0.f;
        taken[t] = s >= n;
    }
    for (int k = 0; k < K; ++k) {
        unsigned long long best = 0ull;
        bool have = false;
#pragma unroll
        for (int t = 0; t < kTopSlots; ++t) {
            if (!taken[t]) {
                // order-preserving map of a float onto unsigned: flip the sign bit, or all bits of a negative
                unsigned u = __float_as_uint(v[t]);
                u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                const unsigned long long key = ((unsigned long long)u << 32) | (0xFFFFFFFFu - ((unsigned)(t * 64) + lane));
                if (!have || key > best) best = key;
                have = true;
            }
        }
        best = wave_max_u64(have ? best : 0ull);
        const unsigned wslot = 0xFFFFFFFFu - (unsigned)best;
#pragma unroll
        for (int t = 0; t < kTopSlots; ++t)
            if ((unsigned)(t * 64) + lane == wslot) {
                taken[t] = true;
                vals[(size_t)row * K + k] = v[t];
                idx[(size_t)row * K + k] = (int)wslot;
            }
    }
}

int topk_create(const htf_potential_desc *d, TopkDevice **out) {
    HTF_REQUIRE(d->K >= 1 && d->K <= kTopMax, "top-k network: top_neighs %d outside [1, %d]", d->K, kTopMax);
    HTF_REQUIRE(d->H1 >= 1 && d->H1 <= kTopH && d->H2 >= 1 && d->H2 <= kTopH, "top-k network: hidden widths %d, %d outside [1, %d]", d->H1, d->H2, kTopH);
    HTF_REQUIRE(d->W1 && d->b1 && d->W2 && d->b2 && d->W3 && d->b3, "top-k network: null weight pointer");
    HTF_REQUIRE(d->activation == HTF_ACT_LINEAR || d->activation == HTF_ACT_TANH, "top-k network: unknown activation %d", d->activation);
    TopkDevice *m = new (std::nothrow) TopkDevice();
    if (!m) {
        set_error("top-k network: out of host memory");
        return HTF_ERR_NOMEM;
    }
    m->K = d->K; m->H1 = d->H1; m->H2 = d->H2; m->act = d->activation;
    std::vector<float> flat;
    flat.insert(flat.end(), d->W1, d->W1 + (size_t)d->K * d->H1);
    flat.insert(flat.end(), d->b1, d->b1 + d->H1);
    flat.insert(flat.end(), d->W2, d->W2 + (size_t)d->H1 * d->H2);
    flat.insert(flat.end(), d->b2, d->b2 + d->H2);
    flat.insert(flat.end(), d->W3, d->W3 + d->H2);
    flat.push_back(d->b3[0]);
    hipError_t e = hipMalloc((void **)&m->w, flat.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(m->w, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        set_error("top-k network: device copy of the weights failed: %s", hipGetErrorString(e));
        if (m->w) (void)hipFree(m->w);
        delete m;
        return HTF_ERR_DEVICE;
    }
    *out = m;
    return HTF_OK;
}

void topk_destroy(TopkDevice *m) {
    if (!m) return;
    if (m->w) (void)hipFree(m->w);
    delete m;
}

int topk_eval(const TopkDevice *m, const void *nlist, int in_dtype, unsigned B, unsigned NN, void *force,
              int force_dtype, void *virial9, hipStream_t s) {
    HTF_REQUIRE(m, "top-k network: null potential");
    HTF_REQUIRE(NN >= (unsigned)m->K, "top-k network: the neighbor rows hold %u slots, fewer than top_neighs = %d", NN, m->K);
    HTF_REQUIRE(NN <= 64u * kTopSlots, "top-k network: NN %u > %d", NN, 64 * kTopSlots);
    const int out_f64 = force_dtype == HTF_F64;
    const unsigned grid = (B + 3) / 4;
    const size_t lds = (((size_t)m->n_floats() + 3) & ~(size_t)3) * sizeof(float) + 4 * kTopH * sizeof(float);
#define HTF_TK(TANH, VIR, T, V4)                                                                                       \
    hipLaunchKernelGGL((topk_mlp_kernel<TANH, VIR, T>), dim3(grid), dim3(256), lds, s, (const V4 *)nlist, B, NN, force, \
                       virial9, out_f64, m->w, m->K, m->H1, m->H2)
#define HTF_TK2(TANH, VIR)                                                                                             \
    do {                                                                                                               \
        if (in_dtype == HTF_F32) HTF_TK(TANH, VIR, float, float4); else HTF_TK(TANH, VIR, double, double4);            \
    } while (0)
    if (m->act == HTF_ACT_TANH) {
        if (virial9) HTF_TK2(true, true); else HTF_TK2(true, false);
    } else {
        if (virial9) HTF_TK2(false, true); else HTF_TK2(false, false);
    }
#undef HTF_TK2
#undef HTF_TK
    return check_launch("topk_mlp_kernel");
}

} // namespace htf

extern "C" int htf_top_k(const float *d_x, unsigned B, unsigned n, unsigned k, float *d_values, int *d_indices,
                         htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_x && d_values && d_indices, "htf_top_k: null pointer");
    HTF_REQUIRE(k >= 1 && k <= n, "htf_top_k: k = %u outside [1, %u]", k, n);
    HTF_REQUIRE(n <= 64u * kTopSlots, "htf_top_k: rows of %u entries > %d", n, 64 * kTopSlots);
    if (B == 0) return HTF_OK;
    hipLaunchKernelGGL(topk_values_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_x, B, n, (int)k, d_values,
                       d_indices);
    return check_launch("topk_values_kernel");
}
