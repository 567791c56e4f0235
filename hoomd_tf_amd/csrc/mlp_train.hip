// Online force matching for the pair-MLP (FORCE_MODE::hoomd2tf, tensorflowcompute.py:347-370,
// SURVEY 8(f)-1): d(sum of squared residuals)/d(theta) for
//   pred_i = (F_i, E_i),  F_i = sum_j m_ij u'(r_ij) t_ij / r_ij,  E_i = 1/2 sum_j m_ij u(r_ij)
// with u = Dense(1) o act o Dense(H2) o act o Dense(H1) o RBF (layers.py:46-49 + Keras Dense).
// Keras gets this by back-propagating through tf.gradients (a double backward over
// [N, NN, K] / [N, NN, H] tensors).  Here it is ONE sweep over the pair vectors: with
//   a_ij = 2 m (res_i . t_ij) / r_ij,   b_ij = m res_iE,
// the loss gradient is d/dtheta sum_ij (a u' + b u); per pair that is a forward pass of
// the value and of its r-tangent, and ONE reverse pass over both.
//
// MI355X mapping: training runs every `period` steps, so this kernel favours exact fp32
// and simplicity over the matrix cores.  A wave walks the pairs of its rows one after the
// other; lane f owns hidden feature f: column f of W1/W2 and row f of W2 sit in its
// registers (160 VGPRs), activations of the other features arrive by v_readlane
// broadcasts (SGPR operands of the FMAs), and the weight gradients are outer-product
// accumulators dW1[k][f], dW2[f1][f] held in registers (96 VGPRs) for the whole kernel:
// no LDS, no atomics, one partial per wave, reduced in a fixed order (deterministic).
// ~900 VALU instructions per live pair; 4 waves per CU (one per SIMD, ~300 VGPRs).
#include "htf_common.h"
#include "htf_internal.h"
#include "pair_mlp.h"

namespace htf {

struct MlpDims {
    int K, H1, H2, oB1, oW2, oB2, oW3, oB3;
};

template <bool TANH>
__device__ __forceinline__ float act_val(float z) {
    if constexpr (!TANH) return z;
    float e = __expf(-2.0f * fabsf(z));
    float t = __fdividef(1.0f - e, 1.0f + e);
    return copysignf(t, z);
}

__device__ __forceinline__ float bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <bool TANH, typename IT>
__global__ __launch_bounds__(256, 1) void mlp_grad_kernel(const typename Vec4<IT>::type *__restrict__ nlist, unsigned B,
                                                          unsigned NN, const void *__restrict__ labels, int lab_f64,
                                                          const float4 *__restrict__ pred,
                                                          const float *__restrict__ theta, MlpDims dm,
                                                          const float *__restrict__ tab_c, float ginv,
                                                          float *__restrict__ partial, unsigned stride) {
    const int f = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned nwaves = (gridDim.x * blockDim.x) >> 6;
    const bool in1 = f < dm.H1, in2 = f < dm.H2;

    float w1c[kK], w2c[kH], w2r[kH];
#pragma unroll
    for (int k = 0; k < kK; ++k) w1c[k] = (k < dm.K && in1) ? theta[k * dm.H1 + f] : 0.f;
#pragma unroll
    for (int a = 0; a < kH; ++a) {
        w2c[a] = (a < dm.H1 && in2) ? theta[dm.oW2 + a * dm.H2 + f] : 0.f;
        w2r[a] = (in1 && a < dm.H2) ? theta[dm.oW2 + f * dm.H2 + a] : 0.f;
    }
    const float b1f = in1 ? theta[dm.oB1 + f] : 0.f;
    const float b2f = in2 ? theta[dm.oB2 + f] : 0.f;
    const float w3f = in2 ? theta[dm.oW3 + f] : 0.f;
    // centre of RBF k = f & 31 from the operand-ordered table (pair_mlp.h: feature f0(v) + 4 h)
    const int kf = f & 31;
    const float cen = tab_c[((kf >> 2) & 1) * 16 + (kf & 3) + 4 * (kf >> 3)];

    float g1[kK], g2[kH];
#pragma unroll
    for (int k = 0; k < kK; ++k) g1[k] = 0.f;
#pragma unroll
    for (int a = 0; a < kH; ++a) g2[a] = 0.f;
    float gb1 = 0.f, gb2 = 0.f, gw3 = 0.f, gb3 = 0.f, loss = 0.f;

    for (unsigned row = wave; row < B; row += nwaves) {
        const float4 pr = pred[row];
        float lx, ly, lz, lw;
        if (lab_f64) {
            const double4 l = ((const double4 *)labels)[row];
            lx = (float)l.x; ly = (float)l.y; lz = (float)l.z; lw = (float)l.w;
        } else {
            const float4 l = ((const float4 *)labels)[row];
            lx = l.x; ly = l.y; lz = l.z; lw = l.w;
        }
        const float rx = pr.x - lx, ry = pr.y - ly, rz = pr.z - lz, re = pr.w - lw;
        loss += rx * rx + ry * ry + rz * rz + re * re;
        const typename Vec4<IT>::type *rp = nlist + (size_t)row * NN;
        for (unsigned j = 0; j < NN; ++j) {
            const auto v = rp[j];
            const float tx = (float)v.x + kNormDelta, ty = (float)v.y + kNormDelta, tz = (float)v.z + kNormDelta;
            const float r = sqrtf(tx * tx + ty * ty + tz * tz);
            if (!(r > kRinvDelta)) continue; // padded slot (wave-uniform)
            const float a = 2.0f * (rx * tx + ry * ty + rz * tz) / r;
            const float b = re;
            const float d = r - cen;
            const float phi = __expf(-d * d * ginv);
            const float dphi = -2.0f * d * ginv * phi;
            // value and r-tangent, layer 1
            float z1 = b1f, zd1 = 0.f;
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                z1 = fmaf(bcast(phi, k), w1c[k], z1);
                zd1 = fmaf(bcast(dphi, k), w1c[k], zd1);
            }
            const float h1 = act_val<TANH>(z1);
            const float s1 = TANH ? 1.0f - h1 * h1 : 1.0f;
            const float hd1 = s1 * zd1;
            float z2 = b2f, zd2 = 0.f;
#pragma unroll
            for (int k = 0; k < kH; ++k) {
                z2 = fmaf(bcast(h1, k), w2c[k], z2);
                zd2 = fmaf(bcast(hd1, k), w2c[k], zd2);
            }
            const float h2 = act_val<TANH>(z2);
            const float s2 = TANH ? 1.0f - h2 * h2 : 1.0f;
            const float hd2 = s2 * zd2;
            // reverse over S = a u' + b u  (u = w3 . h2 + b3, u' = w3 . hd2)
            gw3 += b * h2 + a * hd2;
            gb3 += b;
            const float hb2 = b * w3f, hdb2 = a * w3f;
            const float c2 = TANH ? -2.0f * h2 * s2 : 0.f; // act''(z2)
            const float zb2 = hb2 * s2 + hdb2 * c2 * zd2;
            const float zdb2 = hdb2 * s2;
            gb2 += zb2;
            float hb1 = 0.f, hdb1 = 0.f;
#pragma unroll
            for (int k = 0; k < kH; ++k) {
                hb1 = fmaf(w2r[k], bcast(zb2, k), hb1);
                hdb1 = fmaf(w2r[k], bcast(zdb2, k), hdb1);
            }
            const float c1 = TANH ? -2.0f * h1 * s1 : 0.f;
            const float zb1 = hb1 * s1 + hdb1 * c1 * zd1;
            const float zdb1 = hdb1 * s1;
            gb1 += zb1;
            // outer products: dW2[f1][f] += h1[f1] zb2[f] + hd1[f1] zdb2[f];  dW1[k][f] likewise
#pragma unroll
            for (int k = 0; k < kH; ++k) g2[k] = fmaf(bcast(h1, k), zb2, fmaf(bcast(hd1, k), zdb2, g2[k]));
#pragma unroll
            for (int k = 0; k < kK; ++k) g1[k] = fmaf(bcast(phi, k), zb1, fmaf(bcast(dphi, k), zdb1, g1[k]));
        }
    }
    float *out = partial + (size_t)wave * stride;
    if (f == 0) {
        out[0] = loss;
        out[1 + dm.oB3] = gb3;
    }
#pragma unroll
    for (int k = 0; k < kK; ++k)
        if (k < dm.K && in1) out[1 + k * dm.H1 + f] = g1[k];
#pragma unroll
    for (int a = 0; a < kH; ++a)
        if (a < dm.H1 && in2) out[1 + dm.oW2 + a * dm.H2 + f] = g2[a];
    if (in1) out[1 + dm.oB1 + f] = gb1;
    if (in2) {
        out[1 + dm.oB2 + f] = gb2;
        out[1 + dm.oW3 + f] = gw3;
    }
}

// accum[c] = sum over waves, fixed order
__global__ void mlp_reduce_partials_kernel(const float *__restrict__ partial, unsigned nwaves, unsigned stride,
                                           unsigned ncols, float *__restrict__ accum) {
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    float s = 0.f;
    for (unsigned w = 0; w < nwaves; ++w) s += partial[(size_t)w * stride + c];
    accum[c] = s;
}

static unsigned train_waves(const MlpDevice *m, unsigned B) {
    unsigned w = (unsigned)m->n_cu * 4u;
    if (w > B) w = (B + 3u) & ~3u;
    return w;
}

static unsigned train_stride(const MlpDevice *m) { return ((unsigned)m->num_params() + 1u + 3u) & ~3u; }

size_t mlp_train_scratch_floats(const MlpDevice *m, unsigned B) {
    if (!m) return 0;
    return (size_t)train_waves(m, B) * train_stride(m) + (size_t)B * 4;
}

int mlp_train_grad(const MlpDevice *m, const void *nlist, int in_dtype, unsigned B, unsigned NN, const void *labels,
                   int lab_f64, void *pred, float *accum, float *scratch, hipStream_t stream) {
    HTF_REQUIRE(m, "pair-MLP: null potential");
    const unsigned nw = train_waves(m, B), stride = train_stride(m);
    float *partial = scratch;
    float4 *predbuf = pred ? (float4 *)pred : (float4 *)(scratch + (size_t)nw * stride);
    int rc = mlp_eval(m, nlist, in_dtype, B, NN, predbuf, HTF_F32, stream);
    if (rc != HTF_OK) return rc;
    MlpDims dm{m->K, m->H1, m->H2, m->off_b1(), m->off_W2(), m->off_b2(), m->off_W3(), m->off_b3()};
    const float *tab_c = m->images + (m->precision == HTF_MLP_BF16 ? Img<true>::TabC : Img<false>::TabC);
    const float ginv = 1.0f / m->gap;
    const dim3 grid(nw / 4), block(256);
    const bool th = m->act == HTF_ACT_TANH;
#define HTF_LAUNCH_MLPG(T, IT, V4)                                                                                     \
    hipLaunchKernelGGL((mlp_grad_kernel<T, IT>), grid, block, 0, stream, (const V4 *)nlist, B, NN, labels, lab_f64,    \
                       predbuf, m->theta, dm, tab_c, ginv, partial, stride)
    if (in_dtype == HTF_F32) {
        if (th) HTF_LAUNCH_MLPG(true, float, float4); else HTF_LAUNCH_MLPG(false, float, float4);
    } else {
        if (th) HTF_LAUNCH_MLPG(true, double, double4); else HTF_LAUNCH_MLPG(false, double, double4);
    }
#undef HTF_LAUNCH_MLPG
    rc = check_launch("mlp_grad_kernel");
    if (rc != HTF_OK) return rc;
    const unsigned ncols = (unsigned)m->num_params() + 1u;
    hipLaunchKernelGGL(mlp_reduce_partials_kernel, dim3((ncols + 63) / 64), dim3(64), 0, stream, partial, nw, stride,
                       ncols, accum);
    return check_launch("mlp_reduce_partials_kernel");
}

} // namespace htf
