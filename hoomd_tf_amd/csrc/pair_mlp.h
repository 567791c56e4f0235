// Shared definitions of the pair-MLP potential: operand-image layout (pair_mlp.hip), the
// trainable flat parameter vector and the map between the two (mlp_train.hip).
#pragma once
#include "htf_common.h"
#include "htf_internal.h"

namespace htf {

constexpr int kK = 32;  // RBF count (padded with zero weights below 32)
constexpr int kH = 64;  // hidden width (padded with zero weights below 64)

// Offsets (in floats) inside the device image buffer / LDS.  One operand block covers a
// 32 x 32 (feature x feature) weight tile: fp32 [g 4][lane 64][4 floats] = 1024 floats,
// bf16 [s 2][lane 64][8 bf16] = 512 floats; split (HTF_MLP_SPLIT) three such bf16 blocks
// [part 3: hi, mid, lo][s 2][lane 64][8 bf16] = 1536 floats.  P = htf_mlp_precision.
template <int P>
struct Img {
    static constexpr int BS = P == 0 ? 1024 : (P == 1 ? 512 : (P == 2 ? 1536 : 1024)); // split16: [part 2: hi, lo][s 2][lane 64][8 fp16]
    static constexpr int L1 = 0;            // [nb 2]
    static constexpr int L2 = 2 * BS;       // [nb 2][kb 2]
    static constexpr int B2 = 6 * BS;       // [fb 2][kb 2]
    static constexpr int B1 = 10 * BS;      // [kb 2]
    static constexpr int TabB1 = 12 * BS;   // [b 2][h 2][v 16]
    static constexpr int TabB2 = TabB1 + 64;
    static constexpr int TabW3 = TabB2 + 64;
    static constexpr int TabC = TabW3 + 64; // [h 2][v 16] RBF centres
    static constexpr int TabB3 = TabC + 32; // output bias (+3 pad floats)
    static constexpr int Floats = TabB3 + 4; // fp32: 12516 floats (50 KB); bf16: 6372 (25 KB); split: 18660 (75 KB); split16: 12516 (50 KB)
};

// Every weight-carrying element of an image is theta[map[e]] (or 0 when map[e] < 0):
// e < kMapW indexes the operand blocks element-wise (fp32 float / bf16 half, both 12288; the split
// images use the bf16 order, element e of block b landing in each of the block's three parts),
// then 192 table floats (TabB1, TabB2, TabW3) and the output bias.
constexpr int kMapW = 12 * 1024;
constexpr int kMapT = 192;
constexpr int kMapN = kMapW + kMapT + 1;

// theta, flat, Keras get_weights() order: W1 [K][H1] | b1 [H1] | W2 [H1][H2] | b2 [H2] | W3 [H2] | b3
struct MlpDevice {
    float *images = nullptr;      // Img<>::Floats floats, operand order
    int *map = nullptr;           // kMapN ints
    float *eval_images = nullptr; // split16 + tanh: the EVALUATOR's images in u-form (pair_mlp.hip kUForm); == images otherwise
    float *train_images = nullptr; // fp32 operand images for the training sweep (== images unless bf16)
    int *train_map = nullptr;
    const float *theta = nullptr; // device parameter vector the images are built from
    float *own_theta = nullptr;   // ... owned copy unless the caller supplied d_theta
    int *range_flag = nullptr;    // split16: host-mapped word the image build sets when a weight left fp16's range (cleared by the next build)
    float centers[kK];            // float32 linspace(low, high, K), 0 beyond K
    float gap = 1.f;
    int K = 0, H1 = 0, H2 = 0;
    int act = HTF_ACT_LINEAR;
    int precision = HTF_MLP_FP32;
    int n_cu = 256;
    int num_params() const { return K * H1 + H1 + H1 * H2 + H2 + H2 + 1; }
    int off_b1() const { return K * H1; }
    int off_W2() const { return K * H1 + H1; }
    int off_b2() const { return off_W2() + H1 * H2; }
    int off_W3() const { return off_b2() + H2; }
    int off_b3() const { return off_W3() + H2; }
};

__host__ __device__ constexpr int f0(int v) { return (v & 3) + 8 * (v >> 2); }


// ---- device helpers shared by the evaluator and the training kernel
using f32x16 = __attribute__((ext_vector_type(16))) float;


template <bool TANH>
__device__ __forceinline__ float act_fwd(float z) {
    if constexpr (!TANH) return z;
    // tanh(z) = 1 - 2 / (1 + exp(2z)): v_mul, v_exp_f32, v_add, v_rcp_f32 (1 ulp), v_fma -- five
    // instructions (the sign-symmetric (1 - e) / (1 + e) form needs nine; a/b would expand to the
    // 10-instruction IEEE division).  exp overflow -> rcp(inf) = 0 -> +1; underflow -> -1.
    // Absolute error ~2e-7, the same cancellation near 0 as the symmetric form.
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * z)), 1.0f);
}

#define HTF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x16 load_tab(const float *tab, int b, unsigned h) {
    const float4 *p = reinterpret_cast<const float4 *>(tab + (b * 2 + h) * 16);
    float4 a = p[0], bq = p[1], c = p[2], d = p[3];
    f32x16 r;
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w;
    r[4] = bq.x; r[5] = bq.y; r[6] = bq.z; r[7] = bq.w;
    r[8] = c.x; r[9] = c.y; r[10] = c.z; r[11] = c.w;
    r[12] = d.x; r[13] = d.y; r[14] = d.z; r[15] = d.w;
    return r;
}


// ---- shared by the training sweeps (mlp_train.hip: fp32 MFMA; mlp_train16.hip: fp16 pipeline)
struct MlpDims {
    int K, H1, H2, oB1, oW2, oB2, oW3, oB3;
};

__device__ __forceinline__ f32x16 zero16() {
    f32x16 r;
#pragma unroll
    for (int v = 0; v < 16; ++v) r[v] = 0.f;
    return r;
}

__device__ __forceinline__ float4 residual(const float4 *pred, const void *labels, int lab_f64, unsigned row) {
    const float4 pr = pred[row];
    if (lab_f64) {
        const double4 l = ((const double4 *)labels)[row];
        return make_float4(pr.x - (float)l.x, pr.y - (float)l.y, pr.z - (float)l.z, pr.w - (float)l.w);
    }
    const float4 l = ((const float4 *)labels)[row];
    return make_float4(pr.x - l.x, pr.y - l.y, pr.z - l.z, pr.w - l.w);
}

// Reverse seeds are scaled by a power of two S chosen per launch from the largest residual (mlp_resid_max_kernel) so that the
// largest |seed| sits in [1, 2): everything downstream of the seeds is linear in them and travels as fp16 hi + lo -- large
// residuals (early training, close contacts) would leave fp16's range, small ones its normal numbers.  The
// accumulated gradient is multiplied by 1 / S, exactly, when the block partials are combined.
__device__ __forceinline__ float seed_scale(float m) {
    if (!(m > 0.f) || !(m < 3.0e38f)) return 1.0f;
    int e = 0;
    (void)frexpf(m, &e); // m = f 2^e, f in [0.5, 1)
    e = 1 - e;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return ldexpf(1.0f, e);
}

// The split16 sweep takes its rows in kTrainWindows launches by the size of their residual (windows of 2^kWindowBits below the
// launch's largest, the last one open-ended), each with its own seed scale -- mlp_train16.hip.
constexpr int kTrainWindows = 4;
constexpr int kWindowBits = 6;

// split16 potentials: prediction (evaluator), the launch's largest residual, the sweep; block partials land in `partial`
// (kTrainWindows x nblk_out of them, window by window), window w still scaled by seed_scale(*resid_max) 2^(kWindowBits w)
int mlp_train_grad16(const MlpDevice *m, const void *nlist, int in_dtype, unsigned B, unsigned NN, const void *labels, int lab_f64,
                     float4 *predbuf, float *partial, unsigned stride, float *resid_max, unsigned *nblk_out, hipStream_t stream);
int mlp_refresh(const MlpDevice *m, hipStream_t stream);
bool mlp_out_of_range(const MlpDevice *m);
int mlp_train_grad(const MlpDevice *m, const void *nlist, int in_dtype, unsigned B, unsigned NN, const void *labels,
                   int lab_f64, void *pred, float *accum, float *scratch, hipStream_t stream);
size_t mlp_train_scratch_floats(const MlpDevice *m, unsigned B, unsigned NN);

} // namespace htf
