// Observables and small layers of the path:
// compute_rdf / masked_nlist, RBFExpansion, EDSLayer, wrap_vector.
#include <cmath>

#include "htf_common.h"
#include "htf_internal.h"

namespace htf {

// ---- compute_rdf: privatised LDS histogram per block, one global atomic per bin per block
constexpr unsigned kMaxBins = 2048;

template <typename IT>
__global__ __launch_bounds__(256) void rdf_hist_kernel(const typename Vec4<IT>::type *__restrict__ nlist, unsigned B,
                                                       unsigned NN, float r0, float r1, unsigned nb,
                                                       const float *__restrict__ type_tensor, unsigned type_stride,
                                                       int type_i, int type_j, unsigned *__restrict__ hist) {
    __shared__ unsigned sh[kMaxBins];
    for (unsigned i = threadIdx.x; i < nb; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    const float width = r1 - r0;
    const size_t total = (size_t)B * NN;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned n_lo = 0, n_hi = 0; // end bins (all padded slots land in bin 0) counted in registers
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) {
        const unsigned row = (unsigned)(s / NN);
        // masked_nlist type_i: boolean_mask drops whole rows (simmodel.py:684-686)
        if (type_i >= 0 && type_tensor[(size_t)row * type_stride] != (float)type_i) continue;
        auto v = load_stream(&nlist[s]);
        float x = (float)v.x, y = (float)v.y, z = (float)v.z;
        // masked_nlist type_j: nlist * mask zeroes the slot (simmodel.py:687-691)
        if (type_j >= 0 && (float)v.w != (float)type_j) x = y = z = 0.f;
        const float r = plain_norm3(x, y, z);
        // tf.histogram_fixed_width: floor(nbins * (v - lo) / (hi - lo)) clipped to [0, nbins-1]
        float fi = floorf((float)nb * ((r - r0) / width));
        int idx = fi < 0.f ? 0 : (fi > (float)(nb - 1) ? (int)(nb - 1) : (int)fi);
        if (idx == 0) ++n_lo;
        else if (idx == (int)nb - 1) ++n_hi;
        else atomicAdd(&sh[idx], 1u);
    }
    n_lo = group_sum_u<64>(n_lo);
    n_hi = group_sum_u<64>(n_hi);
    if ((threadIdx.x & 63) == 0) {
        if (n_lo) atomicAdd(&sh[0], n_lo);
        if (n_hi) atomicAdd(&sh[nb - 1], n_hi);
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < nb; i += blockDim.x)
        if (sh[i]) atomicAdd(&hist[i], sh[i]);
}

__global__ void rdf_finalize_kernel(const unsigned *__restrict__ hist, unsigned nbins, float r0, float r1,
                                    float *__restrict__ rdf, float *__restrict__ rs) {
    unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbins) return;
    // shell = linspace(r0, r1, nbins + 1) evaluated in double then rounded to fp32 (as numpy does)
    const double step = ((double)r1 - (double)r0) / (double)nbins;
    const float lo = (float)((double)r0 + b * step);
    const float hi = (b + 1 == nbins) ? r1 : (float)((double)r0 + (b + 1) * step);
    const float vol = hi * hi * hi - lo * lo * lo;
    rdf[b] = (float)hist[b + 1] / vol;
    rs[b] = (hi + lo) * 0.5f;
}

// ---- RBFExpansion
__global__ __launch_bounds__(256) void rbf_kernel(const float *__restrict__ x, size_t n, float low, double step,
                                                  float high, unsigned count, float gap,
                                                  float *__restrict__ out) {
    const size_t total = n * count;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const size_t i = e / count;
        const unsigned k = (unsigned)(e - i * count);
        const float c = (k + 1 == count) ? high : (float)((double)low + k * step);
        const float d = x[i] - c;
        out[e] = expf(-(d * d) / gap);
    }
}

// ---- EDSLayer (layers.py:159-195) + tf.compat.v1.train.AdamOptimizer
__global__ void eds_kernel(float *__restrict__ st, const float *__restrict__ cvp, float set_point, int period,
                           float lr, float cv_scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float mean = st[0], ssd = st[1], alpha = st[2], am = st[3], av = st[4];
    int n = (int)st[5], t = (int)st[6];
    const float cv = *cvp;
    const float reset = n != 0 ? 1.f : 0.f; // reset statistics if n is 0
    mean *= reset;
    ssd *= reset;
    const float um = n > period / 2 ? 1.f : 0.f;
    const float delta = (cv - mean) * um;
    const float den = (float)(n - period / 2);
    mean += den == 0.f ? 0.f : delta / den; // divide_no_nan
    ssd += delta * (cv - mean);
    if (n == period - 1) {
        const float grad = -2.f * (mean - set_point) * ssd / (float)period / 2.f / cv_scale;
        t += 1;
        const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
        const float lr_t = (float)((double)lr * sqrt(1.0 - pow(b2, (double)t)) / (1.0 - pow(b1, (double)t)));
        am += (grad - am) * (float)(1.0 - b1);
        av += (grad * grad - av) * (float)(1.0 - b2);
        alpha -= lr_t * am / (sqrtf(av) + (float)eps);
    }
    n = (n + 1) % period;
    st[0] = mean; st[1] = ssd; st[2] = alpha; st[3] = am; st[4] = av;
    st[5] = (float)n; st[6] = (float)t;
}

template <typename T>
__global__ __launch_bounds__(256) void wrap_kernel(const T *__restrict__ r, size_t n3, T bx, T by, T bz,
                                                   T *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    const unsigned c = (unsigned)(i % 3);
    const T bs = c == 0 ? bx : (c == 1 ? by : bz);
    const T v = r[i];
    out[i] = v - rint(v / bs) * bs;
}

} // namespace htf

using namespace htf;

extern "C" int htf_rdf_histogram(const void *d_nlist, int nlist_dtype, unsigned B, unsigned NN, float r0, float r1,
                                 unsigned nbins_total, const float *d_type_tensor, unsigned type_stride, int type_i,
                                 int type_j, unsigned *d_hist, htf_stream stream) {
    HTF_REQUIRE(d_nlist && d_hist, "htf_rdf_histogram: null pointer");
    HTF_REQUIRE(nlist_dtype == HTF_F32 || nlist_dtype == HTF_F64, "htf_rdf_histogram: bad dtype %d", nlist_dtype);
    HTF_REQUIRE(nbins_total >= 3 && nbins_total <= kMaxBins, "htf_rdf_histogram: nbins + 2 = %u outside [3, %u]", nbins_total, kMaxBins);
    HTF_REQUIRE(r1 > r0, "htf_rdf_histogram: empty r_range");
    HTF_REQUIRE(type_i < 0 || d_type_tensor, "htf_rdf_histogram: type_i needs a type tensor");
    if (B == 0 || NN == 0) return HTF_OK;
    size_t total = (size_t)B * NN;
    unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (nlist_dtype == HTF_F32)
        hipLaunchKernelGGL((rdf_hist_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4 *)d_nlist, B, NN, r0, r1, nbins_total, d_type_tensor, type_stride, type_i, type_j, d_hist);
    else
        hipLaunchKernelGGL((rdf_hist_kernel<double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const double4 *)d_nlist, B, NN, r0, r1, nbins_total, d_type_tensor, type_stride, type_i, type_j, d_hist);
    return check_launch("rdf_hist_kernel");
}

extern "C" int htf_rdf_finalize(const unsigned *d_hist, unsigned nbins, float r0, float r1, float *d_rdf, float *d_rs,
                                htf_stream stream) {
    HTF_REQUIRE(d_hist && d_rdf && d_rs, "htf_rdf_finalize: null pointer");
    HTF_REQUIRE(nbins >= 1, "htf_rdf_finalize: nbins must be >= 1");
    hipLaunchKernelGGL(rdf_finalize_kernel, dim3((nbins + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_hist, nbins, r0, r1, d_rdf, d_rs);
    return check_launch("rdf_finalize_kernel");
}

extern "C" int htf_rbf_expansion(const float *d_x, size_t n, double low, double high, unsigned count, float *d_out,
                                 htf_stream stream) {
    HTF_REQUIRE(d_x && d_out, "htf_rbf_expansion: null pointer");
    HTF_REQUIRE(count >= 2, "htf_rbf_expansion: count must be >= 2");
    if (n == 0) return HTF_OK;
    const double step = (high - low) / (double)(count - 1);
    const float c0 = (float)low, c1 = (count == 2) ? (float)high : (float)(low + step);
    const float gap = c1 - c0;
    size_t total = n * count;
    unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(rbf_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_x, n, (float)low, step, (float)high, count, gap, d_out);
    return check_launch("rbf_kernel");
}

extern "C" int htf_eds_update(float *d_state, const float *d_cv, float set_point, int period, float learning_rate,
                              float cv_scale, htf_stream stream) {
    HTF_REQUIRE(d_state && d_cv, "htf_eds_update: null pointer");
    HTF_REQUIRE(period >= 1, "htf_eds_update: period must be >= 1");
    hipLaunchKernelGGL(eds_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_state, d_cv, set_point, period, learning_rate, cv_scale);
    return check_launch("eds_kernel");
}

extern "C" int htf_wrap_vector(const void *d_r, int dtype, size_t n, const htf_box *box, void *d_out, htf_stream stream) {
    HTF_REQUIRE(d_r && d_out && box, "htf_wrap_vector: null pointer");
    if (n == 0) return HTF_OK;
    size_t n3 = n * 3;
    unsigned grid = (unsigned)((n3 + 255) / 256);
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((wrap_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float *)d_r, n3, (float)box->hi[0] - (float)box->lo[0], (float)box->hi[1] - (float)box->lo[1], (float)box->hi[2] - (float)box->lo[2], (float *)d_out);
    else if (dtype == HTF_F64)
        hipLaunchKernelGGL((wrap_kernel<double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const double *)d_r, n3, box->hi[0] - box->lo[0], box->hi[1] - box->lo[1], box->hi[2] - box->lo[2], (double *)d_out);
    else {
        set_error("htf_wrap_vector: bad dtype %d", dtype);
        return HTF_ERR_INVALID;
    }
    return check_launch("wrap_kernel");
}
