// Small companions of the force path: the reference's bookkeeping kernels, kept
// because HOOMD-facing semantics depend on them (accumulating virial fold-in,
// reference-force summation, type un-stuffing).  All are trivially HBM-bound and
// touch O(N) bytes (<= 4.5 MiB at N = 131072) -- three orders below the nlist.
#include "htf_common.h"

namespace htf {

// TensorflowCompute.cu:41-55 htf_gpu_add_virial_kernel / .cc:284-301 receiveVirial
template <typename T>
__global__ __launch_bounds__(256) void add_virial_kernel(T *__restrict__ dest, const T *__restrict__ src,
                                                         unsigned N, size_t pitch) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const T *s = src + (size_t)i * 9;
    dest[0 * pitch + i] += s[0]; // xx
    dest[1 * pitch + i] += s[1]; // xy
    dest[2 * pitch + i] += s[2]; // xz
    dest[3 * pitch + i] += s[4]; // yy
    dest[4 * pitch + i] += s[5]; // yz
    dest[5 * pitch + i] += s[8]; // zz
}

// TensorflowCompute.cu:11-23 htf_gpu_add_scalar4_kernel
template <typename V>
__global__ __launch_bounds__(256) void add_scalar4_kernel(V *__restrict__ dest, const V *__restrict__ src, unsigned N) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    V d = dest[i], s = src[i];
    d.x += s.x;
    d.y += s.y;
    d.z += s.z;
    d.w += s.w;
    dest[i] = d;
}

__device__ __forceinline__ int w_as_int(float w) { return __float_as_int(w); }
__device__ __forceinline__ int w_as_int(double w) { return (int)(__double_as_longlong(w) & 0xffffffffll); }

// TFArrayComm.h:86-130 receiveArray + TFArrayComm.cu:9-15 htf_gpu_unstuff4_kerenl
template <typename ST, typename DT>
__global__ __launch_bounds__(256) void copy_positions_kernel(typename Vec4<DT>::type *__restrict__ dest,
                                                             const typename Vec4<ST>::type *__restrict__ src,
                                                             unsigned offset, unsigned N, int unstuff4) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    auto s = src[offset + i];
    typename Vec4<DT>::type d;
    d.x = (DT)s.x;
    d.y = (DT)s.y;
    d.z = (DT)s.z;
    d.w = unstuff4 ? (DT)w_as_int(s.w) : (DT)s.w;
    dest[i] = d;
}

} // namespace htf

extern "C" int htf_add_virial(void *d_dest, const void *d_src9, int dtype, unsigned N, size_t pitch, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_dest && d_src9, "htf_add_virial: null pointer");
    HTF_REQUIRE(pitch >= N, "htf_add_virial: pitch %zu < N %u", pitch, N);
    if (N == 0) return HTF_OK;
    unsigned grid = (N + 255) / 256;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((add_virial_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (float *)d_dest, (const float *)d_src9, N, pitch);
    else if (dtype == HTF_F64)
        hipLaunchKernelGGL((add_virial_kernel<double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (double *)d_dest, (const double *)d_src9, N, pitch);
    else {
        set_error("htf_add_virial: bad dtype %d", dtype);
        return HTF_ERR_INVALID;
    }
    return check_launch("add_virial_kernel");
}

extern "C" int htf_add_scalar4(void *d_dest, const void *d_src, int dtype, unsigned N, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_dest && d_src, "htf_add_scalar4: null pointer");
    if (N == 0) return HTF_OK;
    unsigned grid = (N + 255) / 256;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((add_scalar4_kernel<float4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (float4 *)d_dest, (const float4 *)d_src, N);
    else if (dtype == HTF_F64)
        hipLaunchKernelGGL((add_scalar4_kernel<double4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (double4 *)d_dest, (const double4 *)d_src, N);
    else {
        set_error("htf_add_scalar4: bad dtype %d", dtype);
        return HTF_ERR_INVALID;
    }
    return check_launch("add_scalar4_kernel");
}

namespace htf {
// dest.xyz = src.xyz, dest.w (HOOMD's stuffed type) untouched
template <typename TS, typename TD>
__global__ void copy3_kernel(typename Vec4<TD>::type *__restrict__ dest, const typename Vec4<TS>::type *__restrict__ src,
                             unsigned N) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const auto v = src[i];
    dest[i].x = (TD)v.x;
    dest[i].y = (TD)v.y;
    dest[i].z = (TD)v.z;
}
} // namespace htf

namespace htf {
// sum_i force[i].w in double, one block, fixed order (ForceCompute::calcEnergySum)
template <typename V4>
__global__ __launch_bounds__(1024) void energy_sum_kernel(const V4 *__restrict__ force, unsigned N, double *__restrict__ out) {
    __shared__ double part[16];
    double s = 0.0;
    for (unsigned i = threadIdx.x; i < N; i += blockDim.x) s += (double)force[i].w;
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (unsigned w = 0; w < blockDim.x / 64; ++w) t += part[w];
        *out = t;
    }
}
} // namespace htf

extern "C" int htf_energy_sum(const void *d_force, int dtype, unsigned N, double *d_out, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_force && d_out, "htf_energy_sum: null pointer");
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((energy_sum_kernel<float4>), dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float4 *)d_force, N, d_out);
    else if (dtype == HTF_F64)
        hipLaunchKernelGGL((energy_sum_kernel<double4>), dim3(1), dim3(1024), 0, (hipStream_t)stream, (const double4 *)d_force, N, d_out);
    else {
        set_error("htf_energy_sum: bad dtype %d", dtype);
        return HTF_ERR_INVALID;
    }
    return check_launch("energy_sum_kernel");
}

extern "C" int htf_copy3(void *d_dest, int dest_dtype, const void *d_src, int src_dtype, unsigned N, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_dest && d_src, "htf_copy3: null pointer");
    if (N == 0) return HTF_OK;
    const unsigned grid = (N + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == HTF_F32 && dest_dtype == HTF_F32)
        hipLaunchKernelGGL((copy3_kernel<float, float>), dim3(grid), dim3(256), 0, s, (float4 *)d_dest, (const float4 *)d_src, N);
    else if (src_dtype == HTF_F64 && dest_dtype == HTF_F32)
        hipLaunchKernelGGL((copy3_kernel<double, float>), dim3(grid), dim3(256), 0, s, (float4 *)d_dest, (const double4 *)d_src, N);
    else if (src_dtype == HTF_F64 && dest_dtype == HTF_F64)
        hipLaunchKernelGGL((copy3_kernel<double, double>), dim3(grid), dim3(256), 0, s, (double4 *)d_dest, (const double4 *)d_src, N);
    else if (src_dtype == HTF_F32 && dest_dtype == HTF_F64)
        hipLaunchKernelGGL((copy3_kernel<float, double>), dim3(grid), dim3(256), 0, s, (double4 *)d_dest, (const float4 *)d_src, N);
    else {
        set_error("htf_copy3: bad dtype (%d, %d)", src_dtype, dest_dtype);
        return HTF_ERR_INVALID;
    }
    return check_launch("copy3_kernel");
}

extern "C" int htf_copy_positions(void *d_dest, int dest_dtype, const void *d_src, int src_dtype, unsigned offset,
                                  unsigned N, int unstuff4, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_dest && d_src, "htf_copy_positions: null pointer");
    if (N == 0) return HTF_OK;
    unsigned grid = (N + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == HTF_F32 && dest_dtype == HTF_F32)
        hipLaunchKernelGGL((copy_positions_kernel<float, float>), dim3(grid), dim3(256), 0, s, (float4 *)d_dest, (const float4 *)d_src, offset, N, unstuff4);
    else if (src_dtype == HTF_F64 && dest_dtype == HTF_F32)
        hipLaunchKernelGGL((copy_positions_kernel<double, float>), dim3(grid), dim3(256), 0, s, (float4 *)d_dest, (const double4 *)d_src, offset, N, unstuff4);
    else if (src_dtype == HTF_F64 && dest_dtype == HTF_F64)
        hipLaunchKernelGGL((copy_positions_kernel<double, double>), dim3(grid), dim3(256), 0, s, (double4 *)d_dest, (const double4 *)d_src, offset, N, unstuff4);
    else if (src_dtype == HTF_F32 && dest_dtype == HTF_F64)
        hipLaunchKernelGGL((copy_positions_kernel<float, double>), dim3(grid), dim3(256), 0, s, (double4 *)d_dest, (const float4 *)d_src, offset, N, unstuff4);
    else {
        set_error("htf_copy_positions: bad dtype (%d, %d)", src_dtype, dest_dtype);
        return HTF_ERR_INVALID;
    }
    return check_launch("copy_positions_kernel");
}

namespace htf {
// compute_positions_forces (simmodel.py:492-506) for a per-particle radial energy of the positions row,
// e_i = coef * |p_i|^power over the first `ncomp` columns (tf.norm(positions, axis=1) takes all four, the
// un-stuffed type included: build_examples.py:59-64 BenchmarkNonlistModel, e = divide_no_nan(1., |p|)):
// force_i = -d(sum e)/d p_i[:3] = -coef * power * |p_i|^(power - 2) * p_i[:3], energy column = e_i.
// |p| = 0: divide_no_nan gives e = 0 for negative powers, and TensorFlow's norm gradient is 0/0 there; the
// kernel returns zeros (a particle exactly at the origin with type 0 -- measure zero).
template <typename T>
__global__ __launch_bounds__(256) void positions_radial_kernel(const typename Vec4<T>::type *__restrict__ pos, unsigned N,
                                                               int ncomp, int power, float coef,
                                                               void *__restrict__ force, int out_f64) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const auto p = pos[i];
    const float x = (float)p.x, y = (float)p.y, z = (float)p.z, w = ncomp == 4 ? (float)p.w : 0.f;
    const float n2 = x * x + y * y + z * z + w * w;
    float e = 0.f, c = 0.f;
    if (n2 > 0.f) {
        const float n = sqrtf(n2);
        // |p|^power and |p|^(power - 2) by repeated multiplication of n or 1 / n (power is a small integer)
        const float b = power >= 0 ? n : 1.0f / n;
        float pw = 1.f;
        for (int k = 0, a = power >= 0 ? power : -power; k < a; ++k) pw *= b;
        e = coef * pw;
        c = -coef * (float)power * (pw / n2);
    }
    if (out_f64)
        ((double4 *)force)[i] = make_double4(c * x, c * y, c * z, e);
    else
        ((float4 *)force)[i] = make_float4(c * x, c * y, c * z, e);
}
} // namespace htf

extern "C" int htf_positions_forces_radial(const void *d_positions, int dtype, unsigned N, int ncomp, int power,
                                           double coef, void *d_force, int force_dtype, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_positions && d_force, "htf_positions_forces_radial: null pointer");
    HTF_REQUIRE(dtype == HTF_F32 || dtype == HTF_F64, "htf_positions_forces_radial: bad dtype %d", dtype);
    HTF_REQUIRE(force_dtype == HTF_F32 || force_dtype == HTF_F64, "htf_positions_forces_radial: bad force dtype %d", force_dtype);
    HTF_REQUIRE(ncomp == 3 || ncomp == 4, "htf_positions_forces_radial: the norm runs over 3 or 4 columns (got %d)", ncomp);
    HTF_REQUIRE(power >= -16 && power <= 16 && power != 0, "htf_positions_forces_radial: power %d outside [-16, 16] \\ {0}", power);
    if (N == 0) return HTF_OK;
    const unsigned grid = (N + 255) / 256;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((positions_radial_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)d_positions, N, ncomp, power, (float)coef, d_force, force_dtype == HTF_F64);
    else
        hipLaunchKernelGGL((positions_radial_kernel<double>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const double4 *)d_positions, N, ncomp, power, (float)coef, d_force, force_dtype == HTF_F64);
    return check_launch("positions_radial_kernel");
}
