// HOOMD BoxDim arithmetic shared by the pair-vector build and the fused evaluator.
// Every function here pins fp contraction OFF so that pair vectors (and therefore the
// keep / drop decision at r_cut) are bit-identical in both kernels and to the oracle
// (oracle/htf_oracle.py:min_image), whatever flags the including TU is built with.
#pragma once
#include "htf_common.h"

namespace htf {

template <typename T>
struct BoxT {
    T L[3], Linv[3], xy, xz, yz;
    int periodic[3];
    int ortho; // all tilt factors are zero: the shear corrections below subtract exact zeros, skip them
};

template <typename T>
static BoxT<T> make_boxt(const htf_box *hb) {
    BoxT<T> b;
    for (int d = 0; d < 3; ++d) {
        b.L[d] = (T)hb->hi[d] - (T)hb->lo[d];
        b.Linv[d] = (T)1 / b.L[d];
        b.periodic[d] = hb->periodic[d];
    }
    b.xy = (T)hb->tilt[0];
    b.xz = (T)hb->tilt[1];
    b.yz = (T)hb->tilt[2];
    b.ortho = (b.xy == (T)0 && b.xz == (T)0 && b.yz == (T)0) ? 1 : 0;
    return b;
}

template <typename T> __device__ __forceinline__ T rint_t(T x);
template <> __device__ __forceinline__ float rint_t<float>(float x) { return rintf(x); }
template <> __device__ __forceinline__ double rint_t<double>(double x) { return rint(x); }

// HOOMD-blue 2.x BoxDim::minImage, device (rint) form.
template <typename T>
__device__ __forceinline__ void min_image(T &x, T &y, T &z, const BoxT<T> &b) {
#pragma clang fp contract(off)
    // (the empty asm keeps the wave-uniform tilt branches BRANCHES: if-converted, their six multiplies and subtractions
    //  ran for every candidate of every orthorhombic box -- the kernels that call this are VALU-issue bound)
    if (b.periodic[2]) {
        T img = rint_t<T>(z * b.Linv[2]);
        z -= b.L[2] * img;
        if (!b.ortho) { // wave-uniform
            asm volatile("" ::: "memory");
            y -= b.L[2] * b.yz * img;
            x -= b.L[2] * b.xz * img;
        }
    }
    if (b.periodic[1]) {
        T img = rint_t<T>(y * b.Linv[1]);
        y -= b.L[1] * img;
        if (!b.ortho) {
            asm volatile("" ::: "memory");
            x -= b.L[1] * b.xy * img;
        }
    }
    if (b.periodic[0]) {
        T img = rint_t<T>(x * b.Linv[0]);
        x -= b.L[0] * img;
    }
}

// dx = minimage(pk - pi) and |dx|^2, exactly as prepareNeighbors computes them
template <typename T, typename V>
__device__ __forceinline__ T pair_vector(const V &pk, const V &pi, const BoxT<T> &b, T &dx, T &dy, T &dz) {
#pragma clang fp contract(off)
    dx = pk.x - pi.x;
    dy = pk.y - pi.y;
    dz = pk.z - pi.z;
    min_image<T>(dx, dy, dz, b);
    return dx * dx + dy * dy + dz * dz;
}

// The same for an orthorhombic box periodic in x, y and z -- the case every kernel's fast path takes: BoxDim::minImage's
// operations in its order (z, y, x), without the tilt and non-periodic cases (12 instructions instead of 22).
template <typename T, typename V>
__device__ __forceinline__ T pair_vector_simple(const V &pk, const V &pi, const BoxT<T> &b, T &dx, T &dy, T &dz) {
#pragma clang fp contract(off)
    dx = pk.x - pi.x;
    dy = pk.y - pi.y;
    dz = pk.z - pi.z;
    dz -= b.L[2] * rint_t<T>(dz * b.Linv[2]);
    dy -= b.L[1] * rint_t<T>(dy * b.Linv[1]);
    dx -= b.L[0] * rint_t<T>(dx * b.Linv[0]);
    return dx * dx + dy * dy + dz * dz;
}

// A neighbor's position for the row-building kernels (pair_vectors.hip, fused_eval.hip): x, y, z and the type bits, which are the LOW dword of w -- for fp64
// positions 28 of the 32 bytes (a 16-B and a 12-B load instead of two 16-B ones: an eighth less data through the texture path
// that the fp64 wire keeps busiest, one register less per gathered position)
__device__ __forceinline__ float4 load_neighbor(const float4 *__restrict__ pos, unsigned k) { return pos[k]; }
__device__ __forceinline__ double4 load_neighbor(const double4 *__restrict__ pos, unsigned k) {
    const double *p = reinterpret_cast<const double *>(pos + k);
    double4 r;
    r.x = p[0];
    r.y = p[1];
    r.z = p[2];
    r.w = __longlong_as_double((long long)(unsigned)reinterpret_cast<const int *>(p)[6]);
    return r;
}

// HOOMD __scalar_as_int: the int type id lives in the (low) 32 bits of pos.w
__device__ __forceinline__ int scalar_as_int(float w) { return __float_as_int(w); }
__device__ __forceinline__ int scalar_as_int(double w) { return (int)(__double_as_longlong(w) & 0xffffffffll); }

} // namespace htf
