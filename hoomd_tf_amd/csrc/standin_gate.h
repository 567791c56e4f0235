// Internal to the stand-in's translation units (standin.hip, brick.hip): the device-side gate of a conditional rebuild and the
// convention for INERT rows of fixed-capacity particle arrays.
#pragma once
#include "htf_common.h"

namespace htf {

// Device-side gate of a conditional neighbor-list rebuild: between htfs_set_gate(d_disp2, thr2) and
// htfs_set_gate(NULL, 0) every binning / search / migration kernel launched through the stand-in returns at entry
// unless *d_disp2 > thr2 -- the decision NeighborList::distanceCheck takes on the host is taken by the
// kernels themselves, so the step loop never waits for a read-back.
struct Gate {
    const float *disp2;
    float thr2;
    __device__ __forceinline__ bool closed() const { return disp2 != nullptr && !(*disp2 > thr2); }
};
extern thread_local Gate g_gate;

// An INERT row (round 5, hoomd_tf_amd/brick.py): a slot of a fixed-capacity particle array that holds no particle.  Its x is
// NaN: every distance to it compares false (never a neighbor, never inside a cutoff), the binning leaves it out of every
// cell, its neighbor row is empty (force 0), its velocity is 0 -- so every kernel may run over the array's CAPACITY and no
// launch is sized by a particle count the host would have to read back.
template <typename T>
__device__ __forceinline__ bool is_inert(T x) { return !(x == x); }
constexpr unsigned kDeadCell = 0xFFFFFFFFu;

template <typename T>
struct SBox {
    T lo[3], L[3], Linv[3];
    int periodic[3];
};

template <typename T>
static SBox<T> make_sbox(const htf_box *b) {
    SBox<T> s;
    for (int d = 0; d < 3; ++d) {
        s.lo[d] = (T)b->lo[d];
        s.L[d] = (T)b->hi[d] - (T)b->lo[d];
        s.Linv[d] = (T)1 / s.L[d];
        s.periodic[d] = b->periodic[d];
    }
    return s;
}

template <typename T>
__device__ __forceinline__ T wrap1(T x, T lo, T L, T Linv, int periodic) {
    if (!periodic) return x;
    T f = floor((x - lo) * Linv);
    return x - f * L;
}

template <typename T>
__device__ __forceinline__ T mimg(T d, T L, T Linv, int periodic) {
    return periodic ? d - L * rint(d * Linv) : d;
}

// the leapfrog update of one particle (IntegratorTwoStep + TwoStepNVE analogue, unit mass): shared by nve_step_kernel and the
// brick decomposition's integrate-and-pack kernel so that both produce the same bits
template <typename T, typename V4>
__device__ __forceinline__ void nve_advance(V4 &p, V4 &v, const V4 &f, T dt, const SBox<T> &b) {
    v.x += dt * f.x;
    v.y += dt * f.y;
    v.z += dt * f.z;
    p.x = wrap1<T>(p.x + dt * v.x, b.lo[0], b.L[0], b.Linv[0], b.periodic[0]);
    p.y = wrap1<T>(p.y + dt * v.y, b.lo[1], b.L[1], b.Linv[1], b.periodic[1]);
    p.z = wrap1<T>(p.z + dt * v.z, b.lo[2], b.L[2], b.Linv[2], b.periodic[2]);
}

} // namespace htf
