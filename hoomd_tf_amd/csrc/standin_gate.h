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

} // namespace htf
