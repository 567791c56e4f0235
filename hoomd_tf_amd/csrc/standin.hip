// HOOMD stand-in kernels (include/htf_standin.h): leapfrog NVE step, displacement
// check, binned neighbor search.  Outside the drop-in boundary; exists so the force
// path can be driven and timed as MD without HOOMD-blue in the image.
#include <cstdlib>
#include <unordered_map>

#include "htf_common.h"
#include "box_math.h"
#include "htf_standin.h"
#include "standin_gate.h"
#include "key_sort.h"

namespace htf {

thread_local Gate g_gate = {nullptr, 0.f};

template <typename T>
__global__ __launch_bounds__(256) void nve_step_kernel(typename Vec4<T>::type *__restrict__ pos,
                                                       typename Vec4<T>::type *__restrict__ vel,
                                                       const typename Vec4<T>::type *__restrict__ force,
                                                       unsigned N, T dt, SBox<T> b) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    auto p = pos[i];
    auto v = vel[i];
    nve_advance<T>(p, v, force[i], dt, b);
    pos[i] = p;
    vel[i] = v;
}

template <typename T>
__global__ __launch_bounds__(1024) void max_disp_kernel(const typename Vec4<T>::type *__restrict__ pos,
                                                        const typename Vec4<T>::type *__restrict__ ref, unsigned N,
                                                        SBox<T> b, float *__restrict__ out) {
    __shared__ float s_max[16];
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    float d2 = 0.f;
    if (i < N) {
        auto p = pos[i];
        auto r = ref[i];
        T dx = mimg<T>(p.x - r.x, b.L[0], b.Linv[0], b.periodic[0]);
        T dy = mimg<T>(p.y - r.y, b.L[1], b.Linv[1], b.periodic[1]);
        T dz = mimg<T>(p.z - r.z, b.L[2], b.Linv[2], b.periodic[2]);
        d2 = (float)(dx * dx + dy * dy + dz * dz);
        if (!(d2 == d2)) d2 = 0.f; // an inert row (standin_gate.h) has not moved
    }
    for (int m = 1; m < 64; m <<= 1) d2 = fmaxf(d2, __shfl_xor(d2, m));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = d2;
    __syncthreads();
    if (threadIdx.x < 16) {
        d2 = s_max[threadIdx.x];
        for (int m = 1; m < 16; m <<= 1) d2 = fmaxf(d2, __shfl_xor(d2, m));
        // d2 >= 0: uint order == float order.  One atomic per 1024 particles (same-address
        // atomics serialise at ~12 ns each: one per wave cost 25 us at N = 131072).
        if (threadIdx.x == 0 && __float_as_uint(d2) > *(volatile unsigned *)out)
            atomicMax((unsigned *)out, __float_as_uint(d2));
    }
}

// The distance check of a replayed cycle in ONE launch (round 5: inside a hipGraph every dependent node costs >= 4.5 us, and
// "zero the word, fill it, copy it to the host" was three): the blocks accumulate into work[0] as max_disp_kernel does, the LAST
// block to finish (ticket in work[1]) publishes out = [largest d^2, cycle number + 1] and leaves both work words zero for the next
// launch.  The same block is the cycle's mailman: it copies the status words other kernels left on the device (htfs_mirror: the
// decomposition's counts and flags, the list's largest row) and its own result straight into PINNED HOST memory -- the cycle
// number last, behind a system-scope fence, so a host that sees cycle c sees everything that belongs to it -- where the captured
// chain had a copy node each (a second stream forked inside the capture was worse: 29.6 -> 42 us per step, every kernel of a
// multi-branch graph slower and 16-22 us at each fork).
template <typename T>
__global__ __launch_bounds__(1024) void check_disp_kernel(const typename Vec4<T>::type *__restrict__ pos,
                                                          const typename Vec4<T>::type *__restrict__ ref, unsigned N,
                                                          SBox<T> b, unsigned *__restrict__ work, float *__restrict__ out,
                                                          float *__restrict__ h_out, htfs_mirror mirror) {
    __shared__ float s_max[16];
    __shared__ unsigned s_last;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    float d2 = 0.f;
    if (i < N) {
        auto p = pos[i];
        auto r = ref[i];
        T dx = mimg<T>(p.x - r.x, b.L[0], b.Linv[0], b.periodic[0]);
        T dy = mimg<T>(p.y - r.y, b.L[1], b.Linv[1], b.periodic[1]);
        T dz = mimg<T>(p.z - r.z, b.L[2], b.Linv[2], b.periodic[2]);
        d2 = (float)(dx * dx + dy * dy + dz * dz);
        if (!(d2 == d2)) d2 = 0.f; // an inert row has not moved
    }
    for (int m = 1; m < 64; m <<= 1) d2 = fmaxf(d2, __shfl_xor(d2, m));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = d2;
    if (threadIdx.x == 0) s_last = 0u;
    __syncthreads();
    if (threadIdx.x < 16) {
        d2 = s_max[threadIdx.x];
        for (int m = 1; m < 16; m <<= 1) d2 = fmaxf(d2, __shfl_xor(d2, m));
        if (threadIdx.x == 0) {
            if (__float_as_uint(d2) > *(volatile unsigned *)work) atomicMax(work, __float_as_uint(d2));
            __threadfence();
            if (atomicAdd(work + 1, 1u) == gridDim.x - 1u) s_last = 1u; // every other block's maximum is in
        }
    }
    __syncthreads();
    if (s_last == 0u) return;
    for (unsigned m = 0; m < mirror.n; ++m) {
        const unsigned *src = (const unsigned *)mirror.src[m];
        unsigned *dst = (unsigned *)mirror.dst[m];
        for (unsigned w = threadIdx.x; w < mirror.words[m]; w += blockDim.x) dst[w] = __builtin_nontemporal_load(src + w);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        // the cycle number is an unsigned WORD in the second slot (bit pattern, not a float value: as a float it stopped counting at
        // 2^24 -- ten minutes of replayed cycles at 30 us per step).  The host compares modulo 2^32.
        const float worst = __uint_as_float(atomicExch(work, 0u));
        const unsigned cycle = __float_as_uint(out[1]) + 1u;
        out[0] = worst;
        out[1] = __uint_as_float(cycle);
        work[1] = 0u;
        if (h_out != nullptr) {
            h_out[0] = worst;
            __threadfence_system();
            *(volatile unsigned *)(h_out + 1) = cycle;
        }
    }
}

template <typename T>
__device__ __forceinline__ int cell_coord(T x, T lo, T Linv, int n) {
    int c = (int)floor((x - lo) * Linv * (T)n);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

template <typename T>
__global__ __launch_bounds__(256) void cell_index_kernel(const typename Vec4<T>::type *__restrict__ pos, unsigned Ntot,
                                                         SBox<T> b, int nx, int ny, int nz,
                                                         unsigned *__restrict__ cell_of, Gate gate) {
    if (gate.closed()) return;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ntot) return;
    auto p = pos[i];
    if (is_inert(p.x)) { // in no cell: never a candidate, never searched for
        cell_of[i] = kDeadCell;
        return;
    }
    int cx = cell_coord<T>(p.x, b.lo[0], b.Linv[0], nx);
    int cy = cell_coord<T>(p.y, b.lo[1], b.Linv[1], ny);
    int cz = cell_coord<T>(p.z, b.lo[2], b.Linv[2], nz);
    cell_of[i] = (unsigned)((cz * ny + cy) * nx + cx);
}

// One G-lane group per particle.  The stencil reaches w = 1 cell (cells of width >= r_list) or
// w = 2 cells (width >= r_list / 2: 125 cells of 1/8 the volume, 1.7x fewer candidate checks --
// the search is VALU-issue bound, PMC: 120 M wave-instructions at C3) per direction.  For every
// (dy, dz) row of the stencil the 2 wx + 1 x-adjacent cells are contiguous in the cell-sorted
// arrays, so they are walked as one range (plus a second, usually empty, range when the stencil
// wraps around the box).
// Candidates are read from the CELL-SORTED position copy: 16 lanes x 16 B contiguous.
// Hits are compacted with a ballot restricted to the group.  Every lane of the wave runs
// the same trip counts (ranges of other groups are padded to the wave maximum) so the
// ballots and shuffles are convergent.
// Per cell and per (dy, dz) row of its stencil: the candidate ranges of the cell-sorted arrays, as
// (begin, length) of the main x-run and of the run that wraps around the box (length 0 if none).
// Every particle of a cell walks the same ranges: the search kernel then needs ONE load per row
// instead of index arithmetic and two dependent cell_start loads.
// a range table entry: (start, length) of the row's run of candidates and of the run that wraps around the box in x; the top
// four bits of a length say through which periodic image the run is seen (y, z codes on .y; the x code on .w)
// the w of a cell-sorted position (htfs_gather4_tagged): particle index | (type >= type_split) << 31
constexpr unsigned kTagSide = 1u << 31;
constexpr unsigned kRangeWrapShift = 28u, kRangeLenMask = (1u << kRangeWrapShift) - 1u;
struct RangesArgs {
    int nx, ny, nz, wx, wy, wz, px, py, pz;
    const unsigned *cell_start;
    uint4 *table;
    unsigned *max_neigh;
};
__device__ __forceinline__ void cell_ranges_body(const unsigned t, const RangesArgs &a) {
    const int nx = a.nx, ny = a.ny, nz = a.nz, wx = a.wx, wy = a.wy, wz = a.wz, px = a.px, py = a.py, pz = a.pz;
    const unsigned *__restrict__ cell_start = a.cell_start;
    uint4 *__restrict__ table = a.table;
    unsigned *__restrict__ max_neigh = a.max_neigh;
    const int nrow = (2 * wy + 1) * (2 * wz + 1);
    if (t == 0) *max_neigh = 0u; // the search kernel behind this one accumulates the largest row into it
    const unsigned ncell = (unsigned)(nx * ny * nz);
    if (t >= ncell * (unsigned)nrow) return;
    const int c = (int)(t / nrow), r = (int)(t % nrow);
    const int cx = c % nx, cy = (c / nx) % ny, cz = c / (nx * ny);
    const int dz = r / (2 * wy + 1) - wz, dy = r % (2 * wy + 1) - wy;
    int a0 = cx - wx, a1 = cx + wx, b0 = 0, b1 = -1;
    if (a0 < 0) {
        if (px) { b0 = nx + a0; b1 = nx - 1; }
        a0 = 0;
    } else if (a1 >= nx) {
        if (px) { b0 = 0; b1 = a1 - nx; }
        a1 = nx - 1;
    }
    int ay = cy + dy, az = cz + dz;
    bool skip = false;
    // which periodic image the row's cells are seen through: 1 = the one a box length below, 2 = above (kRangeWrap*)
    unsigned wrap_y = 0u, wrap_z = 0u;
    if (ay < 0) { skip |= !py; ay += ny; wrap_y = 1u; } else if (ay >= ny) { skip |= !py; ay -= ny; wrap_y = 2u; }
    if (az < 0) { skip |= !pz; az += nz; wrap_z = 1u; } else if (az >= nz) { skip |= !pz; az -= nz; wrap_z = 2u; }
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (!skip) {
        const unsigned rowbase = (unsigned)((az * ny + ay) * nx);
        o.x = cell_start[rowbase + a0];
        o.y = (cell_start[rowbase + a1 + 1] - o.x) | (wrap_y << kRangeWrapShift) | (wrap_z << (kRangeWrapShift + 2));
        if (b1 >= b0) {
            o.z = cell_start[rowbase + b0];
            o.w = (cell_start[rowbase + b1 + 1] - o.z) | ((b0 == 0 ? 2u : 1u) << kRangeWrapShift);
        }
    }
    table[t] = o;
}
__global__ __launch_bounds__(256) void cell_ranges_kernel(RangesArgs a, Gate gate) {
    if (gate.closed()) return;
    cell_ranges_body(blockIdx.x * blockDim.x + threadIdx.x, a);
}

template <typename T, int G, bool SHIFT>
__global__ __launch_bounds__(256) void build_nlist_kernel(const typename Vec4<T>::type *__restrict__ pos,
                                                          const typename Vec4<T>::type *__restrict__ pos_sorted,
                                                          unsigned N, SBox<T> b, T rl2, int nx, int ny, int nz,
                                                          int wx, int wy, int wz,
                                                          const unsigned *__restrict__ cell_start, unsigned pitch,
                                                          int type_split,
                                                          unsigned *__restrict__ n_neigh, unsigned *__restrict__ head_list,
                                                          unsigned *__restrict__ nlist, unsigned *__restrict__ max_neigh,
                                                          const uint4 *__restrict__ ranges, Gate gate) {
    if (gate.closed()) return;
    const unsigned lane = threadIdx.x & 63u, g = lane % G, sub = lane / G;
    const unsigned i = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * (64 / G) + sub;
    const bool in_range = i < N;
    const auto pi = pos[in_range ? i : 0];
    const bool active = in_range && !is_inert(pi.x); // an inert row walks nothing and gets an empty row
    const int cx = cell_coord<T>(pi.x, b.lo[0], b.Linv[0], nx);
    const int cy = cell_coord<T>(pi.y, b.lo[1], b.Linv[1], ny);
    const int cz = cell_coord<T>(pi.z, b.lo[2], b.Linv[2], nz);
    const bool side_i = type_split >= 0 && scalar_as_int(pi.w) >= type_split;
    unsigned count = 0;
    unsigned *row = nlist + (size_t)(active ? i : 0) * pitch;
    // the cell's candidate ranges come from the table (cell_ranges_kernel); the next row's entry is
    // requested before the current row is walked
    const int nrow = (2 * wy + 1) * (2 * wz + 1);
    const uint4 *mine = ranges + (size_t)((cz * ny + cy) * nx + cx) * nrow;
    uint4 rg_next = active ? mine[0] : make_uint4(0u, 0u, 0u, 0u);
    for (int r = 0; r < nrow; ++r) {
        const uint4 rg = rg_next;
        if (r + 1 < nrow) rg_next = active ? mine[r + 1] : make_uint4(0u, 0u, 0u, 0u);
        // SHIFT (every periodic axis has >= 7 cells): the cell a candidate sits in says which periodic image of it is the
        // near one, so particle i is moved by that box vector once per row instead of rint() per candidate and axis
        T piy = pi.y, piz = pi.z, six = (T)0;
        if (SHIFT) {
            const unsigned wy_ = (rg.y >> kRangeWrapShift) & 3u, wz_ = rg.y >> (kRangeWrapShift + 2), wx_ = rg.w >> kRangeWrapShift;
            piy += wy_ == 1u ? b.L[1] : (wy_ == 2u ? -b.L[1] : (T)0);
            piz += wz_ == 1u ? b.L[2] : (wz_ == 2u ? -b.L[2] : (T)0);
            six = wx_ == 1u ? b.L[0] : -b.L[0];
        }
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const unsigned beg = part ? rg.z : rg.x, len = (part ? rg.w : rg.y) & kRangeLenMask;
            const T pix = (SHIFT && part) ? pi.x + six : pi.x;
            // a per-lane trip count: the lanes of a group share it and leave the loop together, so the ballot below still
            // sees whole groups (a wave-wide maximum through three dependent cross-lane reads per row cost more)
            // (four sub-trips per iteration with their loads hoisted: no gain -- the walk is VALU-issue bound, ~35
            //  instructions per candidate and 3.7 candidates per hit, not latency-bound)
            const unsigned n = active ? len : 0u;
            for (unsigned t = 0; t < n; t += G) {
                // No branch inside: the load index is clamped into the run, and a candidate's verdict is the AND of
                // single-compare wave masks (scalar), read back as the lane's predicate only for the store -- the walk is
                // VALU-issue bound.  One 16-B (32-B) load per candidate: its index rides in w (htfs_gather4_tagged).
                const unsigned m_idx = t + g;
                const auto pk = pos_sorted[beg + min(m_idx, n - 1u)];
                const unsigned tag = (unsigned)scalar_as_int(pk.w);
                const unsigned k = tag & ~kTagSide;
                T ddx = pk.x - pix, ddy = pk.y - piy, ddz = pk.z - piz;
                if (!SHIFT) {
                    ddx = mimg<T>(ddx, b.L[0], b.Linv[0], b.periodic[0]);
                    ddy = mimg<T>(ddy, b.L[1], b.Linv[1], b.periodic[1]);
                    ddz = mimg<T>(ddz, b.L[2], b.Linv[2], b.periodic[2]);
                }
                unsigned long long hits = ballot64(m_idx < n) & ballot64(k != i) & ballot64(ddx * ddx + ddy * ddy + ddz * ddz <= rl2);
                if (type_split >= 0) { // wave-uniform
                    asm volatile("" ::: "memory");
                    hits &= ballot64(((tag & kTagSide) != 0u) == side_i);
                }
                // the group's G bits of the wave mask: rank and count from two population counts of a 32-bit word
                const unsigned seg = (unsigned)(hits >> (G * sub)) & (G >= 32 ? ~0u : ((1u << G) - 1u));
                const unsigned rank = count + (unsigned)__popc(seg & ((1u << g) - 1u));
                if (inverse_ballot64(hits) && rank < pitch) row[rank] = k;
                count += (unsigned)__popc(seg);
            }
        }
    }
    if (in_range && g == 0) {
        n_neigh[i] = count < pitch ? count : pitch;
        head_list[i] = i * pitch;
        if (count > *(volatile unsigned *)max_neigh) atomicMax(max_neigh, count);
    }
}


// One WAVE per cell (round 4).  The kernel above walks, for each of a group's particles, every stencil row as its own run of
// G-lane trips: a row of the fine grid holds ~21 candidates, so a third of the lanes of its last trip idle, and the per-row
// bookkeeping (range entry, image shift, trip setup: ~55 instructions) costs as much as two of its three trips -- ~530
// instructions per particle at C3.  All particles of a cell see the SAME candidates.  Here a wave takes a cell: its <= 50
// candidate runs (stencil row x {main, wrapped in x}) sit one per lane, an exclusive scan of their lengths numbers the
// cell's candidates 0..total-1 in walk order, and each trip of 64 finds its candidates' runs by a six-step binary search
// over the lanes (ds_bpermute) -- full lanes whatever the row lengths, no per-row work.  The trip's candidates are loaded
// once, moved to the periodic image their run is seen through, and tested against every particle of the cell (up to eight
// at a time, their coordinates broadcast into registers once per cell): ~16 instructions per 64 tests against ~38.
// A particle's neighbors come out in the order the walk above gives them (runs in table order, candidates in cell-sorted
// order), so the list is the same up to the rounding of a shifted image (the shift is applied to the candidate here, to
// the particle there).
__device__ __forceinline__ void set_tag(float &w, unsigned tag) { w = __uint_as_float(tag); }
__device__ __forceinline__ void set_tag(double &w, unsigned tag) { w = __longlong_as_double((long long)tag); }
__device__ __forceinline__ float bcast_lane_t(float v, unsigned lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)lane));
}
__device__ __forceinline__ double bcast_lane_t(double v, unsigned lane) {
    const long long q = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)q, (int)lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(q >> 32), (int)lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ unsigned bperm_u(unsigned lane_idx, unsigned v) {
    return (unsigned)__builtin_amdgcn_ds_bpermute((int)(lane_idx << 2), (int)v);
}
__device__ __forceinline__ float bperm_t(unsigned lane_idx, float v) { return __uint_as_float(bperm_u(lane_idx, __float_as_uint(v))); }
__device__ __forceinline__ double bperm_t(unsigned lane_idx, double v) {
    const unsigned long long q = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = bperm_u(lane_idx, (unsigned)q), hi = bperm_u(lane_idx, (unsigned)(q >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
constexpr int kCellBatch = 8; // particles of a cell tested per pass over its candidates (a fine-grid cell holds ~4)

template <typename T, bool SHIFT>
__global__ __launch_bounds__(256) void build_nlist_cells_kernel(const typename Vec4<T>::type *__restrict__ pos_sorted, unsigned N,
                                                                SBox<T> b, T rl2, unsigned ncell, int nrow,
                                                                const unsigned *__restrict__ cell_start, unsigned pitch,
                                                                int type_split, unsigned *__restrict__ n_neigh,
                                                                unsigned *__restrict__ head_list, unsigned *__restrict__ nlist,
                                                                unsigned *__restrict__ max_neigh,
                                                                const uint4 *__restrict__ ranges, unsigned split, Gate gate) {
    if (gate.closed()) return;
    using V4 = typename Vec4<T>::type;
    const unsigned lane = threadIdx.x & 63u;
    // `split` waves per cell, wave b of a cell taking its batches b, b + split, ... of kCellBatch particles (round 5: on a grid that
    // does not fill the chip the kernel lasts as long as its longest wave -- one of the few cells with more than one batch -- so a
    // second wave per cell, which leaves at once where there is no second batch, halves that; same rows, same order, bit for bit)
    const unsigned wv = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const unsigned c = wv / split, b0 = wv % split;
    if (c >= ncell) return;
    const unsigned p_begin = cell_start[c], p_end = cell_start[c + 1];
    if (p_begin + b0 * (unsigned)kCellBatch >= p_end) return;
    // a cell of ghosts only (the decomposed step's halo cells: candidates for others, no row of their own) has nothing to search
    if (p_end - p_begin <= 64u) {
        const bool local = lane < p_end - p_begin && ((unsigned)scalar_as_int(pos_sorted[p_begin + lane].w) & ~kTagSide) < N;
        if (ballot64(local) == 0ull) return;
    }

    // ---- the cell's candidate runs, one per lane: lane d = 2 row + (0: main run, 1: the run that wraps around the box in x)
    unsigned beg = 0u, len = 0u;
    T shx = (T)0, shy = (T)0, shz = (T)0; // what comes off a candidate of this run (the image the cell sees it through)
    if (lane < 2u * (unsigned)nrow) {
        const uint4 rg = ranges[(size_t)c * nrow + (lane >> 1)];
        const bool wrapped = (lane & 1u) != 0u;
        beg = wrapped ? rg.z : rg.x;
        len = (wrapped ? rg.w : rg.y) & kRangeLenMask;
        if (SHIFT) {
            const unsigned wy_ = (rg.y >> kRangeWrapShift) & 3u, wz_ = rg.y >> (kRangeWrapShift + 2), wx_ = rg.w >> kRangeWrapShift;
            shy = wy_ == 1u ? b.L[1] : (wy_ == 2u ? -b.L[1] : (T)0);
            shz = wz_ == 1u ? b.L[2] : (wz_ == 2u ? -b.L[2] : (T)0);
            shx = !wrapped ? (T)0 : (wx_ == 1u ? b.L[0] : -b.L[0]);
        }
    }
    // exclusive scan of the lengths over the lanes (Hillis-Steele through ds_bpermute: six steps, once per cell)
    unsigned incl = len;
#pragma unroll
    for (unsigned d = 1; d < 64u; d <<= 1) {
        const unsigned up = bperm_u(lane >= d ? lane - d : lane, incl);
        incl += lane >= d ? up : 0u;
    }
    const unsigned start = incl - len; // non-decreasing over the lanes; == total behind the last run
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    const bool any_shift = SHIFT && ballot64(len != 0u && (shx != (T)0 || shy != (T)0 || shz != (T)0)) != 0ull;

    for (unsigned pb = p_begin + b0 * (unsigned)kCellBatch; pb < p_end; pb += split * (unsigned)kCellBatch) {
        const unsigned np = min(p_end - pb, (unsigned)kCellBatch);
        // the batch's particles: lane p loads particle p, then everything about it becomes wave-uniform
        V4 me;
        me.x = me.y = me.z = (T)0;
        set_tag(me.w, ~0u);
        if (lane < np) me = pos_sorted[pb + lane];
        const unsigned my_tag = (unsigned)scalar_as_int(me.w);
        if (ballot64(lane < np && (my_tag & ~kTagSide) < N) == 0ull) continue; // a batch of ghosts
        unsigned tag_p[kCellBatch], count[kCellBatch];
        T px[kCellBatch], py[kCellBatch], pz[kCellBatch];
        unsigned row_p[kCellBatch]; // first slot of the particle's row (head_list[i] = i * pitch: 32 bits, as the list's own index)
#pragma unroll
        for (int p = 0; p < kCellBatch; ++p) {
            tag_p[p] = ~0u; // (past the batch: index >= N, skipped like a ghost)
            px[p] = py[p] = pz[p] = (T)0;
            count[p] = 0u;
            row_p[p] = 0u;
            if ((unsigned)p < np) { // wave-uniform
                tag_p[p] = (unsigned)__builtin_amdgcn_readlane((int)my_tag, p);
                px[p] = in_vgpr(bcast_lane_t(me.x, p));
                py[p] = in_vgpr(bcast_lane_t(me.y, p));
                pz[p] = in_vgpr(bcast_lane_t(me.z, p));
                row_p[p] = (tag_p[p] & ~kTagSide) * pitch;
            }
        }
        for (unsigned q0 = 0; q0 < total; q0 += 64u) {
            // ---- which run does candidate q belong to: the last lane whose start is <= q
            const unsigned q = q0 + lane;
            unsigned at = 0u;
#pragma unroll
            for (unsigned step = 32u; step != 0u; step >>= 1) {
                const unsigned t = at + step;
                at = bperm_u(t, start) <= q ? t : at;
            }
            const bool valid = q < total;
            const unsigned src = bperm_u(at, beg) + (q - bperm_u(at, start));
            V4 pk;
            pk.x = pk.y = pk.z = (T)0;
            set_tag(pk.w, ~0u);
            if (valid) pk = pos_sorted[src];
            if (any_shift) { // wave-uniform: cells within two of a periodic face
                pk.x -= bperm_t(at, shx);
                pk.y -= bperm_t(at, shy);
                pk.z -= bperm_t(at, shz);
            }
            const unsigned tag = (unsigned)scalar_as_int(pk.w);
            const unsigned k = tag & ~kTagSide;
            const unsigned long long vmask = ballot64(valid);
#pragma unroll
            for (int p = 0; p < kCellBatch; ++p) {
                const unsigned i = tag_p[p] & ~kTagSide;
                if (i >= N) continue; // wave-uniform: past the batch, or a ghost (candidates for others, no row of its own)
                T ddx = pk.x - px[p], ddy = pk.y - py[p], ddz = pk.z - pz[p];
                if (!SHIFT) {
                    ddx = mimg<T>(ddx, b.L[0], b.Linv[0], b.periodic[0]);
                    ddy = mimg<T>(ddy, b.L[1], b.Linv[1], b.periodic[1]);
                    ddz = mimg<T>(ddz, b.L[2], b.Linv[2], b.periodic[2]);
                }
                // (no branch: the verdict is the AND of single-compare wave masks, as in the walk above)
                unsigned long long hits = vmask & ballot64(k != i) & ballot64(ddx * ddx + ddy * ddy + ddz * ddz <= rl2);
                if (type_split >= 0) { // wave-uniform
                    asm volatile("" ::: "memory");
                    hits &= ballot64(((tag ^ tag_p[p]) & kTagSide) == 0u);
                }
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(hits >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hits, count[p]));
                if (inverse_ballot64(hits) && rank < pitch) nlist[row_p[p] + rank] = k;
                count[p] += (unsigned)__builtin_popcountll(hits);
            }
        }
        unsigned cnt = 0u;
#pragma unroll
        for (int p = 0; p < kCellBatch; ++p) cnt = lane == (unsigned)p ? count[p] : cnt;
        if (lane < np) {
            const unsigned i = my_tag & ~kTagSide;
            if (i < N) {
                n_neigh[i] = cnt < pitch ? cnt : pitch;
                head_list[i] = i * pitch;
                if (cnt > *(volatile unsigned *)max_neigh) atomicMax(max_neigh, cnt);
            }
        }
    }
}

} // namespace htf

using namespace htf;

extern "C" int htfs_nve_step(void *d_pos, void *d_vel, const void *d_force, int dtype, unsigned N, double dt,
                             const htf_box *box, htf_stream stream) {
    HTF_REQUIRE(d_pos && d_vel && d_force && box, "htfs_nve_step: null pointer");
    if (N == 0) return HTF_OK;
    unsigned grid = (N + 255) / 256;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((nve_step_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (float4 *)d_pos, (float4 *)d_vel, (const float4 *)d_force, N, (float)dt, make_sbox<float>(box));
    else
        hipLaunchKernelGGL((nve_step_kernel<double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (double4 *)d_pos, (double4 *)d_vel, (const double4 *)d_force, N, dt, make_sbox<double>(box));
    return check_launch("nve_step_kernel");
}

extern "C" int htfs_max_displacement2(const void *d_pos, const void *d_ref, int dtype, unsigned N, const htf_box *box,
                                      float *d_out, htf_stream stream) {
    HTF_REQUIRE(d_pos && d_ref && box && d_out, "htfs_max_displacement2: null pointer");
    if (N == 0) return HTF_OK;
    unsigned grid = (N + 1023) / 1024;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((max_disp_kernel<float>), dim3(grid), dim3(1024), 0, (hipStream_t)stream, (const float4 *)d_pos, (const float4 *)d_ref, N, make_sbox<float>(box), d_out);
    else
        hipLaunchKernelGGL((max_disp_kernel<double>), dim3(grid), dim3(1024), 0, (hipStream_t)stream, (const double4 *)d_pos, (const double4 *)d_ref, N, make_sbox<double>(box), d_out);
    return check_launch("max_disp_kernel");
}

extern "C" int htfs_check_displacement2(const void *d_pos, const void *d_ref, int dtype, unsigned N, const htf_box *box,
                                        unsigned *d_work, float *d_out, float *h_out, const htfs_mirror *mirror, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_pos && d_ref && box && d_work && d_out, "htfs_check_displacement2: null pointer");
    HTF_REQUIRE(N > 0, "htfs_check_displacement2: no rows");
    htfs_mirror mr = {};
    if (mirror != nullptr) {
        mr = *mirror;
        HTF_REQUIRE(mr.n <= HTFS_MIRROR_MAX, "htfs_check_displacement2: %u mirrors (at most %d)", mr.n, HTFS_MIRROR_MAX);
        for (unsigned m = 0; m < mr.n; ++m)
            HTF_REQUIRE(mr.src[m] && mr.dst[m] && ((uintptr_t)mr.src[m] & 3) == 0 && ((uintptr_t)mr.dst[m] & 3) == 0,
                        "htfs_check_displacement2: mirror %u: null or unaligned pointer", m);
    }
    unsigned grid = (N + 1023) / 1024;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((check_disp_kernel<float>), dim3(grid), dim3(1024), 0, (hipStream_t)stream, (const float4 *)d_pos, (const float4 *)d_ref, N, make_sbox<float>(box), d_work, d_out, h_out, mr);
    else
        hipLaunchKernelGGL((check_disp_kernel<double>), dim3(grid), dim3(1024), 0, (hipStream_t)stream, (const double4 *)d_pos, (const double4 *)d_ref, N, make_sbox<double>(box), d_work, d_out, h_out, mr);
    return check_launch("check_disp_kernel");
}

namespace htf {
// The tail of the binning in ONE launch (round 4): the candidate-range table (blocks [0, nb_ranges)), the cell-sorted tagged
// position copy and the new reference positions + rebuild counter (the blocks behind them) -- three kernels of a rebuild that
// depend on the sort but not on each other.  Each of the small kernels of a rebuild costs its launch, the read of the gate and
// two or three dependent round trips to memory (7-11 us apiece at C3), whatever little it computes.
// Binning on a grid that is NOT periodic along an axis the caller's coordinates are (a decomposed system's brick + ghost layer,
// htfs_rebuild_nlist_ghosts' image_L): along such an axis a coordinate is first moved to its image nearest the grid's centre -- a
// row that left the brick through a face on the logical box's boundary has been wrapped to the far side by the integrator, a ghost
// of such a row arrives a box length away -- for the cell index AND for the sorted copy the search measures plain differences on.
struct Reimage {
    double L[3], c[3]; // L = 0: the axis as it is
};
template <typename T>
__device__ __forceinline__ T reimage1(T x, double L, double c) {
    return L > 0.0 ? (T)((double)x - L * rint(((double)x - c) / L)) : x;
}

template <typename V>
__global__ __launch_bounds__(256) void bins_finish_kernel(RangesArgs ra, unsigned nb_ranges, V *__restrict__ dest, const V *__restrict__ src,
                                                          const int *__restrict__ order, unsigned n, int type_split, V *__restrict__ ref,
                                                          unsigned n_ref, unsigned *__restrict__ counter, Reimage im, Gate gate) {
    if (gate.closed()) return;
    if (blockIdx.x < nb_ranges) {
        cell_ranges_body(blockIdx.x * blockDim.x + threadIdx.x, ra);
        return;
    }
    const unsigned i = (blockIdx.x - nb_ranges) * blockDim.x + threadIdx.x;
    if (i == 0 && counter != nullptr) *counter += 1u;
    if (i >= n || i >= ra.cell_start[ra.nx * ra.ny * ra.nz]) return; // (arrays with inert rows: only the binned particles have an entry)
    const unsigned k = (unsigned)order[i];
    V p = src[k];
    if (ref != nullptr && k < n_ref) ref[k] = p; // (order is a permutation: every reference position is written once)
    p.x = reimage1(p.x, im.L[0], im.c[0]);
    p.y = reimage1(p.y, im.L[1], im.c[1]);
    p.z = reimage1(p.z, im.L[2], im.c[2]);
    set_tag(p.w, k | ((type_split >= 0 && scalar_as_int(p.w) >= type_split) ? kTagSide : 0u));
    dest[i] = p;
}

// cell index of every particle and the cells' populations in one pass (htfs_cell_index + the first kernel of htfs_cell_sort)
template <typename T>
__global__ __launch_bounds__(256) void cell_index_count_kernel(const typename Vec4<T>::type *__restrict__ pos, unsigned Ntot, SBox<T> b,
                                                               int nx, int ny, int nz, unsigned *__restrict__ cell_of,
                                                               unsigned *__restrict__ count, Reimage im, Gate gate) {
    if (gate.closed()) return;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ntot) return;
    auto p = pos[i];
    if (is_inert(p.x)) {
        cell_of[i] = kDeadCell;
        return;
    }
    int cx = cell_coord<T>(reimage1(p.x, im.L[0], im.c[0]), b.lo[0], b.Linv[0], nx);
    int cy = cell_coord<T>(reimage1(p.y, im.L[1], im.c[1]), b.lo[1], b.Linv[1], ny);
    int cz = cell_coord<T>(reimage1(p.z, im.L[2], im.c[2]), b.lo[2], b.Linv[2], nz);
    const unsigned c = (unsigned)((cz * ny + cy) * nx + cx);
    cell_of[i] = c;
    atomicAdd(&count[c], 1u);
}

template <typename V, bool TAG>
__global__ void gather4_kernel(V *__restrict__ dest, const V *__restrict__ src, const int *__restrict__ order, unsigned n,
                               int type_split, Gate gate, const unsigned *__restrict__ n_live = nullptr) {
    if (gate.closed()) return;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (n_live != nullptr && i >= *n_live) return; // arrays with inert rows: only the binned particles have an entry
    const unsigned k = (unsigned)order[i];
    V p = src[k];
    if (TAG) set_tag(p.w, k | ((type_split >= 0 && scalar_as_int(p.w) >= type_split) ? kTagSide : 0u));
    dest[i] = p;
}
} // namespace htf

namespace htf {
// ---- cell binning: a counting sort with a deterministic order inside each cell (ascending particle
// index), in four small kernels -- what the stand-in used torch.sort + searchsorted for (38 + 10 us of
// kernels and a dozen launches per rebuild)
__global__ __launch_bounds__(256) void cell_count_kernel(const unsigned *__restrict__ cell_of, unsigned n,
                                                         unsigned *__restrict__ count, Gate gate) {
    if (gate.closed()) return;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && cell_of[i] != kDeadCell) atomicAdd(&count[cell_of[i]], 1u);
}

// exclusive scan of count[0..ncell) into start[0..ncell], one workgroup; cursor <- start.
// Exclusive scan of the cell counts by ONE block, in chunks of 32 768 cells: every thread owns 32 consecutive cells of the
// chunk IN REGISTERS (eight 16-B loads of its own 128-B line: whole lines consumed), serial sum, a wave scan by shuffles, the
// wave totals through a few words of LDS, serial prefix, eight 16-B stores to each output.  Round 2 staged the chunk through 132 KiB of
// LDS (17 us at 29 791 cells; the first version walked global memory with strided dependent accesses: 61 us) -- and then could
// not be scheduled beside the pair-MLP training kernel, whose persistent blocks hold 157 KiB of every CU's LDS for a whole
// 10 ms sweep: 265 us per call in the C5b trace (profiles/r02_bench_mlp_train_kernel_stats.csv).  With 64 B it co-resides
// with anything.
// And 256 threads, not 1024: a 16-wave workgroup needs four wave slots AND 4 x 88 VGPRs on every SIMD of one CU at once, which
// no CU has while a training wave (383 VGPRs) sits on each of its SIMDs -- the round-3 trace still showed 289 us per call (max
// 8.4 ms: the end of the sweep) with the LDS gone.  Four waves, one per SIMD, fit beside it.
// Round 4: one workgroup PER 2 048 cells instead of one for all of them (32 k cells at C3 took 45 us of dependent chunks, 5 us of
// every MD step): workgroup b first sums the counts of the cells before its own (at most ncell reads per thread-block, from L2),
// then scans its chunk.  The counts are no longer zeroed here -- another workgroup may still be summing them -- but by
// cell_order_kernel, the last kernel of the sort.
constexpr unsigned kScanThreads = 256, kScanPer = 8, kScanChunk = kScanThreads * kScanPer;
__global__ __launch_bounds__(kScanThreads) void cell_scan_kernel(const unsigned *__restrict__ count, unsigned ncell,
                                                         unsigned *__restrict__ start, unsigned *__restrict__ cursor, Gate gate) {
    if (gate.closed()) return;
    __shared__ unsigned s_wave[kScanThreads / 64], s_carry[kScanThreads / 64];
    const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const unsigned base = blockIdx.x * kScanChunk;
    // the cells before this workgroup's chunk (base is a multiple of 2 048: 16-byte loads)
    unsigned before_blk = 0;
    {
        const uint4 *p = reinterpret_cast<const uint4 *>(count);
        for (unsigned i = t; i < base / 4; i += kScanThreads) {
            const uint4 v = p[i];
            before_blk += (v.x + v.y) + (v.z + v.w);
        }
        before_blk = group_sum_u<64>(before_blk);
        if (lane == 0) s_carry[wave] = before_blk;
    }
    const unsigned first = base + t * kScanPer;
    unsigned c[kScanPer];
#pragma unroll
    for (unsigned i = 0; i < kScanPer; ++i) c[i] = first + i < ncell ? count[first + i] : 0u;
    unsigned sum = 0;
#pragma unroll
    for (unsigned i = 0; i < kScanPer; ++i) sum += c[i];
    unsigned incl = sum; // inclusive scan over the wave's 64 lanes
#pragma unroll
    for (unsigned off = 1; off < 64; off <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned before = 0, total = 0, carry = 0;
#pragma unroll
    for (unsigned w = 0; w < kScanThreads / 64; ++w) {
        const unsigned v = s_wave[w];
        before += w < wave ? v : 0u;
        total += v;
        carry += s_carry[w];
    }
    unsigned run = carry + before + (incl - sum);
#pragma unroll
    for (unsigned i = 0; i < kScanPer; ++i) // exclusive prefix
        if (first + i < ncell) {
            start[first + i] = run;
            cursor[first + i] = run;
            run += c[i];
        }
    if (blockIdx.x == gridDim.x - 1 && t == 0) start[ncell] = carry + total;
}

__global__ __launch_bounds__(256) void cell_scatter_kernel(const unsigned *__restrict__ cell_of, unsigned n,
                                                           unsigned *__restrict__ cursor, unsigned *__restrict__ order, Gate gate) {
    if (gate.closed()) return;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && cell_of[i] != kDeadCell) order[atomicAdd(&cursor[cell_of[i]], 1u)] = i;
}

// the scatter's order inside a cell depends on the atomics' timing: sort each cell's few members
__global__ __launch_bounds__(256) void cell_order_kernel(const unsigned *__restrict__ start, unsigned ncell,
                                                         unsigned *__restrict__ order, unsigned *__restrict__ count, Gate gate) {
    if (gate.closed()) return;
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    count[c] = 0u; // left zeroed for the next call (htfs_cell_sort)
    const unsigned b = start[c], e = start[c + 1];
    for (unsigned i = b + 1; i < e; ++i) { // insertion sort (cells hold a handful of particles)
        const unsigned v = order[i];
        unsigned j = i;
        while (j > b && order[j - 1] > v) {
            order[j] = order[j - 1];
            --j;
        }
        order[j] = v;
    }
}
} // namespace htf

// The first half of a binning scratch (the per-cell counts) is zero on entry: every call that RUNS leaves it so
// (cell_order_kernel) and a call the gate holds back does not touch it.  A gated call therefore needs no memset -- as long as
// the last ungated call on this scratch used the same ncell (otherwise the words between the two sizes are stale).
// Remembered per scratch pointer and calling thread; anything else is zeroed again (an unconditional memset is always right).
static int zero_counts_if_needed(unsigned *count, unsigned ncell, hipStream_t s) {
    static thread_local std::unordered_map<const void *, unsigned> zeroed_for;
    HTF_REQUIRE(((uintptr_t)count & 15) == 0, "binning scratch must be 16-byte aligned (cell_scan_kernel reads it with 16-byte loads)");
    auto it = zeroed_for.find(count);
    if (!g_gate.disp2 || it == zeroed_for.end() || it->second != ncell) {
        HTF_CHECK_HIP(hipMemsetAsync(count, 0, (size_t)ncell * sizeof(unsigned), s));
        zeroed_for[count] = ncell;
    }
    return HTF_OK;
}

extern "C" int htfs_cell_sort(const unsigned *d_cell_of, unsigned Ntot, unsigned ncell, unsigned *d_scratch,
                              unsigned *d_cell_start, unsigned *d_order, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_cell_of && d_scratch && d_cell_start && d_order, "htfs_cell_sort: null pointer");
    HTF_REQUIRE(ncell > 0, "htfs_cell_sort: no cells");
    hipStream_t s = (hipStream_t)stream;
    unsigned *count = d_scratch, *cursor = d_scratch + ncell;
    if (int rc = zero_counts_if_needed(count, ncell, s)) return rc;
    if (Ntot) hipLaunchKernelGGL(cell_count_kernel, dim3((Ntot + 255) / 256), dim3(256), 0, s, d_cell_of, Ntot, count, g_gate);
    hipLaunchKernelGGL(cell_scan_kernel, dim3((ncell + kScanChunk - 1) / kScanChunk), dim3(kScanThreads), 0, s, count, ncell, d_cell_start, cursor, g_gate);
    if (Ntot) hipLaunchKernelGGL(cell_scatter_kernel, dim3((Ntot + 255) / 256), dim3(256), 0, s, d_cell_of, Ntot, cursor, d_order, g_gate);
    hipLaunchKernelGGL(cell_order_kernel, dim3((ncell + 255) / 256), dim3(256), 0, s, d_cell_start, ncell, d_order, count, g_gate);
    return check_launch("htfs_cell_sort");
}

static int gather4(void *d_dest, const void *d_src, const int *d_order, int dtype, unsigned n, bool tag, int type_split,
                   htf_stream stream, const unsigned *d_n_live = nullptr) {
    using namespace htf;
    HTF_REQUIRE(d_dest && d_src && d_order, "htfs_gather4: null pointer");
    HTF_REQUIRE(!tag || n <= kTagSide, "htfs_gather4_tagged: %u particles do not fit the 31-bit tag", n);
    if (n == 0) return HTF_OK;
    const dim3 grid((n + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == HTF_F32) {
        if (tag) hipLaunchKernelGGL((gather4_kernel<float4, true>), grid, block, 0, s, (float4 *)d_dest, (const float4 *)d_src, d_order, n, type_split, g_gate, d_n_live);
        else hipLaunchKernelGGL((gather4_kernel<float4, false>), grid, block, 0, s, (float4 *)d_dest, (const float4 *)d_src, d_order, n, type_split, g_gate, d_n_live);
    } else {
        if (tag) hipLaunchKernelGGL((gather4_kernel<double4, true>), grid, block, 0, s, (double4 *)d_dest, (const double4 *)d_src, d_order, n, type_split, g_gate, d_n_live);
        else hipLaunchKernelGGL((gather4_kernel<double4, false>), grid, block, 0, s, (double4 *)d_dest, (const double4 *)d_src, d_order, n, type_split, g_gate, d_n_live);
    }
    return check_launch("gather4_kernel");
}

extern "C" int htfs_gather4(void *d_dest, const void *d_src, const int *d_order, int dtype, unsigned n, htf_stream stream) {
    return gather4(d_dest, d_src, d_order, dtype, n, false, -1, stream);
}

extern "C" int htfs_gather4_tagged(void *d_dest, const void *d_src, const int *d_order, int dtype, unsigned n, int type_split,
                                   htf_stream stream) {
    return gather4(d_dest, d_src, d_order, dtype, n, true, type_split, stream);
}

extern "C" int htfs_gather4_tagged_live(void *d_dest, const void *d_src, const int *d_order, int dtype, unsigned n_max,
                                        const unsigned *d_n_live, int type_split, htf_stream stream) {
    HTF_REQUIRE(d_n_live, "htfs_gather4_tagged_live: null pointer");
    return gather4(d_dest, d_src, d_order, dtype, n_max, true, type_split, stream, d_n_live);
}

// ---------------------------------------------------------------------------- slab decomposition: migration plan
// (the Communicator's part of a rebuild, hoomd_tf_amd/domain.py: SlabDomain.rebuild)
// key = destination * 4 + ghost class, for every local particle:
//   destination 0 stay | 1 the left neighbor | 2 the right neighbor | 3 further than an adjacent slab (an error upstream);
//               with two slabs both faces lead to the one peer and everything that leaves travels as "right";
//   class, IN THE SLAB THE PARTICLE ENDS UP IN: 0 interior | 1 within r_ghost of its left face only | 2 of both faces |
//               3 of its right face only.
// The arithmetic is the torch restatement's (domain.py), in the positions' own precision: owner = #(interior cuts <= x),
// near_l = x < lo(owner) + r_ghost, near_r = x >= hi(owner) - r_ghost.
template <typename T, typename V4>
__global__ __launch_bounds__(256) void slab_classify_kernel(const V4 *__restrict__ pos, unsigned N, const T *__restrict__ bounds,
                                                            int world, int rank, T r_ghost, unsigned *__restrict__ key) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const T x = pos[i].x;
    int owner = 0;
    for (int c = 1; c < world; ++c) owner += (bounds[c] <= x) ? 1 : 0;
    const int left = (rank + world - 1) % world, right = (rank + 1) % world;
    unsigned dest;
    if (world == 2)
        dest = owner != rank ? 2u : 0u;
    else
        dest = owner == rank ? 0u : (owner == left ? 1u : (owner == right ? 2u : 3u));
    const bool near_l = x < bounds[owner] + r_ghost, near_r = x >= bounds[owner + 1] - r_ghost;
    const unsigned cls = near_l ? (near_r ? 2u : 1u) : (near_r ? 3u : 0u);
    key[i] = dest * 4u + cls;
}

extern "C" int htfs_slab_classify(const void *d_pos, int dtype, unsigned N, const void *d_bounds, int world, int rank,
                                  double r_ghost, unsigned *d_key, htf_stream stream) {
    HTF_REQUIRE(d_pos && d_bounds && d_key, "htfs_slab_classify: null pointer");
    HTF_REQUIRE(world >= 2 && rank >= 0 && rank < world, "htfs_slab_classify: rank %d of %d", rank, world);
    if (N == 0) return HTF_OK;
    const unsigned grid = (N + 255) / 256;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((slab_classify_kernel<float, float4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4 *)d_pos, N,
                           (const float *)d_bounds, world, rank, (float)r_ghost, d_key);
    else
        hipLaunchKernelGGL((slab_classify_kernel<double, double4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const double4 *)d_pos,
                           N, (const double *)d_bounds, world, rank, r_ghost, d_key);
    return check_launch("slab_classify_kernel");
}

extern "C" int htfs_key_sort16(const unsigned *d_key, unsigned N, unsigned *d_scratch, unsigned *d_start, unsigned *d_order,
                               htf_stream stream) {
    HTF_REQUIRE(d_key && d_scratch && d_start && d_order, "htfs_key_sort16: null pointer");
    return htf::key_sort<16>(d_key, N, nullptr, d_scratch, d_start, d_order, (hipStream_t)stream); // csrc/key_sort.h
}

// up to HTFS_MAX_SEGMENTS row ranges copied in one launch: dst[dst_start[s] + j] = src[src_start[s] + j], j < count[s],
// rows of `row_words` 32-bit words.  What merges three class-sorted segments (stayed | from the right | from the left)
// into one class-sorted array without a sort: the 12 (segment, class) runs and their destinations are known on the host.
struct SegTable {
    unsigned n;
    unsigned src[HTFS_MAX_SEGMENTS], dst[HTFS_MAX_SEGMENTS], cnt[HTFS_MAX_SEGMENTS];
};

__global__ __launch_bounds__(256) void segment_copy_kernel(unsigned *__restrict__ dst, const unsigned *__restrict__ src,
                                                           unsigned row_words, SegTable t, unsigned total_rows) {
    const unsigned w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= total_rows * row_words) return;
    unsigned row = w / row_words;
    const unsigned col = w - row * row_words;
    unsigned seg = 0;
#pragma unroll 1
    while (seg + 1 < t.n && row >= t.cnt[seg]) { // rows are numbered segment after segment
        row -= t.cnt[seg];
        ++seg;
    }
    dst[(size_t)(t.dst[seg] + row) * row_words + col] = src[(size_t)(t.src[seg] + row) * row_words + col];
}

extern "C" int htfs_segment_copy(void *d_dst, const void *d_src, unsigned row_bytes, unsigned n_segments,
                                 const unsigned *src_start, const unsigned *dst_start, const unsigned *count, htf_stream stream) {
    HTF_REQUIRE(d_dst && d_src && src_start && dst_start && count, "htfs_segment_copy: null pointer");
    HTF_REQUIRE(n_segments >= 1 && n_segments <= HTFS_MAX_SEGMENTS && row_bytes % 4 == 0 && row_bytes > 0,
                "htfs_segment_copy: need 1..%d segments of whole 32-bit words", HTFS_MAX_SEGMENTS);
    SegTable t;
    t.n = 0;
    unsigned total = 0;
    for (unsigned s = 0; s < n_segments; ++s) {
        if (count[s] == 0) continue; // (empty runs would stall the segment walk)
        t.src[t.n] = src_start[s];
        t.dst[t.n] = dst_start[s];
        t.cnt[t.n] = count[s];
        total += count[s];
        ++t.n;
    }
    if (total == 0) return HTF_OK;
    const unsigned words = row_bytes / 4;
    const unsigned grid = (unsigned)(((size_t)total * words + 255) / 256);
    hipLaunchKernelGGL(segment_copy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (unsigned *)d_dst, (const unsigned *)d_src,
                       words, t, total);
    return check_launch("segment_copy_kernel");
}

extern "C" int htfs_cell_index(const void *d_pos, int dtype, unsigned Ntot, const htf_box *box, const int *ncell3,
                               unsigned *d_cell_of, htf_stream stream) {
    HTF_REQUIRE(d_pos && box && ncell3 && d_cell_of, "htfs_cell_index: null pointer");
    if (Ntot == 0) return HTF_OK;
    unsigned grid = (Ntot + 255) / 256;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((cell_index_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float4 *)d_pos, Ntot, make_sbox<float>(box), ncell3[0], ncell3[1], ncell3[2], d_cell_of, g_gate);
    else
        hipLaunchKernelGGL((cell_index_kernel<double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const double4 *)d_pos, Ntot, make_sbox<double>(box), ncell3[0], ncell3[1], ncell3[2], d_cell_of, g_gate);
    return check_launch("cell_index_kernel");
}

// what bins_finish_kernel does beside the range table when the search is part of htfs_rebuild_nlist
struct FinishArgs {
    void *pos_sorted;       // <- pos[order], tagged
    const int *order;
    unsigned n;             // particles binned
    void *ref;              // <- pos (nullable)
    unsigned n_ref;
    unsigned *counter;      // += 1 (nullable)
    htf::Reimage im;
};

static int build_nlist_impl(const void *d_pos, const void *d_pos_sorted, int dtype, unsigned N, unsigned Ntot,
                            const htf_box *box, double r_list, const int *ncell3, const int *stencil3,
                            const unsigned *d_cell_start, unsigned pitch, int type_split,
                            unsigned *d_n_neigh, unsigned *d_head_list, unsigned *d_nlist, unsigned *d_max_neigh,
                            void *d_ranges_v, htf_stream stream, const FinishArgs *fin) {
    (void)Ntot;
    HTF_REQUIRE(d_pos && d_pos_sorted && box && ncell3 && stencil3 && d_cell_start && d_n_neigh && d_head_list && d_nlist && d_max_neigh,
                "htfs_build_nlist: null pointer");
    HTF_REQUIRE(pitch > 0, "htfs_build_nlist: pitch must be > 0");
    for (int d = 0; d < 3; ++d) {
        const double w = (box->hi[d] - box->lo[d]) / ncell3[d];
        const int sw = stencil3[d];
        HTF_REQUIRE((sw == 0 && ncell3[d] == 1) || (sw >= 1 && sw <= 2 && ncell3[d] >= 2 * sw + 1 && w * sw >= r_list * (1.0 - 1e-12)),
                    "htfs_build_nlist: along %d need 1 cell, or >= 2w+1 cells with w cells spanning r_list (got %d cells of %g, stencil %d, r_list %g)",
                    d, ncell3[d], w, sw, r_list);
    }
    if (N == 0) return HTF_OK;
    // candidate ranges per cell and stencil row (2.7 MB at C3): the CALLER's table.  (Until round 5 a thread-local buffer of
    // this file, freed and re-allocated when a larger grid came along -- under the feet of any hipGraph that had captured it.)
    const unsigned ncell = (unsigned)(ncell3[0] * ncell3[1] * ncell3[2]);
    const unsigned nrow = (unsigned)((2 * stencil3[1] + 1) * (2 * stencil3[2] + 1));
    HTF_REQUIRE(d_ranges_v && ((uintptr_t)d_ranges_v & 15) == 0, "htfs_build_nlist: d_ranges must be a 16-byte aligned table of 4 * ncell * rows words");
    uint4 *d_ranges = (uint4 *)d_ranges_v;
    const RangesArgs ra = {ncell3[0], ncell3[1], ncell3[2], stencil3[0], stencil3[1], stencil3[2], (int)box->periodic[0],
                           (int)box->periodic[1], (int)box->periodic[2], d_cell_start, d_ranges, d_max_neigh};
    const unsigned nb_ranges = (ncell * nrow + 255) / 256;
    if (fin == nullptr) {
        hipLaunchKernelGGL(cell_ranges_kernel, dim3(nb_ranges), dim3(256), 0, (hipStream_t)stream, ra, g_gate);
    } else {
        const unsigned nb = nb_ranges + (fin->n + 255) / 256 + (fin->n == 0 ? 1u : 0u);
        if (dtype == HTF_F32)
            hipLaunchKernelGGL((bins_finish_kernel<float4>), dim3(nb), dim3(256), 0, (hipStream_t)stream, ra, nb_ranges, (float4 *)fin->pos_sorted,
                               (const float4 *)d_pos, fin->order, fin->n, type_split, (float4 *)fin->ref, fin->n_ref, fin->counter, fin->im, g_gate);
        else
            hipLaunchKernelGGL((bins_finish_kernel<double4>), dim3(nb), dim3(256), 0, (hipStream_t)stream, ra, nb_ranges, (double4 *)fin->pos_sorted,
                               (const double4 *)d_pos, fin->order, fin->n, type_split, (double4 *)fin->ref, fin->n_ref, fin->counter, fin->im, g_gate);
    }
    const bool fine = stencil3[0] == 2 || stencil3[1] == 2 || stencil3[2] == 2; // short ranges: 8-lane groups waste fewer lanes
#define HTFS_NL(T, V4, G)                                                                                              \
    if (shift) HTFS_NL_(T, V4, G, true); else HTFS_NL_(T, V4, G, false)
#define HTFS_NL_(T, V4, G, S)                                                                                          \
    hipLaunchKernelGGL((build_nlist_kernel<T, G, S>), dim3((N + 4 * (64 / G) - 1) / (4 * (64 / G))), dim3(256), 0, (hipStream_t)stream, \
                       (const V4 *)d_pos, (const V4 *)d_pos_sorted, N, make_sbox<T>(box), (T)(r_list * r_list), ncell3[0],  \
                       ncell3[1], ncell3[2], stencil3[0], stencil3[1], stencil3[2], d_cell_start, pitch,      \
                       type_split, d_n_neigh, d_head_list, d_nlist, d_max_neigh, (const uint4 *)d_ranges, g_gate)
#define HTFS_NLC(T, V4)                                                                                                \
    if (shift) HTFS_NLC_(T, V4, true); else HTFS_NLC_(T, V4, false)
#define HTFS_NLC_(T, V4, S)                                                                                            \
    hipLaunchKernelGGL((build_nlist_cells_kernel<T, S>), dim3((ncell * cell_split + 3) / 4), dim3(256), 0, (hipStream_t)stream, \
                       (const V4 *)d_pos_sorted, N, make_sbox<T>(box), (T)(r_list * r_list), ncell, (int)nrow, d_cell_start, pitch, \
                       type_split, d_n_neigh, d_head_list, d_nlist, d_max_neigh, (const uint4 *)d_ranges, cell_split, g_gate)
    // with >= 7 cells along every periodic axis a stencil (<= 2 cells each way, plus a particle's place inside its own cell)
    // never reaches half a box length: the near image of a candidate follows from its cell alone
    bool shift = true;
    for (int d = 0; d < 3; ++d) shift = shift && (!box->periodic[d] || ncell3[d] >= 7);
    // one wave per cell where cells hold a few particles each (any binned system: ~4 on the fine grid, ~30 on the coarse one);
    // a grid with fewer cells than a wave per SIMD keeps the walk per particle
#ifdef HTF_AB_VARIANTS // (either kernel on any grid: tests/test_gpu_standin.py runs the random boxes through both)
    static const bool per_particle = std::getenv("HTFS_NLIST_PER_PARTICLE") != nullptr, per_cell = std::getenv("HTFS_NLIST_PER_CELL") != nullptr;
#else
    constexpr bool per_particle = false, per_cell = false;
#endif
    // (re-imaged coordinates exist in the sorted copy only: the walk per particle reads the particle's own position from d_pos)
    const bool reimaged = fin != nullptr && (fin->im.L[0] > 0.0 || fin->im.L[1] > 0.0 || fin->im.L[2] > 0.0);
    const bool by_cell = per_cell || reimaged || (!per_particle && ncell >= 1024u);
    // two waves per cell on grids of up to 12 288 cells (a brick + its ghost layer at 16 k rows per rank: 7.7 k; C2: 8 k); the 30 k cells of C3 fill the chip as they are
    const unsigned cell_split = ncell <= 12288u ? 2u : 1u;
    if (by_cell) {
        if (dtype == HTF_F32) { HTFS_NLC(float, float4); } else { HTFS_NLC(double, double4); }
    } else if (dtype == HTF_F32) {
        if (fine) { HTFS_NL(float, float4, 8); } else { HTFS_NL(float, float4, 16); }
    } else {
        if (fine) { HTFS_NL(double, double4, 8); } else { HTFS_NL(double, double4, 16); }
    }
#undef HTFS_NL
#undef HTFS_NL_
#undef HTFS_NLC
#undef HTFS_NLC_
    return check_launch("build_nlist_kernel");
}

extern "C" int htfs_build_nlist(const void *d_pos, const void *d_pos_sorted, int dtype, unsigned N, unsigned Ntot,
                                const htf_box *box, double r_list, const int *ncell3, const int *stencil3,
                                const unsigned *d_cell_start, unsigned pitch, int type_split,
                                unsigned *d_n_neigh, unsigned *d_head_list, unsigned *d_nlist, unsigned *d_max_neigh,
                                void *d_ranges, htf_stream stream) {
    return build_nlist_impl(d_pos, d_pos_sorted, dtype, N, Ntot, box, r_list, ncell3, stencil3, d_cell_start, pitch, type_split, d_n_neigh,
                            d_head_list, d_nlist, d_max_neigh, d_ranges, stream, nullptr);
}

// The whole rebuild of a single-domain list -- htfs_cell_index, htfs_cell_sort, htfs_gather4_tagged, htfs_build_nlist and
// htfs_commit_rebuild on the same arguments -- in six launches instead of nine (index + count in one kernel; range table,
// sorted copy and commit in one): the small kernels are launch- and latency-bound, and a gated rebuild pays for every one of
// them even when the gate is closed.
static int rebuild_nlist_impl(const void *d_pos, int dtype, unsigned N, unsigned Ntot, const htf_box *box, double r_list, const int *ncell3,
                              const int *stencil3, unsigned *d_cell_of, unsigned *d_scratch, unsigned *d_cell_start,
                              unsigned *d_order, void *d_pos_sorted, unsigned pitch, int type_split, unsigned *d_n_neigh,
                              unsigned *d_head_list, unsigned *d_nlist, unsigned *d_max_neigh, void *d_ref,
                              unsigned *d_counter, void *d_ranges, htf_stream stream, bool scratch_clean = false,
                              const double *image_L = nullptr) {
    using namespace htf;
    HTF_REQUIRE(d_pos && box && ncell3 && stencil3 && d_cell_of && d_scratch && d_cell_start && d_order && d_pos_sorted,
                "htfs_rebuild_nlist: null pointer");
    HTF_REQUIRE(Ntot >= N, "htfs_rebuild_nlist: Ntot %u < N %u", Ntot, N);
    if (N == 0) return HTF_OK;
    const unsigned ncell = (unsigned)(ncell3[0] * ncell3[1] * ncell3[2]);
    HTF_REQUIRE(ncell > 0, "htfs_rebuild_nlist: no cells");
    hipStream_t s = (hipStream_t)stream;
    unsigned *count = d_scratch, *cursor = d_scratch + ncell;
    if (!scratch_clean)
        if (int rc = zero_counts_if_needed(count, ncell, s)) return rc; // (as htfs_cell_sort)
    const unsigned grid = (Ntot + 255) / 256;
    Reimage im = {};
    for (int d = 0; d < 3; ++d) {
        im.L[d] = image_L != nullptr && !box->periodic[d] ? image_L[d] : 0.0;
        im.c[d] = 0.5 * (box->lo[d] + box->hi[d]);
        HTF_REQUIRE(im.L[d] == 0.0 || im.L[d] >= box->hi[d] - box->lo[d], "htfs_rebuild_nlist_ghosts: image length %g along axis %d is shorter than the grid", im.L[d], d);
    }
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((cell_index_count_kernel<float>), dim3(grid), dim3(256), 0, s, (const float4 *)d_pos, Ntot, make_sbox<float>(box),
                           ncell3[0], ncell3[1], ncell3[2], d_cell_of, count, im, g_gate);
    else
        hipLaunchKernelGGL((cell_index_count_kernel<double>), dim3(grid), dim3(256), 0, s, (const double4 *)d_pos, Ntot, make_sbox<double>(box),
                           ncell3[0], ncell3[1], ncell3[2], d_cell_of, count, im, g_gate);
    hipLaunchKernelGGL(cell_scan_kernel, dim3((ncell + kScanChunk - 1) / kScanChunk), dim3(kScanThreads), 0, s, count, ncell, d_cell_start, cursor, g_gate);
    hipLaunchKernelGGL(cell_scatter_kernel, dim3(grid), dim3(256), 0, s, d_cell_of, Ntot, cursor, d_order, g_gate);
    hipLaunchKernelGGL(cell_order_kernel, dim3((ncell + 255) / 256), dim3(256), 0, s, d_cell_start, ncell, d_order, count, g_gate);
    const FinishArgs fin = {d_pos_sorted, (const int *)d_order, Ntot, d_ref, N, d_counter, im};
    return build_nlist_impl(d_pos, d_pos_sorted, dtype, N, Ntot, box, r_list, ncell3, stencil3, d_cell_start, pitch, type_split, d_n_neigh,
                            d_head_list, d_nlist, d_max_neigh, d_ranges, stream, &fin);
}

extern "C" int htfs_rebuild_nlist(const void *d_pos, int dtype, unsigned N, const htf_box *box, double r_list, const int *ncell3,
                                  const int *stencil3, unsigned *d_cell_of, unsigned *d_scratch, unsigned *d_cell_start,
                                  unsigned *d_order, void *d_pos_sorted, unsigned pitch, int type_split, unsigned *d_n_neigh,
                                  unsigned *d_head_list, unsigned *d_nlist, unsigned *d_max_neigh, void *d_ref,
                                  unsigned *d_counter, void *d_ranges, htf_stream stream) {
    return rebuild_nlist_impl(d_pos, dtype, N, N, box, r_list, ncell3, stencil3, d_cell_of, d_scratch, d_cell_start, d_order, d_pos_sorted,
                              pitch, type_split, d_n_neigh, d_head_list, d_nlist, d_max_neigh, d_ref, d_counter, d_ranges, stream);
}

// the same for a list whose candidates include ghosts behind the N local rows (and inert rows anywhere: a decomposed system's
// fixed-capacity arrays): Ntot positions binned, N rows searched and committed -- six launches where the separate calls take ten
extern "C" int htfs_rebuild_nlist_ghosts(const void *d_pos, int dtype, unsigned N, unsigned Ntot, const htf_box *box, double r_list,
                                         const int *ncell3, const int *stencil3, unsigned *d_cell_of, unsigned *d_scratch,
                                         unsigned *d_cell_start, unsigned *d_order, void *d_pos_sorted, unsigned pitch, int type_split,
                                         unsigned *d_n_neigh, unsigned *d_head_list, unsigned *d_nlist, unsigned *d_max_neigh, void *d_ref,
                                         unsigned *d_counter, void *d_ranges, int scratch_clean, const double *image_L, htf_stream stream) {
    return rebuild_nlist_impl(d_pos, dtype, N, Ntot, box, r_list, ncell3, stencil3, d_cell_of, d_scratch, d_cell_start, d_order, d_pos_sorted,
                              pitch, type_split, d_n_neigh, d_head_list, d_nlist, d_max_neigh, d_ref, d_counter, d_ranges, stream,
                              scratch_clean != 0, image_L);
}

// A whole check step of a device-decided list in ONE call (the host's share of a small system's step is its enqueue): the
// displacement word zeroed and filled, the gate set to it, htfs_rebuild_nlist behind the gate, the gate taken off, and the two
// status words copied to pinned host memory for the NEXT check to read (nullable).
extern "C" int htfs_check_rebuild_nlist(const void *d_pos, int dtype, unsigned N, const htf_box *box, double r_list, const int *ncell3,
                                        const int *stencil3, unsigned *d_cell_of, unsigned *d_scratch, unsigned *d_cell_start,
                                        unsigned *d_order, void *d_pos_sorted, unsigned pitch, int type_split, unsigned *d_n_neigh,
                                        unsigned *d_head_list, unsigned *d_nlist, unsigned *d_stat2, void *d_ref, float *d_disp2,
                                        double threshold2, unsigned *h_stat2, void *d_ranges, htf_stream stream) {
    HTF_REQUIRE(d_ref && d_disp2 && d_stat2, "htfs_check_rebuild_nlist: null pointer");
    if (N == 0) return HTF_OK;
    HTF_CHECK_HIP(hipMemsetAsync(d_disp2, 0, sizeof(float), (hipStream_t)stream));
    int rc = htfs_max_displacement2(d_pos, d_ref, dtype, N, box, d_disp2, stream);
    if (rc != HTF_OK) return rc;
    rc = htfs_set_gate(d_disp2, threshold2);
    if (rc != HTF_OK) return rc;
    rc = htfs_rebuild_nlist(d_pos, dtype, N, box, r_list, ncell3, stencil3, d_cell_of, d_scratch, d_cell_start, d_order, d_pos_sorted, pitch,
                            type_split, d_n_neigh, d_head_list, d_nlist, d_stat2, d_ref, d_stat2 + 1, d_ranges, stream);
    (void)htfs_set_gate(nullptr, 0.0);
    if (rc != HTF_OK) return rc;
    if (h_stat2 != nullptr)
        HTF_CHECK_HIP(hipMemcpyAsync(h_stat2, d_stat2, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    return HTF_OK;
}

namespace htf {
// the tail of a (conditional) rebuild: reference positions <- current positions, rebuild counter + 1
template <typename V>
__global__ __launch_bounds__(256) void commit_rebuild_kernel(V *__restrict__ ref, const V *__restrict__ pos, unsigned n,
                                                             unsigned *__restrict__ counter, Gate gate) {
    if (gate.closed()) return;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ref[i] = pos[i];
    if (i == 0 && counter != nullptr) *counter += 1u;
}
} // namespace htf

extern "C" int htfs_set_gate(const float *d_disp2, double threshold2) {
    g_gate.disp2 = d_disp2;
    g_gate.thr2 = (float)threshold2;
    return HTF_OK;
}

extern "C" int htfs_commit_rebuild(void *d_ref, const void *d_pos, int dtype, unsigned N, unsigned *d_counter,
                                   htf_stream stream) {
    HTF_REQUIRE(d_ref && d_pos, "htfs_commit_rebuild: null pointer");
    const unsigned grid = (N + 255) / 256 > 0 ? (N + 255) / 256 : 1;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((commit_rebuild_kernel<float4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (float4 *)d_ref, (const float4 *)d_pos, N, d_counter, g_gate);
    else
        hipLaunchKernelGGL((commit_rebuild_kernel<double4>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (double4 *)d_ref, (const double4 *)d_pos, N, d_counter, g_gate);
    return check_launch("commit_rebuild_kernel");
}
