// Brick decomposition with fixed-capacity arrays (include/htf_standin.h, hoomd_tf_amd/brick.py): the migration + ghost plan of
// a rebuild as kernels whose every size is a CAPACITY -- particle counts, message counts and class boundaries never leave the
// device, so a rebuild needs no host synchronisation and a whole check period of the decomposed step, rebuild included, can be
// captured into one hipGraph.  The stand-in for HOOMD's Communicator (migrateParticles + exchangeGhosts + the per-step ghost
// update; the reference inherits them from HOOMD under MPI, test_mpi_tensorflow.py:57-79); outside the drop-in boundary.
#include "htf_common.h"
#include "box_math.h"
#include "htf_standin.h"
#include "standin_gate.h"
#include "key_sort.h"

namespace htf {

// by-value copy of htfs_brick in the positions' precision
template <typename T>
struct BrickArgs {
    int ndim, n_msg, replica;
    int axis[2], p[2], me[2];
    T r_ghost;
    unsigned cap_int, cap_bnd;
    unsigned ghost_cap[HTFS_BRICK_MAX_MSG], ghost_off[HTFS_BRICK_MAX_MSG];
    unsigned mig_cap[HTFS_BRICK_MAX_MSG], mig_off[HTFS_BRICK_MAX_MSG];
    T shift[HTFS_BRICK_MAX_MSG][3], mig_shift[HTFS_BRICK_MAX_MSG][3];
    int halo_wrap, mig_wrap;
    T box_lo[3], box_L[3], box_Linv[3];
};

template <typename T>
static BrickArgs<T> make_args(const htfs_brick *g) {
    BrickArgs<T> a;
    a.ndim = g->ndim;
    a.n_msg = g->n_msg;
    a.replica = g->replica;
    for (int d = 0; d < 2; ++d) {
        a.axis[d] = g->axis[d];
        a.p[d] = g->p[d];
        a.me[d] = g->me[d];
    }
    a.r_ghost = (T)g->r_ghost;
    a.cap_int = g->cap_int;
    a.cap_bnd = g->cap_bnd;
    for (int m = 0; m < HTFS_BRICK_MAX_MSG; ++m) {
        a.ghost_cap[m] = g->ghost_cap[m];
        a.ghost_off[m] = g->ghost_off[m];
        a.mig_cap[m] = g->mig_cap[m];
        a.mig_off[m] = g->mig_off[m];
        for (int c = 0; c < 3; ++c) {
            a.shift[m][c] = (T)g->shift[m][c];
            a.mig_shift[m][c] = (T)g->mig_shift[m][c];
        }
    }
    a.halo_wrap = g->halo_wrap;
    a.mig_wrap = g->mig_wrap;
    for (int c = 0; c < 3; ++c) {
        a.box_lo[c] = (T)g->box_lo[c];
        a.box_L[c] = (T)g->box_L[c];
        a.box_Linv[c] = (T)1 / a.box_L[c];
    }
    return a;
}

template <typename V>
__device__ __forceinline__ auto comp(const V &p, int axis) -> decltype(p.x) { return axis == 0 ? p.x : (axis == 1 ? p.y : p.z); }

// offset digit (0, 1, 2 <-> -1, 0, +1) of message m along decomposed axis d
__device__ __forceinline__ int msg_digit(int m, int ndim, int d) {
    const int centre = ndim == 1 ? 1 : 4;
    const int raw = m < centre ? m : m + 1;
    return d == 0 ? raw % 3 : raw / 3;
}

// does a particle of class key c (k_0 + 4 k_1) travel in halo message m?
__device__ __forceinline__ bool msg_takes_class(int m, int ndim, unsigned c) {
    bool ok = true;
    for (int d = 0; d < ndim; ++d) {
        const int o = msg_digit(m, ndim, d) - 1;
        const unsigned k = (c >> (2 * d)) & 3u;
        ok = ok && (o == 0 || (o < 0 ? (k == 1u || k == 2u) : (k == 2u || k == 3u)));
    }
    return ok;
}

// a position as a message carries it: shifted by `sh` and, if asked, wrapped back into the global box (the integrator's wrap)
template <typename T, typename V4>
__device__ __forceinline__ V4 shifted(V4 p, const BrickArgs<T> &a, const T (&sh)[3], int wrap) {
    if (sh[0] != (T)0) p.x = wrap ? wrap1<T>(p.x + sh[0], a.box_lo[0], a.box_L[0], a.box_Linv[0], 1) : p.x + sh[0];
    if (sh[1] != (T)0) p.y = wrap ? wrap1<T>(p.y + sh[1], a.box_lo[1], a.box_L[1], a.box_Linv[1], 1) : p.y + sh[1];
    if (sh[2] != (T)0) p.z = wrap ? wrap1<T>(p.z + sh[2], a.box_lo[2], a.box_L[2], a.box_Linv[2], 1) : p.z + sh[2];
    return p;
}

template <typename V>
__device__ __forceinline__ V inert_position() {
    V p;
    const auto nan = __builtin_nanf("");
    p.x = nan, p.y = nan, p.z = nan, p.w = 0;
    return p;
}

// ---- K1: destination key of every local row: 0 stay | 1 + message index | n_msg + 1 inert.  The arithmetic is SlabDomain's
// (slab_classify_kernel) per decomposed axis, in the positions' own precision: owner = #(interior cuts <= x).
constexpr unsigned kBrickTile = 1024; // rows per sorting tile (a 256-thread block): ~40 tiles at 16 k rows per rank + ghosts

template <typename T, typename V4>
__global__ __launch_bounds__(256) void brick_dest_kernel(const V4 *__restrict__ pos, unsigned cap, BrickArgs<T> a,
                                                         const T *__restrict__ bounds, unsigned *__restrict__ key,
                                                         unsigned *__restrict__ tile_hist, unsigned *__restrict__ counts) {
    __shared__ unsigned h[16];
    tile_hist_begin<16>(h);
    for (unsigned r = 0; r < kBrickTile / 256; ++r) {
        const unsigned i = blockIdx.x * kBrickTile + r * 256 + threadIdx.x;
        if (i >= cap) continue;
        const V4 p = pos[i];
        unsigned k;
        if (is_inert(p.x)) {
            k = (unsigned)a.n_msg + 1u;
        } else {
            int raw = 0, mul = 1;
            bool stay = true, lost = false;
            for (int d = 0; d < a.ndim; ++d) {
                const T x = comp(p, a.axis[d]);
                const T *b = bounds + d * (HTFS_BRICK_MAX_P + 1);
                int owner = 0;
                for (int c = 1; c < a.p[d]; ++c) owner += (b[c] <= x) ? 1 : 0;
                int off;
                if (a.p[d] == 2 && !a.replica)
                    off = owner != a.me[d] ? 1 : 0; // both faces lead to the one peer: everything that leaves travels "up"
                else if (a.p[d] == 2) {
                    // a replica brick is shifted by the face it leaves through: which one, from the side of the brick's centre the
                    // particle is on (minimum image of the logical box)
                    const T L = b[2] - b[0];
                    T dx = x - (T)0.5 * (b[a.me[d]] + b[a.me[d] + 1]);
                    dx -= L * rint(dx / L);
                    off = owner != a.me[d] ? (dx < (T)0 ? -1 : 1) : 0;
                } else if (owner == a.me[d])
                    off = 0;
                else if (owner == (a.me[d] + a.p[d] - 1) % a.p[d])
                    off = -1;
                else if (owner == (a.me[d] + 1) % a.p[d])
                    off = 1;
                else {
                    off = 0;
                    lost = true;
                }
                stay = stay && off == 0;
                raw += (off + 1) * mul;
                mul *= 3;
            }
            if (lost) atomicOr(&counts[HTFS_BC_FLAGS], (unsigned)HTFS_BF_LOST);
            const int centre = a.ndim == 1 ? 1 : 4;
            k = (stay || lost) ? 0u : 1u + (unsigned)(raw < centre ? raw : raw - 1);
        }
        key[i] = k;
        atomicAdd(&h[k], 1u);
    }
    tile_hist_end<16>(h, tile_hist);
}

// ---- K3: migrants into their messages; row 0 of a message is its header (count in the first word)
template <typename T, typename V4>
__global__ __launch_bounds__(256) void brick_pack_mig_kernel(const V4 *__restrict__ pos, const V4 *__restrict__ vel, BrickArgs<T> a,
                                                             const unsigned *__restrict__ order, const unsigned *__restrict__ start1,
                                                             V4 *__restrict__ send, unsigned total_rows, unsigned *__restrict__ counts) {
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= total_rows) return;
    int m = 0;
    while (m + 1 < a.n_msg && r >= a.mig_off[m + 1]) ++m;
    const unsigned j = r - a.mig_off[m];
    const unsigned first = start1[1 + m], n = start1[2 + m] - first;
    const unsigned room = a.mig_cap[m] - 1u;
    if (j == 0) {
        if (n > room) atomicOr(&counts[HTFS_BC_FLAGS], (unsigned)HTFS_BF_MIG_OVERFLOW);
        *reinterpret_cast<unsigned *>(&send[2 * (size_t)r]) = n < room ? n : room;
        return;
    }
    if (j - 1u >= n) return;
    const unsigned src = order[first + j - 1u];
    send[2 * (size_t)r] = shifted<T>(pos[src], a, a.mig_shift[m], a.mig_wrap);
    send[2 * (size_t)r + 1] = vel[src];
}

// ---- K4: the candidates of this brick after the exchange -- [stayed | from offset index n_msg-1 | ... | from index 0] -- copied
// into the scratch arrays with their class key in THIS brick (k_0 + 4 k_1; 4^ndim = nothing here)
template <typename T, typename V4, unsigned NK>
__global__ __launch_bounds__(256) void brick_home_kernel(const V4 *__restrict__ pos, const V4 *__restrict__ vel, BrickArgs<T> a,
                                                         const T *__restrict__ bounds, const unsigned *__restrict__ order,
                                                         const unsigned *__restrict__ start1, const V4 *__restrict__ recv,
                                                         V4 *__restrict__ tmp_pos, V4 *__restrict__ tmp_vel, unsigned *__restrict__ key2,
                                                         unsigned cand_cap, unsigned *__restrict__ tile_hist, unsigned *__restrict__ counts) {
    __shared__ unsigned h[NK];
    tile_hist_begin<NK>(h);
    const unsigned dead_key = a.ndim == 1 ? 4u : 16u;
    const unsigned n_stay = start1[1];
    for (unsigned rr = 0; rr < kBrickTile / 256; ++rr) {
        const unsigned c = blockIdx.x * kBrickTile + rr * 256 + threadIdx.x;
        if (c >= cand_cap) continue;
        V4 p, v;
        bool live = true;
        if (c < n_stay) {
            const unsigned src = order[c];
            p = pos[src];
            v = vel[src];
        } else {
            unsigned j = c - n_stay, arrived = 0;
            live = false;
            for (int m = a.n_msg - 1; m >= 0; --m) {
                const unsigned n = *reinterpret_cast<const unsigned *>(&recv[2 * (size_t)a.mig_off[m]]);
                arrived += n;
                if (!live && j < n) {
                    live = true;
                    p = recv[2 * (size_t)(a.mig_off[m] + 1u + j)];
                    v = recv[2 * (size_t)(a.mig_off[m] + 1u + j) + 1];
                }
                if (!live) j -= n;
            }
            if (c == n_stay) { // (one thread: the totals of this rebuild)
                counts[HTFS_BC_N_CAND] = n_stay + arrived;
                counts[HTFS_BC_N_ARRIVED] += arrived;
                counts[HTFS_BC_REBUILDS] += 1u;
            }
        }
        unsigned k = dead_key;
        if (live) {
            k = 0;
            for (int d = 0; d < a.ndim; ++d) {
                const T x = comp(p, a.axis[d]);
                const T *b = bounds + d * (HTFS_BRICK_MAX_P + 1);
                const bool near_lo = x < b[a.me[d]] + a.r_ghost, near_hi = x >= b[a.me[d] + 1] - a.r_ghost;
                k |= (near_lo ? (near_hi ? 2u : 1u) : (near_hi ? 3u : 0u)) << (2 * d);
            }
            tmp_pos[c] = p;
            tmp_vel[c] = v;
        }
        key2[c] = k;
        atomicAdd(&h[k], 1u);
    }
    tile_hist_end<NK>(h, tile_hist);
}

// ---- K6: the two segments, inert rows behind the particles; counts, class boundaries, message sizes, overflow flags
template <typename T, typename V4>
__global__ __launch_bounds__(256) void brick_place_kernel(V4 *__restrict__ pos, V4 *__restrict__ vel, BrickArgs<T> a,
                                                          const unsigned *__restrict__ order2, const unsigned *__restrict__ start2,
                                                          const V4 *__restrict__ tmp_pos, const V4 *__restrict__ tmp_vel,
                                                          unsigned *__restrict__ n_neigh, unsigned *__restrict__ counts) {
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned nclass = a.ndim == 1 ? 4u : 16u;
    const unsigned n_int = start2[1], n_live = start2[nclass], n_bnd = n_live - n_int;
    if (r == 0) {
        counts[HTFS_BC_N_INT] = n_int < a.cap_int ? n_int : a.cap_int;
        counts[HTFS_BC_N_BND] = n_bnd < a.cap_bnd ? n_bnd : a.cap_bnd;
        unsigned f = 0;
        if (n_int > a.cap_int) f |= (unsigned)HTFS_BF_INT_OVERFLOW;
        if (n_bnd > a.cap_bnd) f |= (unsigned)HTFS_BF_BND_OVERFLOW;
        if (f) atomicOr(&counts[HTFS_BC_FLAGS], f);
    }
    if (r <= nclass) counts[HTFS_BC_CLASS + r] = start2[r];
    if (r < (unsigned)a.n_msg) {
        unsigned n = 0;
        for (unsigned c = 1; c < nclass; ++c)
            if (msg_takes_class((int)r, a.ndim, c)) n += start2[c + 1] - start2[c];
        if (n > a.ghost_cap[r]) {
            atomicOr(&counts[HTFS_BC_FLAGS], (unsigned)HTFS_BF_GHOST_OVERFLOW);
            n = a.ghost_cap[r];
        }
        counts[HTFS_BC_MSG + r] = n;
    }
    if (r < 16u * HTFS_BRICK_MAX_MSG) { // where class c starts in message m: what brick_nve_halo_kernel looks a row's slots up in
        const unsigned c = r / HTFS_BRICK_MAX_MSG, m = r % HTFS_BRICK_MAX_MSG;
        unsigned first = 0xFFFFFFFFu;
        if (c >= 1u && c < nclass && m < (unsigned)a.n_msg && msg_takes_class((int)m, a.ndim, c)) {
            first = 0;
            for (unsigned cc = 1; cc < c; ++cc)
                if (msg_takes_class((int)m, a.ndim, cc)) first += start2[cc + 1] - start2[cc];
        }
        counts[HTFS_BC_SLOT + r] = first;
    }
    if (r >= a.cap_int + a.cap_bnd) return;
    const bool interior = r < a.cap_int;
    const unsigned j = interior ? r : r - a.cap_int;
    const bool live = j < (interior ? n_int : n_bnd);
    if (live) {
        const unsigned src = order2[(interior ? 0u : n_int) + j];
        pos[r] = tmp_pos[src];
        vel[r] = tmp_vel[src];
    } else {
        pos[r] = inert_position<V4>();
        V4 v;
        v.x = 0, v.y = 0, v.z = 0, v.w = 1;
        vel[r] = v;
        if (n_neigh != nullptr) n_neigh[r] = 0u;
    }
}

// ---- transport "peer": rows stored straight into the receiver's inbox, a signal per message (htf_standin.h htfs_peer)
struct PeerArgs {
    void *inbox[HTFS_BRICK_MAX_MSG];
    unsigned *signal[HTFS_BRICK_MAX_MSG];
    void *my_inbox;
    unsigned *my_signal;
    unsigned *state;      // [0] exchanges done, [1] workgroups finished (this launch), [2] timeouts
    unsigned spin_limit;
    unsigned rows;        // ghost rows per half of an inbox (sum of the message capacities)
};

static PeerArgs make_peer(const htfs_peer *p, const htfs_brick *g) {
    PeerArgs a;
    for (int m = 0; m < HTFS_BRICK_MAX_MSG; ++m) {
        a.inbox[m] = p ? p->inbox[m] : nullptr;
        a.signal[m] = p ? p->signal[m] : nullptr;
    }
    a.my_inbox = p ? p->my_inbox : nullptr;
    a.my_signal = p ? p->my_signal : nullptr;
    a.state = p ? p->state : nullptr;
    a.spin_limit = p ? p->spin_limit : 0u;
    a.rows = g->ghost_off[g->n_msg - 1] + g->ghost_cap[g->n_msg - 1];
    return a;
}

// where row `slot` of MY message m lands in the receiver's inbox: half (seq & 1), the region of source offset n_msg - 1 - m
template <typename V4>
__device__ __forceinline__ V4 *peer_slot(const PeerArgs &pa, int m, int n_msg, const unsigned *ghost_off, unsigned seq, unsigned slot) {
    return reinterpret_cast<V4 *>(pa.inbox[m]) + (size_t)(seq & 1u) * pa.rows + ghost_off[n_msg - 1 - m] + slot;
}

// Called by EVERY thread at the end of a packing kernel: when the launch's last workgroup has stored its rows, publish each
// message's row count and the new sequence number to its receiver (release at system scope: the rows are visible before the
// number is) and advance this rank's own counter.
__device__ __forceinline__ void peer_publish(const PeerArgs &pa, int n_msg, const unsigned *counts, unsigned seq) {
    // ONE system-scope fence per workgroup, behind its barrier (a fence per thread wrote the L2 back 256 times per workgroup:
    // +12 us on a 30 us step); the rows themselves were stored past the caches (store_stream)
    __syncthreads();
    if (threadIdx.x != 0) return;
    __threadfence_system();
    const unsigned done = __hip_atomic_fetch_add(&pa.state[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done + 1u != gridDim.x) return;
    pa.state[1] = 0u;
    for (int m = 0; m < n_msg; ++m) {
        unsigned *sig = pa.signal[m] + 2 * (n_msg - 1 - m);
        __hip_atomic_store(sig + 1, counts[HTFS_BC_MSG + m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(sig, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(&pa.state[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- K8 (every step): the halo messages, packed; a message's rows beyond its count are inert
template <typename T, typename V4>
__global__ __launch_bounds__(256) void brick_pack_halo_peer_kernel(const V4 *__restrict__ pos, BrickArgs<T> a, const unsigned *__restrict__ counts,
                                                                   PeerArgs pa, unsigned total_rows) {
    const unsigned seq = pa.state[0] + 1u; // (every workgroup reads it before the launch's last one advances it)
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < total_rows) {
        int m = 0;
        while (m + 1 < a.n_msg && r >= a.ghost_off[m + 1]) ++m;
        unsigned j = r - a.ghost_off[m];
        const unsigned slot = j;
        if (j < counts[HTFS_BC_MSG + m]) { // (rows beyond the count are not sent: the receiver fills them from the count)
            const unsigned nclass = a.ndim == 1 ? 4u : 16u;
            const unsigned n_int = counts[HTFS_BC_CLASS + 1];
            for (unsigned c = 1; c < nclass; ++c) {
                if (!msg_takes_class(m, a.ndim, c)) continue;
                const unsigned first = counts[HTFS_BC_CLASS + c], n = counts[HTFS_BC_CLASS + c + 1] - first;
                if (j < n) {
                    store_stream(peer_slot<V4>(pa, m, a.n_msg, a.ghost_off, seq, slot), shifted<T>(pos[a.cap_int + (first - n_int) + j], a, a.shift[m], a.halo_wrap));
                    break;
                }
                j -= n;
            }
        }
    }
    peer_publish(pa, a.n_msg, counts, seq);
}

// the receiving side: wait for the sequence number of every incoming message, then its rows (inert beyond the sender's count)
// from the inbox half of that number into the ghost region
template <typename T, typename V4>
__global__ __launch_bounds__(256) void brick_unpack_halo_kernel(V4 *__restrict__ pos, BrickArgs<T> a, PeerArgs pa, unsigned total_rows,
                                                                unsigned *__restrict__ counts) {
    const unsigned seq = pa.state[0]; // (advanced by this step's packing kernel, earlier in the stream)
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    // ONE poller per workgroup (a 256-row workgroup may straddle messages: it waits for every message its rows touch), relaxed
    // polls and a single acquire fence behind them: an acquire per poll and per wave invalidated the caches hundreds of times
    __shared__ int s_late;
    const unsigned rb = min(blockIdx.x * blockDim.x, total_rows - 1u), re = min(blockIdx.x * blockDim.x + blockDim.x - 1u, total_rows - 1u);
    int j0 = 0;
    while (j0 + 1 < a.n_msg && rb >= a.ghost_off[j0 + 1]) ++j0;
    if (threadIdx.x == 0) {
        int jl = j0;
        while (jl + 1 < a.n_msg && re >= a.ghost_off[jl + 1]) ++jl;
        int late_ = 0;
        for (int j = j0; j <= jl && !late_; ++j) {
            unsigned n = 0;
            while ((int)(__hip_atomic_load(pa.my_signal + 2 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
                if (++n > pa.spin_limit) {
                    late_ = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __threadfence_system(); // (acquire side: what the senders stored before their signals is visible to the loads below)
        if (late_) {
            atomicOr(&counts[HTFS_BC_FLAGS], (unsigned)HTFS_BF_HALO_TIMEOUT);
            atomicAdd(&pa.state[2], 1u);
        }
        s_late = late_;
    }
    __syncthreads();
    const bool late = s_late != 0;
    if (r >= total_rows) return;
    int j = j0;
    while (j + 1 < a.n_msg && r >= a.ghost_off[j + 1]) ++j;
    const unsigned slot = r - a.ghost_off[j];
    const unsigned cnt = __hip_atomic_load(pa.my_signal + 2 * j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    V4 p = inert_position<V4>();
    if (slot < cnt && !late) {
        // (read past the caches: the rows were written by another agent -- or another process -- behind the acquire above)
        p = load_stream(reinterpret_cast<const V4 *>(pa.my_inbox) + (size_t)(seq & 1u) * pa.rows + r);
    }
    pos[a.cap_int + a.cap_bnd + r] = p;
}

template <typename T, typename V4>
__global__ __launch_bounds__(256) void brick_pack_halo_kernel(const V4 *__restrict__ pos, BrickArgs<T> a, const unsigned *__restrict__ counts,
                                                              V4 *__restrict__ send, V4 *__restrict__ direct, unsigned total_rows) {
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= total_rows) return;
    int m = 0;
    while (m + 1 < a.n_msg && r >= a.ghost_off[m + 1]) ++m;
    unsigned j = r - a.ghost_off[m];
    V4 p = inert_position<V4>();
    if (j < counts[HTFS_BC_MSG + m]) {
        const unsigned nclass = a.ndim == 1 ? 4u : 16u;
        const unsigned n_int = counts[HTFS_BC_CLASS + 1];
        for (unsigned c = 1; c < nclass; ++c) {
            if (!msg_takes_class(m, a.ndim, c)) continue;
            const unsigned first = counts[HTFS_BC_CLASS + c], n = counts[HTFS_BC_CLASS + c + 1] - first;
            if (j < n) {
                p = shifted<T>(pos[a.cap_int + (first - n_int) + j], a, a.shift[m], a.halo_wrap);
                break;
            }
            j -= n;
        }
    }
    if (send != nullptr) send[r] = p;
    // this rank as its own neighbor: message m (to offset o) is what it receives from offset -o, index n_msg - 1 - m
    if (direct != nullptr) direct[a.ghost_off[a.n_msg - 1 - m] + (r - a.ghost_off[m])] = p;
}

// ---- every step, instead of nve_step + K8: the leapfrog update of every local row (nve_advance: the stand-in integrator's own
// arithmetic) and, for a boundary row, its new position straight into every halo message that carries it -- the row knows its
// class from the class boundaries and its slot in a message from the counts of the classes before it.  One launch where the
// integrator and the packer were two (each >= 4.5 us inside a hipGraph whatever it moves).  The messages' inert tails were
// written by the rebuild's own pack and stay put until the next one.
// The counts block (HTFS_BC_WORDS words: class boundaries, message counts, slot table) staged in LDS by the whole workgroup with ONE
// round of loads, issued beside the rows' own: the per-step kernels used to walk it with dependent global loads -- boundary count ->
// class boundaries (a loop) -> slot table -> message counts, four or five trips to the L2 in a kernel that is otherwise one.
__device__ __forceinline__ void stage_counts(unsigned *lds, const unsigned *__restrict__ counts) {
    for (unsigned w = threadIdx.x; w < (unsigned)HTFS_BC_WORDS; w += blockDim.x) lds[w] = counts[w];
}

template <typename T, typename V4>
__global__ __launch_bounds__(256) void brick_nve_halo_peer_kernel(V4 *__restrict__ pos, V4 *__restrict__ vel, const V4 *__restrict__ force,
                                                                  T dt, SBox<T> box, BrickArgs<T> a, const unsigned *__restrict__ counts,
                                                                  PeerArgs pa) {
    __shared__ unsigned cnt[HTFS_BC_WORDS];
    const unsigned seq = pa.state[0] + 1u;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool row = i < a.cap_int + a.cap_bnd;
    V4 p, v, f;
    if (row) p = pos[i], v = vel[i], f = force[i];
    stage_counts(cnt, counts);
    __syncthreads();
    if (row) {
        nve_advance<T>(p, v, f, dt, box);
        pos[i] = p;
        vel[i] = v;
        const unsigned j = i - a.cap_int;
        if (i >= a.cap_int && j < cnt[HTFS_BC_N_BND]) {
            const unsigned nclass = a.ndim == 1 ? 4u : 16u;
            const unsigned n_int = cnt[HTFS_BC_CLASS + 1];
            unsigned c = 1;
            while (c + 1 < nclass && j >= cnt[HTFS_BC_CLASS + c + 1] - n_int) ++c;
            const unsigned in_class = j - (cnt[HTFS_BC_CLASS + c] - n_int);
            for (int m = 0; m < a.n_msg; ++m) {
                const unsigned first = cnt[HTFS_BC_SLOT + c * HTFS_BRICK_MAX_MSG + m];
                if (first == 0xFFFFFFFFu) continue;
                const unsigned slot = first + in_class;
                if (slot >= cnt[HTFS_BC_MSG + m]) continue;
                store_stream(peer_slot<V4>(pa, m, a.n_msg, a.ghost_off, seq, slot), shifted<T>(p, a, a.shift[m], a.halo_wrap));
            }
        }
    }
    peer_publish(pa, a.n_msg, counts, seq);
}

template <typename T, typename V4>
__global__ __launch_bounds__(256) void brick_nve_halo_kernel(V4 *__restrict__ pos, V4 *__restrict__ vel, const V4 *__restrict__ force,
                                                             T dt, SBox<T> box, BrickArgs<T> a, const unsigned *__restrict__ counts,
                                                             V4 *__restrict__ send, V4 *__restrict__ direct) {
    __shared__ unsigned cnt[HTFS_BC_WORDS];
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool row = i < a.cap_int + a.cap_bnd;
    V4 p, v, f;
    if (row) p = pos[i], v = vel[i], f = force[i];
    const bool boundary_block = (blockIdx.x + 1u) * blockDim.x > a.cap_int; // (workgroup-uniform: interior-only workgroups skip the table)
    if (boundary_block) {
        stage_counts(cnt, counts);
        __syncthreads();
    }
    if (!row) return;
    nve_advance<T>(p, v, f, dt, box);
    pos[i] = p;
    vel[i] = v;
    if (i < a.cap_int) return;
    const unsigned j = i - a.cap_int;
    if (j >= cnt[HTFS_BC_N_BND]) return;
    const unsigned nclass = a.ndim == 1 ? 4u : 16u;
    const unsigned n_int = cnt[HTFS_BC_CLASS + 1];
    unsigned c = 1;
    while (c + 1 < nclass && j >= cnt[HTFS_BC_CLASS + c + 1] - n_int) ++c;
    const unsigned in_class = j - (cnt[HTFS_BC_CLASS + c] - n_int);
    for (int m = 0; m < a.n_msg; ++m) {
        const unsigned first = cnt[HTFS_BC_SLOT + c * HTFS_BRICK_MAX_MSG + m]; // (written by the rebuild's place kernel)
        if (first == 0xFFFFFFFFu) continue;
        const unsigned slot = first + in_class;
        if (slot >= cnt[HTFS_BC_MSG + m]) continue; // (beyond the message's capacity: flagged by the rebuild)
        const V4 q = shifted<T>(p, a, a.shift[m], a.halo_wrap);
        if (send != nullptr) send[a.ghost_off[m] + slot] = q;
        if (direct != nullptr) direct[a.ghost_off[a.n_msg - 1 - m] + slot] = q;
    }
}

// the slot of every boundary row in every halo message, once per re-plan (what brick_nve_halo_kernel works out per row and step):
// the table the one-kernel step's epilogue reads (htf_internal.h step_epilogue_lane)
template <typename T>
__global__ __launch_bounds__(256) void brick_row_slots_kernel(BrickArgs<T> a, const unsigned *__restrict__ counts, unsigned *__restrict__ row_slots) {
    __shared__ unsigned cnt[HTFS_BC_WORDS];
    stage_counts(cnt, counts);
    __syncthreads();
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.cap_bnd) return;
    unsigned out[HTFS_BRICK_MAX_MSG];
#pragma unroll
    for (int m = 0; m < HTFS_BRICK_MAX_MSG; ++m) out[m] = 0xFFFFFFFFu;
    if (j < cnt[HTFS_BC_N_BND]) {
        const unsigned nclass = a.ndim == 1 ? 4u : 16u;
        const unsigned n_int = cnt[HTFS_BC_CLASS + 1];
        unsigned c = 1;
        while (c + 1 < nclass && j >= cnt[HTFS_BC_CLASS + c + 1] - n_int) ++c;
        const unsigned in_class = j - (cnt[HTFS_BC_CLASS + c] - n_int);
#pragma unroll
        for (int m = 0; m < HTFS_BRICK_MAX_MSG; ++m) {
            if (m >= a.n_msg) continue;
            const unsigned first = cnt[HTFS_BC_SLOT + c * HTFS_BRICK_MAX_MSG + m];
            if (first == 0xFFFFFFFFu) continue;
            const unsigned slot = first + in_class;
            if (slot < cnt[HTFS_BC_MSG + m]) out[m] = slot;
        }
    }
    uint4 *dst = reinterpret_cast<uint4 *>(row_slots) + (size_t)j * 2;
    dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
    dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
}

static int check_geom(const htfs_brick *g, const char *who) {
    HTF_REQUIRE(g, "%s: null geometry", who);
    HTF_REQUIRE((g->ndim == 1 && g->n_msg == 2) || (g->ndim == 2 && g->n_msg == 8), "%s: ndim %d with %d messages (1 / 2 or 2 / 8)", who,
                g->ndim, g->n_msg);
    for (int d = 0; d < g->ndim; ++d)
        HTF_REQUIRE(g->axis[d] >= 0 && g->axis[d] < 3 && g->p[d] >= 2 && g->p[d] <= HTFS_BRICK_MAX_P && g->me[d] >= 0 && g->me[d] < g->p[d],
                    "%s: axis %d: %d bricks, coordinate %d", who, g->axis[d], g->p[d], g->me[d]);
    for (int m = 0; m < g->n_msg; ++m) {
        HTF_REQUIRE(g->mig_cap[m] >= 2 && g->ghost_cap[m] >= 1, "%s: message %d has no room", who, m);
        HTF_REQUIRE(g->ghost_cap[m] == g->ghost_cap[g->n_msg - 1 - m] && g->mig_cap[m] == g->mig_cap[g->n_msg - 1 - m],
                    "%s: capacities of opposite messages must agree (%d)", who, m);
        if (m > 0)
            HTF_REQUIRE(g->mig_off[m] == g->mig_off[m - 1] + g->mig_cap[m - 1] && g->ghost_off[m] == g->ghost_off[m - 1] + g->ghost_cap[m - 1],
                        "%s: message %d does not follow message %d", who, m, m - 1);
        else
            HTF_REQUIRE(g->mig_off[0] == 0 && g->ghost_off[0] == 0, "%s: message 0 must start at row 0", who);
    }
    HTF_REQUIRE(g->cap_int + g->cap_bnd > 0, "%s: no local rows", who);
    return HTF_OK;
}

} // namespace htf

using namespace htf;

extern "C" int htfs_brick_migrate_pack(const htfs_brick *g, const void *d_pos, const void *d_vel, int dtype, const void *d_bounds,
                                       const htfs_brick_work *w, void *d_mig_send, unsigned *d_counts, htf_stream stream) {
    if (int rc = check_geom(g, "htfs_brick_migrate_pack")) return rc;
    HTF_REQUIRE(d_pos && d_vel && d_bounds && w && w->key && w->order && w->sort_scratch && w->start1 && d_mig_send && d_counts,
                "htfs_brick_migrate_pack: null pointer");
    HTF_REQUIRE(dtype == HTF_F32 || dtype == HTF_F64, "htfs_brick_migrate_pack: bad dtype %d", dtype);
    hipStream_t s = (hipStream_t)stream;
    const unsigned cap = g->cap_int + g->cap_bnd;
    const unsigned mig_rows = g->mig_off[g->n_msg - 1] + g->mig_cap[g->n_msg - 1];
    const unsigned tiles = (cap + kBrickTile - 1) / kBrickTile;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((brick_dest_kernel<float, float4>), dim3(tiles), dim3(256), 0, s, (const float4 *)d_pos, cap,
                           make_args<float>(g), (const float *)d_bounds, w->key, w->sort_scratch, d_counts);
    else
        hipLaunchKernelGGL((brick_dest_kernel<double, double4>), dim3(tiles), dim3(256), 0, s, (const double4 *)d_pos, cap,
                           make_args<double>(g), (const double *)d_bounds, w->key, w->sort_scratch, d_counts);
    if (int rc = key_sort_from_hist<16, kBrickTile>(w->key, cap, w->sort_scratch, w->start1, w->order, s)) return rc;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((brick_pack_mig_kernel<float, float4>), dim3((mig_rows + 255) / 256), dim3(256), 0, s, (const float4 *)d_pos,
                           (const float4 *)d_vel, make_args<float>(g), w->order, w->start1, (float4 *)d_mig_send, mig_rows, d_counts);
    else
        hipLaunchKernelGGL((brick_pack_mig_kernel<double, double4>), dim3((mig_rows + 255) / 256), dim3(256), 0, s, (const double4 *)d_pos,
                           (const double4 *)d_vel, make_args<double>(g), w->order, w->start1, (double4 *)d_mig_send, mig_rows, d_counts);
    return check_launch("brick_pack_mig_kernel");
}

extern "C" int htfs_brick_migrate_merge(const htfs_brick *g, void *d_pos, void *d_vel, int dtype, const void *d_bounds,
                                        const htfs_brick_work *w, const void *d_mig_recv, unsigned *d_n_neigh, unsigned *d_counts,
                                        htf_stream stream) {
    if (int rc = check_geom(g, "htfs_brick_migrate_merge")) return rc;
    HTF_REQUIRE(d_pos && d_vel && d_bounds && w && w->key && w->order && w->sort_scratch && w->start1 && w->start2 && w->tmp_pos &&
                    w->tmp_vel && d_mig_recv && d_counts,
                "htfs_brick_migrate_merge: null pointer");
    HTF_REQUIRE(dtype == HTF_F32 || dtype == HTF_F64, "htfs_brick_migrate_merge: bad dtype %d", dtype);
    hipStream_t s = (hipStream_t)stream;
    const unsigned cap = g->cap_int + g->cap_bnd;
    const unsigned mig_rows = g->mig_off[g->n_msg - 1] + g->mig_cap[g->n_msg - 1];
    const unsigned cand = cap + mig_rows;
    // (w->key and w->order serve both sorts: brick_home_kernel has consumed the first order before the second sort writes its own)
    const unsigned tiles = (cand + kBrickTile - 1) / kBrickTile;
#define HTFS_HOME(T, V4, NK)                                                                                                       \
    hipLaunchKernelGGL((brick_home_kernel<T, V4, NK>), dim3(tiles), dim3(256), 0, s, (const V4 *)d_pos, (const V4 *)d_vel, make_args<T>(g), \
                       (const T *)d_bounds, w->order, w->start1, (const V4 *)d_mig_recv, (V4 *)w->tmp_pos, (V4 *)w->tmp_vel, w->key, cand, \
                       w->sort_scratch, d_counts)
    if (dtype == HTF_F32) {
        if (g->ndim == 1) HTFS_HOME(float, float4, 16); else HTFS_HOME(float, float4, 32);
    } else {
        if (g->ndim == 1) HTFS_HOME(double, double4, 16); else HTFS_HOME(double, double4, 32);
    }
#undef HTFS_HOME
    int rc = g->ndim == 1 ? key_sort_from_hist<16, kBrickTile>(w->key, cand, w->sort_scratch, w->start2, w->order, s)
                          : key_sort_from_hist<32, kBrickTile>(w->key, cand, w->sort_scratch, w->start2, w->order, s);
    if (rc) return rc;
    const unsigned rows = cap > 128u ? cap : 128u; // (the first threads also publish counts, message sizes and the slot table)
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((brick_place_kernel<float, float4>), dim3((rows + 255) / 256), dim3(256), 0, s, (float4 *)d_pos, (float4 *)d_vel,
                           make_args<float>(g), w->order, w->start2, (const float4 *)w->tmp_pos, (const float4 *)w->tmp_vel, d_n_neigh,
                           d_counts);
    else
        hipLaunchKernelGGL((brick_place_kernel<double, double4>), dim3((rows + 255) / 256), dim3(256), 0, s, (double4 *)d_pos,
                           (double4 *)d_vel, make_args<double>(g), w->order, w->start2, (const double4 *)w->tmp_pos,
                           (const double4 *)w->tmp_vel, d_n_neigh, d_counts);
    return check_launch("brick_place_kernel");
}

extern "C" int htfs_brick_pack_halo(const htfs_brick *g, const void *d_pos, int dtype, const unsigned *d_counts, void *d_send,
                                    void *d_ghost_direct, htf_stream stream) {
    if (int rc = check_geom(g, "htfs_brick_pack_halo")) return rc;
    HTF_REQUIRE(d_pos && d_counts && (d_send || d_ghost_direct), "htfs_brick_pack_halo: null pointer");
    HTF_REQUIRE(dtype == HTF_F32 || dtype == HTF_F64, "htfs_brick_pack_halo: bad dtype %d", dtype);
    const unsigned rows = g->ghost_off[g->n_msg - 1] + g->ghost_cap[g->n_msg - 1];
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((brick_pack_halo_kernel<float, float4>), dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)d_pos, make_args<float>(g), d_counts, (float4 *)d_send, (float4 *)d_ghost_direct, rows);
    else
        hipLaunchKernelGGL((brick_pack_halo_kernel<double, double4>), dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const double4 *)d_pos, make_args<double>(g), d_counts, (double4 *)d_send, (double4 *)d_ghost_direct, rows);
    return check_launch("brick_pack_halo_kernel");
}

extern "C" int htfs_brick_nve_halo(const htfs_brick *g, void *d_pos, void *d_vel, const void *d_force, int dtype, double dt,
                                   const htf_box *box, const unsigned *d_counts, void *d_send, void *d_ghost_direct, htf_stream stream) {
    if (int rc = check_geom(g, "htfs_brick_nve_halo")) return rc;
    HTF_REQUIRE(d_pos && d_vel && d_force && box && d_counts && (d_send || d_ghost_direct), "htfs_brick_nve_halo: null pointer");
    HTF_REQUIRE(dtype == HTF_F32 || dtype == HTF_F64, "htfs_brick_nve_halo: bad dtype %d", dtype);
    const unsigned cap = g->cap_int + g->cap_bnd;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((brick_nve_halo_kernel<float, float4>), dim3((cap + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float4 *)d_pos,
                           (float4 *)d_vel, (const float4 *)d_force, (float)dt, make_sbox<float>(box), make_args<float>(g), d_counts,
                           (float4 *)d_send, (float4 *)d_ghost_direct);
    else
        hipLaunchKernelGGL((brick_nve_halo_kernel<double, double4>), dim3((cap + 255) / 256), dim3(256), 0, (hipStream_t)stream, (double4 *)d_pos,
                           (double4 *)d_vel, (const double4 *)d_force, dt, make_sbox<double>(box), make_args<double>(g), d_counts,
                           (double4 *)d_send, (double4 *)d_ghost_direct);
    return check_launch("brick_nve_halo_kernel");
}

static int check_peer(const htfs_peer *p, const htfs_brick *g, const char *who) {
    HTF_REQUIRE(p && p->my_inbox && p->my_signal && p->state && p->spin_limit > 0, "%s: incomplete htfs_peer", who);
    for (int m = 0; m < g->n_msg; ++m) HTF_REQUIRE(p->inbox[m] && p->signal[m], "%s: message %d has no destination", who, m);
    return HTF_OK;
}

extern "C" int htfs_brick_row_slots(const htfs_brick *g, const unsigned *d_counts, unsigned *d_row_slots, htf_stream stream) {
    if (int rc = check_geom(g, "htfs_brick_row_slots")) return rc;
    HTF_REQUIRE(d_counts && d_row_slots, "htfs_brick_row_slots: null pointer");
    static_assert(HTFS_BRICK_MAX_MSG == 8, "two uint4 per row");
    if (g->cap_bnd == 0) return HTF_OK;
    hipLaunchKernelGGL((brick_row_slots_kernel<float>), dim3((g->cap_bnd + 255) / 256), dim3(256), 0, (hipStream_t)stream, make_args<float>(g),
                       d_counts, d_row_slots);
    return check_launch("brick_row_slots_kernel");
}

extern "C" int htfs_brick_pack_halo_peer(const htfs_brick *g, const void *d_pos, int dtype, const unsigned *d_counts, const htfs_peer *peer,
                                         htf_stream stream) {
    if (int rc = check_geom(g, "htfs_brick_pack_halo_peer")) return rc;
    if (int rc = check_peer(peer, g, "htfs_brick_pack_halo_peer")) return rc;
    HTF_REQUIRE(d_pos && d_counts && (dtype == HTF_F32 || dtype == HTF_F64), "htfs_brick_pack_halo_peer: null pointer or bad dtype");
    const unsigned rows = g->ghost_off[g->n_msg - 1] + g->ghost_cap[g->n_msg - 1];
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((brick_pack_halo_peer_kernel<float, float4>), dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)d_pos, make_args<float>(g), d_counts, make_peer(peer, g), rows);
    else
        hipLaunchKernelGGL((brick_pack_halo_peer_kernel<double, double4>), dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const double4 *)d_pos, make_args<double>(g), d_counts, make_peer(peer, g), rows);
    return check_launch("brick_pack_halo_peer_kernel");
}

extern "C" int htfs_brick_unpack_halo(const htfs_brick *g, void *d_pos, int dtype, const htfs_peer *peer, unsigned *d_counts,
                                      htf_stream stream) {
    if (int rc = check_geom(g, "htfs_brick_unpack_halo")) return rc;
    if (int rc = check_peer(peer, g, "htfs_brick_unpack_halo")) return rc;
    HTF_REQUIRE(d_pos && d_counts && (dtype == HTF_F32 || dtype == HTF_F64), "htfs_brick_unpack_halo: null pointer or bad dtype");
    const unsigned rows = g->ghost_off[g->n_msg - 1] + g->ghost_cap[g->n_msg - 1];
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((brick_unpack_halo_kernel<float, float4>), dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (float4 *)d_pos, make_args<float>(g), make_peer(peer, g), rows, d_counts);
    else
        hipLaunchKernelGGL((brick_unpack_halo_kernel<double, double4>), dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (double4 *)d_pos, make_args<double>(g), make_peer(peer, g), rows, d_counts);
    return check_launch("brick_unpack_halo_kernel");
}

extern "C" int htfs_brick_nve_halo_peer(const htfs_brick *g, void *d_pos, void *d_vel, const void *d_force, int dtype, double dt,
                                        const htf_box *box, const unsigned *d_counts, const htfs_peer *peer, htf_stream stream) {
    if (int rc = check_geom(g, "htfs_brick_nve_halo_peer")) return rc;
    if (int rc = check_peer(peer, g, "htfs_brick_nve_halo_peer")) return rc;
    HTF_REQUIRE(d_pos && d_vel && d_force && box && d_counts && (dtype == HTF_F32 || dtype == HTF_F64), "htfs_brick_nve_halo_peer: null pointer or bad dtype");
    const unsigned cap = g->cap_int + g->cap_bnd;
    if (dtype == HTF_F32)
        hipLaunchKernelGGL((brick_nve_halo_peer_kernel<float, float4>), dim3((cap + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float4 *)d_pos,
                           (float4 *)d_vel, (const float4 *)d_force, (float)dt, make_sbox<float>(box), make_args<float>(g), d_counts,
                           make_peer(peer, g));
    else
        hipLaunchKernelGGL((brick_nve_halo_peer_kernel<double, double4>), dim3((cap + 255) / 256), dim3(256), 0, (hipStream_t)stream, (double4 *)d_pos,
                           (double4 *)d_vel, (const double4 *)d_force, dt, make_sbox<double>(box), make_args<double>(g), d_counts,
                           make_peer(peer, g));
    return check_launch("brick_nve_halo_peer_kernel");
}
