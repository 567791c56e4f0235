// Device-side message passing between ranks WITHOUT a communication library (include/htf_standin.h htfs_mailbox): what transport
// "peer" of brick.py needs beside the per-step halo (brick.hip *_peer_kernel) so that a whole decomposed run -- the migration
// messages of a re-plan and the all-reduced distance check included -- is ordinary kernel work on one stream, capturable into a
// hipGraph: the sender stores its message straight into the receiver's mailbox (its own memory, or a mapping of another process's /
// another device's: htfs_shared_alloc + htfs_ipc_export / htfs_ipc_import) and publishes a sequence number behind a system-scope
// release; the receiver polls its own signal words (bounded: a flag, never a hang) and copies what has arrived.
//
// What HOOMD's Communicator does with MPI for the reference (migrateParticles; the reference itself only reads the result,
// htf/TensorflowCompute.cc:143-148); outside the drop-in boundary.
#include "htf_common.h"
#include "htf_standin.h"

namespace htf {

struct MailboxArgs {
    void *remote[HTFS_MBOX_MAX_MSG];
    unsigned *remote_signal[HTFS_MBOX_MAX_MSG];
    void *mine;
    unsigned *my_signal;
    unsigned *state;
    unsigned spin_limit, half_units;
    unsigned local_off[HTFS_MBOX_MAX_MSG], units[HTFS_MBOX_MAX_MSG], box_off[HTFS_MBOX_MAX_MSG];
    int n_msg, row_units;
};

static int make_mailbox(const htfs_mailbox *mb, int n_msg, const unsigned *local_off, const unsigned *units, const unsigned *box_off,
                        int row_units, MailboxArgs &a, const char *who) {
    HTF_REQUIRE(mb && mb->mine && mb->my_signal && mb->state && mb->spin_limit > 0, "%s: incomplete htfs_mailbox", who);
    HTF_REQUIRE(n_msg >= 1 && n_msg <= HTFS_MBOX_MAX_MSG && local_off && units && box_off && row_units >= 0, "%s: bad message table", who);
    for (int m = 0; m < n_msg; ++m) {
        HTF_REQUIRE(mb->remote[m] && mb->remote_signal[m], "%s: message %d has no destination", who, m);
        HTF_REQUIRE((unsigned long long)box_off[m] + units[m] <= mb->half_units, "%s: message %d does not fit the mailbox", who, m);
        a.remote[m] = mb->remote[m];
        a.remote_signal[m] = mb->remote_signal[m];
        a.local_off[m] = local_off[m];
        a.units[m] = units[m];
        a.box_off[m] = box_off[m];
    }
    a.mine = mb->mine;
    a.my_signal = mb->my_signal;
    a.state = mb->state;
    a.spin_limit = mb->spin_limit;
    a.half_units = mb->half_units;
    a.n_msg = n_msg;
    a.row_units = row_units;
    return HTF_OK;
}

__device__ __forceinline__ void store_unit(uint4 *p, const uint4 &v) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z);
    __builtin_nontemporal_store(v.w, &p->w);
}
__device__ __forceinline__ uint4 load_unit(const uint4 *p) {
    return make_uint4(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y), __builtin_nontemporal_load(&p->z),
                      __builtin_nontemporal_load(&p->w));
}

// units of message m that travel: all of them, or (row_units > 0) row 0 + as many rows as its first word says
__device__ __forceinline__ unsigned live_units(const MailboxArgs &a, int m, const uint4 *first) {
    if (a.row_units <= 0) return a.units[m];
    const unsigned long long want = (1ull + (unsigned long long)first->x) * (unsigned)a.row_units;
    return want < a.units[m] ? (unsigned)want : a.units[m];
}

// blockIdx.y = message; the launch's last workgroup publishes the sequence number of every message and advances the counter
__global__ __launch_bounds__(256) void mailbox_push_kernel(const uint4 *__restrict__ send, MailboxArgs a) {
    const unsigned seq = a.state[0] + 1u; // (read by every workgroup before the last one advances it)
    const int m = (int)blockIdx.y;
    const uint4 *src = send + a.local_off[m];
    const unsigned n = live_units(a, m, src);
    uint4 *dst = reinterpret_cast<uint4 *>(a.remote[m]) + (size_t)(seq & 1u) * a.half_units + a.box_off[m];
    for (unsigned u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) store_unit(dst + u, src[u]);
    __syncthreads();
    if (threadIdx.x != 0) return;
    __threadfence_system(); // (one per workgroup, behind its barrier: the units are out before the number is)
    const unsigned done = __hip_atomic_fetch_add(&a.state[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done + 1u != gridDim.x * gridDim.y) return;
    a.state[1] = 0u;
    for (int k = 0; k < a.n_msg; ++k) __hip_atomic_store(a.remote_signal[k], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&a.state[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// blockIdx.y = message (by ITS slot in my mailbox): wait for its number, copy it out
__global__ __launch_bounds__(256) void mailbox_pull_kernel(uint4 *__restrict__ recv, MailboxArgs a, unsigned *__restrict__ flags, unsigned flag_bit) {
    const unsigned seq = a.state[0]; // (advanced by this exchange's push, earlier in the stream)
    const int j = (int)blockIdx.y;
    __shared__ int s_late;
    if (threadIdx.x == 0) {
        unsigned n = 0;
        int late = 0;
        while ((int)(__hip_atomic_load(a.my_signal + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            if (++n > a.spin_limit) {
                late = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        __threadfence_system(); // (acquire side)
        if (late) {
            if (flags != nullptr) atomicOr(flags, flag_bit);
            atomicAdd(&a.state[2], 1u);
        }
        s_late = late;
    }
    __syncthreads();
    uint4 *dst = recv + a.local_off[j];
    if (s_late) { // nothing arrived: an EMPTY message (count 0), so that whatever reads it moves nobody
        if (blockIdx.x == 0 && threadIdx.x == 0) dst[0] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const uint4 *src = reinterpret_cast<const uint4 *>(a.mine) + (size_t)(seq & 1u) * a.half_units + a.box_off[j];
    uint4 first = load_unit(src);
    const unsigned n = live_units(a, j, &first);
    for (unsigned u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) dst[u] = load_unit(src + u);
}

// value <- max over the ranks.  One workgroup: lane r stores {sequence, my value} as ONE 8-byte word into slot `rank` of rank r's
// table (no fence needed: the word is the message), then polls slot r of its own table for this sequence number.
struct ReduceArgs {
    unsigned long long *remote[HTFS_MBOX_MAX_RANKS];
    unsigned long long *mine;
    unsigned *state;
    unsigned spin_limit;
    int world, rank;
};

__global__ __launch_bounds__(64) void mailbox_allreduce_max_kernel(float *__restrict__ value, ReduceArgs a, unsigned *__restrict__ flags,
                                                                   unsigned flag_bit) {
    const unsigned seq = a.state[0] + 1u;
    const int r = (int)threadIdx.x;
    const float v = value[0];
    float got = 0.f; // (displacements: non-negative)
    int late = 0;
    if (r < a.world) {
        const unsigned long long word = ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(v);
        __hip_atomic_store(a.remote[r] + 2 * a.rank + (seq & 1u), word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        unsigned n = 0;
        unsigned long long w;
        while ((int)((unsigned)((w = __hip_atomic_load(a.mine + 2 * r + (seq & 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) >> 32) - seq) < 0) {
            if (++n > a.spin_limit) {
                late = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        got = late ? 0.f : __uint_as_float((unsigned)w);
    }
    for (int d = 32; d >= 1; d >>= 1) {
        got = fmaxf(got, __shfl_xor(got, d));
        late |= __shfl_xor(late, d);
    }
    if (r == 0) {
        // a rank that did not answer: report the largest float, so that the caller rebuilds rather than trusts a stale list
        value[0] = late ? 3.0e38f : got;
        if (late) {
            if (flags != nullptr) atomicOr(flags, flag_bit);
            atomicAdd(&a.state[2], 1u);
        }
        a.state[0] = seq;
    }
}

} // namespace htf

// ---- memory another process / device can map
extern "C" int htfs_shared_alloc(size_t bytes, int finegrained, void **out) {
    using namespace htf;
    HTF_REQUIRE(out && bytes > 0, "htfs_shared_alloc: null pointer or zero size");
    void *p = nullptr;
    hipError_t e = finegrained ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) : hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        set_error("htfs_shared_alloc: %s of %zu bytes failed: %s", finegrained ? "hipExtMallocWithFlags(hipDeviceMallocFinegrained)" : "hipMalloc",
                  bytes, hipGetErrorString(e));
        (void)hipGetLastError();
        return HTF_ERR_NOMEM;
    }
    e = hipMemset(p, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        set_error("htfs_shared_alloc: hipMemset failed: %s", hipGetErrorString(e));
        (void)hipFree(p);
        return HTF_ERR_DEVICE;
    }
    *out = p;
    return HTF_OK;
}

extern "C" int htfs_shared_free(void *p) {
    using namespace htf;
    if (!p) return HTF_OK;
    HTF_CHECK_HIP(hipFree(p));
    return HTF_OK;
}

extern "C" int htfs_ipc_export(const void *p, void *handle64) {
    using namespace htf;
    static_assert(sizeof(hipIpcMemHandle_t) <= HTFS_IPC_HANDLE_BYTES, "htf_standin.h HTFS_IPC_HANDLE_BYTES is too small");
    HTF_REQUIRE(p && handle64, "htfs_ipc_export: null pointer");
    hipIpcMemHandle_t h;
    HTF_CHECK_HIP(hipIpcGetMemHandle(&h, const_cast<void *>(p)));
    std::memset(handle64, 0, HTFS_IPC_HANDLE_BYTES);
    std::memcpy(handle64, &h, sizeof h);
    return HTF_OK;
}

extern "C" int htfs_ipc_import(const void *handle64, void **out) {
    using namespace htf;
    HTF_REQUIRE(handle64 && out, "htfs_ipc_import: null pointer");
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, sizeof h);
    void *p = nullptr;
    HTF_CHECK_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *out = p;
    return HTF_OK;
}

extern "C" int htfs_ipc_close(void *p) {
    using namespace htf;
    if (!p) return HTF_OK;
    HTF_CHECK_HIP(hipIpcCloseMemHandle(p));
    return HTF_OK;
}

// ---- the exchange
extern "C" int htfs_mailbox_push(const htfs_mailbox *mb, int n_msg, const void *d_send, const unsigned *send_off, const unsigned *units,
                                 const unsigned *box_off, int row_units, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_send, "htfs_mailbox_push: null pointer");
    MailboxArgs a;
    if (int rc = make_mailbox(mb, n_msg, send_off, units, box_off, row_units, a, "htfs_mailbox_push")) return rc;
    unsigned most = 1;
    for (int m = 0; m < n_msg; ++m) most = units[m] > most ? units[m] : most;
    unsigned gx = (most + 255u) / 256u;
    gx = gx > 16u ? 16u : gx;
    hipLaunchKernelGGL(mailbox_push_kernel, dim3(gx, (unsigned)n_msg), dim3(256), 0, (hipStream_t)stream, (const uint4 *)d_send, a);
    return check_launch("mailbox_push_kernel");
}

extern "C" int htfs_mailbox_pull(const htfs_mailbox *mb, int n_msg, void *d_recv, const unsigned *recv_off, const unsigned *units,
                                 const unsigned *box_off, int row_units, unsigned *d_flags, unsigned flag_bit, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(d_recv, "htfs_mailbox_pull: null pointer");
    MailboxArgs a;
    // (the pull reads only its own side: the remote table is not needed, but an incomplete one is a caller's bug all the same)
    if (int rc = make_mailbox(mb, n_msg, recv_off, units, box_off, row_units, a, "htfs_mailbox_pull")) return rc;
    unsigned most = 1;
    for (int m = 0; m < n_msg; ++m) most = units[m] > most ? units[m] : most;
    unsigned gx = (most + 255u) / 256u;
    gx = gx > 16u ? 16u : gx;
    hipLaunchKernelGGL(mailbox_pull_kernel, dim3(gx, (unsigned)n_msg), dim3(256), 0, (hipStream_t)stream, (uint4 *)d_recv, a, d_flags, flag_bit);
    return check_launch("mailbox_pull_kernel");
}

extern "C" int htfs_mailbox_allreduce_max_f32(const htfs_reduce_box *rb, float *d_value, unsigned *d_flags, unsigned flag_bit, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(rb && d_value && rb->mine && rb->state && rb->spin_limit > 0, "htfs_mailbox_allreduce_max_f32: incomplete htfs_reduce_box");
    HTF_REQUIRE(rb->world >= 1 && rb->world <= HTFS_MBOX_MAX_RANKS && rb->rank >= 0 && rb->rank < rb->world,
                "htfs_mailbox_allreduce_max_f32: rank %d of %d (at most %d ranks)", rb->rank, rb->world, HTFS_MBOX_MAX_RANKS);
    ReduceArgs a;
    for (int r = 0; r < rb->world; ++r) {
        HTF_REQUIRE(rb->remote[r], "htfs_mailbox_allreduce_max_f32: rank %d's table is not mapped", r);
        a.remote[r] = (unsigned long long *)rb->remote[r];
    }
    a.mine = (unsigned long long *)rb->mine;
    a.state = rb->state;
    a.spin_limit = rb->spin_limit;
    a.world = rb->world;
    a.rank = rb->rank;
    hipLaunchKernelGGL(mailbox_allreduce_max_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_value, a, d_flags, flag_bit);
    return check_launch("mailbox_allreduce_max_kernel");
}
