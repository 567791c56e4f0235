// Forward ghost-POSITION halo over RCCL, from the C side: per step, each rank sends the
// positions of its particles within r_ghost of a slab face to that face's neighbor and receives its ghosts'
// new positions --
//
//     ncclGroupStart; ncclSend(left); ncclSend(right); ncclRecv(right); ncclRecv(left); ncclGroupEnd
//
// on a dedicated halo stream, ordered against the force kernels with two events: the exchange starts after
// everything already queued on the caller's stream (the integrator has written the positions), and the
// caller's stream waits for it only when htf_halo_exchange_end is called -- so htf_compute_forces_rows can
// evaluate the rows that have no ghost neighbor in between.  This is what HOOMD-blue's Communicator does for the
// reference under MPI (ghost update each step; the plugin reads ghosts through pos[N .. N + n_ghost),
// TensorflowCompute.cc:143-148); outside HOOMD, hoomd_tf_amd/domain.py drives it.
//
// librccl is bound at run time (dlopen, the copy torch has already loaded when there is one): a build box
// without a GPU, and a single-GPU run, never need it.
#include <dlfcn.h>

#include <new>

#include "htf_common.h"

namespace htf {

typedef struct ncclComm *ncclComm_t;
struct NcclUniqueId { char internal[128]; };
enum { kNcclSuccess = 0, kNcclChar = 0, kNcclFloat32 = 7, kNcclMax = 2 };

struct Rccl {
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(ncclComm_t, int *) = nullptr;      // (queries: optional, htf_halo_comm_info)
    int (*CommUserRank)(ncclComm_t, int *) = nullptr;
    int (*CommCuDevice)(ncclComm_t, int *) = nullptr;
    bool ok = false;
    Rccl() {
        void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD); // torch's copy, if torch is in the process
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(h, "ncclSend");
        Recv = (decltype(Recv))dlsym(h, "ncclRecv");
        AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        CommCount = (decltype(CommCount))dlsym(h, "ncclCommCount");
        CommUserRank = (decltype(CommUserRank))dlsym(h, "ncclCommUserRank");
        CommCuDevice = (decltype(CommCuDevice))dlsym(h, "ncclCommCuDevice");
        ok = GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv && AllReduce && GetErrorString;
    }
};

static Rccl &rccl() {
    static Rccl r;
    return r;
}

#define HTF_CHECK_NCCL(expr)                                                                                           \
    do {                                                                                                               \
        int _r = (expr);                                                                                               \
        if (_r != kNcclSuccess) {                                                                                      \
            htf::set_error("%s failed: %s", #expr, htf::rccl().GetErrorString(_r));                                   \
            return HTF_ERR_DEVICE;                                                                                     \
        }                                                                                                              \
    } while (0)

} // namespace htf

struct htf_halo {
    htf::ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr, done = nullptr; // positions ready on the caller's stream; exchange done on the halo stream
    bool pending = false;
};

extern "C" int htf_halo_available(void) { return htf::rccl().ok ? 1 : 0; }

extern "C" int htf_halo_unique_id(void *id128) {
    using namespace htf;
    HTF_REQUIRE(id128, "htf_halo_unique_id: null pointer");
    HTF_REQUIRE(rccl().ok, "htf_halo_unique_id: librccl could not be loaded");
    NcclUniqueId id;
    HTF_CHECK_NCCL(rccl().GetUniqueId(&id));
    std::memcpy(id128, id.internal, sizeof id.internal);
    return HTF_OK;
}

extern "C" int htf_halo_create(const void *id128, int rank, int world, htf_halo **out) {
    using namespace htf;
    HTF_REQUIRE(id128 && out, "htf_halo_create: null pointer");
    HTF_REQUIRE(world >= 1 && rank >= 0 && rank < world, "htf_halo_create: rank %d outside [0, %d)", rank, world);
    HTF_REQUIRE(rccl().ok, "htf_halo_create: librccl could not be loaded");
    htf_halo *h = new (std::nothrow) htf_halo();
    if (!h) {
        set_error("htf_halo_create: out of host memory");
        return HTF_ERR_NOMEM;
    }
    h->rank = rank;
    h->world = world;
    NcclUniqueId id;
    std::memcpy(id.internal, id128, sizeof id.internal);
    int rc = rccl().CommInitRank(&h->comm, world, id, rank);
    if (rc != kNcclSuccess) {
        set_error("ncclCommInitRank failed: %s", rccl().GetErrorString(rc));
        delete h;
        return HTF_ERR_DEVICE;
    }
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->done, hipEventDisableTiming);
    if (e != hipSuccess) {
        set_error("htf_halo_create: stream / event creation failed: %s", hipGetErrorString(e));
        htf_halo_destroy(h);
        return HTF_ERR_DEVICE;
    }
    *out = h;
    return HTF_OK;
}

extern "C" int htf_halo_comm_info(htf_halo *h, int *nranks, int *rank, int *device) {
    using namespace htf;
    HTF_REQUIRE(h && h->comm, "htf_halo_comm_info: null communicator");
    HTF_REQUIRE(rccl().CommCount && rccl().CommUserRank && rccl().CommCuDevice, "htf_halo_comm_info: this librccl has no communicator queries");
    int n = 0, r = 0, d = 0;
    HTF_CHECK_NCCL(rccl().CommCount(h->comm, &n));
    HTF_CHECK_NCCL(rccl().CommUserRank(h->comm, &r));
    HTF_CHECK_NCCL(rccl().CommCuDevice(h->comm, &d));
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (device) *device = d;
    return HTF_OK;
}

extern "C" void htf_halo_destroy(htf_halo *h) {
    if (!h) return;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->comm) (void)htf::rccl().CommDestroy(h->comm);
    if (h->ready) (void)hipEventDestroy(h->ready);
    if (h->done) (void)hipEventDestroy(h->done);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int htf_halo_exchange_begin(htf_halo *h, void *d_pos, int dtype, int left, int right, unsigned send_left_first,
                                       unsigned send_left_count, unsigned send_right_first, unsigned send_right_count,
                                       unsigned recv_left_first, unsigned recv_left_count, unsigned recv_right_first,
                                       unsigned recv_right_count, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(h && d_pos, "htf_halo_exchange_begin: null pointer");
    HTF_REQUIRE(dtype == HTF_F32 || dtype == HTF_F64, "htf_halo_exchange_begin: bad dtype %d", dtype);
    HTF_REQUIRE(left >= 0 && left < h->world && right >= 0 && right < h->world, "htf_halo_exchange_begin: neighbor ranks %d, %d outside [0, %d)", left, right, h->world);
    HTF_REQUIRE(!h->pending, "htf_halo_exchange_begin: the previous exchange has not been ended");
    const size_t s4 = dtype == HTF_F64 ? 32 : 16; // bytes of a Scalar4
    char *p = (char *)d_pos;
    // the positions must be final on the caller's stream before the halo stream reads them
    HTF_CHECK_HIP(hipEventRecord(h->ready, (hipStream_t)stream));
    HTF_CHECK_HIP(hipStreamWaitEvent(h->stream, h->ready, 0));
    // Every rank posts [send L>, send R>] and [recv L> (from the right neighbor), recv R> (from the left one)]:
    // with two ranks both neighbors are the same peer and the pairs still match in order of issue.
    HTF_CHECK_NCCL(rccl().GroupStart());
    int rc = kNcclSuccess;
    if (send_left_count) rc = rccl().Send(p + (size_t)send_left_first * s4, (size_t)send_left_count * s4, kNcclChar, left, h->comm, h->stream);
    if (rc == kNcclSuccess && send_right_count) rc = rccl().Send(p + (size_t)send_right_first * s4, (size_t)send_right_count * s4, kNcclChar, right, h->comm, h->stream);
    if (rc == kNcclSuccess && recv_right_count) rc = rccl().Recv(p + (size_t)recv_right_first * s4, (size_t)recv_right_count * s4, kNcclChar, right, h->comm, h->stream);
    if (rc == kNcclSuccess && recv_left_count) rc = rccl().Recv(p + (size_t)recv_left_first * s4, (size_t)recv_left_count * s4, kNcclChar, left, h->comm, h->stream);
    const int rc_end = rccl().GroupEnd();
    HTF_CHECK_NCCL(rc);
    HTF_CHECK_NCCL(rc_end);
    HTF_CHECK_HIP(hipEventRecord(h->done, h->stream));
    h->pending = true;
    return HTF_OK;
}

extern "C" int htf_halo_exchange_end(htf_halo *h, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(h, "htf_halo_exchange_end: null pointer");
    if (!h->pending) return HTF_OK;
    HTF_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, h->done, 0));
    h->pending = false;
    return HTF_OK;
}

// Any number of messages in one group (round 5: the brick decomposition's 2 or 8 halo messages, and the fixed-size migration
// messages of a rebuild -- hoomd_tf_amd/brick.py).  Sends are posted in the order given, then the receives in the order given:
// the caller orders them so that between any two ranks the k-th send meets the k-th receive.  async != 0: on the halo stream
// behind everything queued on `stream`, ended by htf_halo_exchange_end (work on `stream` in between overlaps the transfer);
// async == 0: on `stream` itself.  Both forms are plain stream work: a hipGraph capture of `stream` records them (the halo
// stream joins the capture through its two events).
extern "C" int htf_halo_exchange_n(htf_halo *h, int n_send, const void *const *send_ptrs, const size_t *send_bytes, const int *send_peers,
                                   int n_recv, void *const *recv_ptrs, const size_t *recv_bytes, const int *recv_peers,
                                   htf_stream stream, int async) {
    using namespace htf;
    HTF_REQUIRE(h && (n_send == 0 || (send_ptrs && send_bytes && send_peers)) && (n_recv == 0 || (recv_ptrs && recv_bytes && recv_peers)),
                "htf_halo_exchange_n: null pointer");
    HTF_REQUIRE(n_send >= 0 && n_recv >= 0 && n_send <= 64 && n_recv <= 64, "htf_halo_exchange_n: %d sends, %d receives", n_send, n_recv);
    HTF_REQUIRE(!h->pending, "htf_halo_exchange_n: the previous exchange has not been ended");
    for (int i = 0; i < n_send; ++i) HTF_REQUIRE(send_peers[i] >= 0 && send_peers[i] < h->world, "htf_halo_exchange_n: peer %d", send_peers[i]);
    for (int i = 0; i < n_recv; ++i) HTF_REQUIRE(recv_peers[i] >= 0 && recv_peers[i] < h->world, "htf_halo_exchange_n: peer %d", recv_peers[i]);
    hipStream_t on = (hipStream_t)stream;
    if (async) {
        HTF_CHECK_HIP(hipEventRecord(h->ready, (hipStream_t)stream));
        HTF_CHECK_HIP(hipStreamWaitEvent(h->stream, h->ready, 0));
        on = h->stream;
    }
    HTF_CHECK_NCCL(rccl().GroupStart());
    int rc = kNcclSuccess;
    for (int i = 0; i < n_send && rc == kNcclSuccess; ++i)
        if (send_bytes[i]) rc = rccl().Send(send_ptrs[i], send_bytes[i], kNcclChar, send_peers[i], h->comm, on);
    for (int i = 0; i < n_recv && rc == kNcclSuccess; ++i)
        if (recv_bytes[i]) rc = rccl().Recv(recv_ptrs[i], recv_bytes[i], kNcclChar, recv_peers[i], h->comm, on);
    const int rc_end = rccl().GroupEnd();
    HTF_CHECK_NCCL(rc);
    HTF_CHECK_NCCL(rc_end);
    if (async) {
        HTF_CHECK_HIP(hipEventRecord(h->done, h->stream));
        h->pending = true;
    }
    return HTF_OK;
}

// *d_value <- max over ranks, in place, on `stream` (the distance check of a decomposed system: every rank takes the same
// rebuild decision)
extern "C" int htf_halo_allreduce_max_f32(htf_halo *h, float *d_value, unsigned n, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(h && d_value && n > 0, "htf_halo_allreduce_max_f32: null pointer");
    HTF_CHECK_NCCL(rccl().AllReduce(d_value, d_value, n, kNcclFloat32, kNcclMax, h->comm, (hipStream_t)stream));
    return HTF_OK;
}
