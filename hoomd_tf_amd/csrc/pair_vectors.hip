// Pair-vector build: HOOMD ragged index neighbor list -> dense zero-padded
// [B, NN, 4] pair vectors (dx, dy, dz, type_j).
//
// Replaces htf_gpu_reshape_nlist_kernel (TensorflowCompute.cu:80-151, thread per
// particle, uncoalesced stride-NN stores) and follows the CPU prepareNeighbors
// semantics (TensorflowCompute.cc:303-374): zero fill, keep unless rsq > rcut^2,
// slot = (slot + 1) % NN so that on overflow the LAST writer wins.
//
// MI355X mapping: one wave64 per particle.  Lanes take consecutive neighbor indices
// (coalesced 256-B index reads), gather pos[k] (2 MB table, L2/Infinity-Cache
// resident), apply the minimum image and compact the survivors with a wave ballot +
// mbcnt prefix, so kept neighbors land in consecutive float4 slots (coalesced
// stores).  The zero tail is written by the same wave: no separate memset pass.
// Measured alternatives (tools/kernel_ab.py, 131072 x 128, r_buff 0.4): this layout 85 us;
// LDS-staged metadata + next-particle index prefetch 130 us; four 16-lane groups per wave
// 119 us.  The kernel is bound by the 16-B position gathers (one L1 line lookup per lane:
// 18.2 M lanes / 256 CUs ~ 30 us at one lane per clock) plus the 268 MB row write, not by
// the dependent-load latency the restructurings targeted.
// Ablation (same harness): 84 us full, 76 us with the random gather replaced by a local
// one, 35 us with the row stores removed.  A wave keeps its slot until its stores have
// completed (~3 us under load, 60 % of its lifetime), and gfx950 retires loads and stores
// on ONE in-order counter, so a multi-particle-per-wave pipeline only overlaps them if the
// number of stores between a load and its use is a compile-time constant (otherwise hipcc
// falls back to vmcnt(0) at the loop head -- measured 130 us).  A loader/storer split (waves
// 0-3 only load + compact into an LDS row ring, waves 4-7 only store, raw s_barrier hand-over)
// was also built and is bit-exact, but ran 101 us: hipcc drains the loader's prefetched loads
// (vmcnt(0)) in front of every stage barrier, so the prefetch never spans a stage.  Getting
// Finally, 2 and 4 particles per wave with ALL their loads issued up front (same dependent
// chain, 2-4x the rows behind every wave slot) ran 81.4 / 80.8 us against 82.0: the kernel is
// not wave-slot- or latency-bound either, and writing 190 MB instead of 268 MB (delta
// zero-fill, PMC-verified) does not move it.  Assembling the row in LDS and leaving with aligned
// full-width stores (2 x 1 KiB per row instead of ~4 partial ones): 82.1 us, no change either.
// For scale (tools/bw_probe.py on the same box): memset of the 268 MB tensor 37 us (6.9-7.3
// TB/s), 268 MB copy 98 us (5.5 TB/s r+w).  The load phase (35 us) and the store phase (~40 us)
// of this kernel simply add up: a CU's vector-memory pipeline issues in order, so stores that
// wait for write-buffer space hold back the loads of the other waves on that CU.
//
// Compiled with -ffp-contract=off: the arithmetic is then op-for-op the oracle's
// (oracle/htf_oracle.py:min_image / prepare_neighbors), so pair vectors are
// bit-exact, not merely within tolerance.
#include <cstdlib>

#include "htf_common.h"
#include "box_math.h"

namespace htf {

constexpr int kChunk = 3; // index loads hoisted per lane: n_neigh <= 192 in one trip (C3 with r_buff 0.4: ~139)

template <typename PT, typename DT, bool REPLAY>
__device__ __forceinline__ unsigned sweep(typename Vec4<DT>::type *__restrict__ row,
                                          const typename Vec4<PT>::type *__restrict__ pos,
                                          const unsigned *__restrict__ nl, unsigned nn,
                                          const typename Vec4<PT>::type pi, const BoxT<PT> &box,
                                          PT rmaxsq, unsigned NN, unsigned lane, unsigned lo) {
    using PV = typename Vec4<PT>::type;
    using DV = typename Vec4<DT>::type;
    unsigned Q = 0;
    for (unsigned base = 0; base < nn; base += 64 * kChunk) {
        unsigned k[kChunk];
        PV pk[kChunk];
#pragma unroll
        for (int t = 0; t < kChunk; ++t) {
            unsigned j = base + t * 64 + lane;
            k[t] = nl[j < nn ? j : nn - 1];
        }
#pragma unroll
        for (int t = 0; t < kChunk; ++t) pk[t] = pos[k[t]];
#pragma unroll
        for (int t = 0; t < kChunk; ++t) {
            if (base + t * 64 >= nn) break; // wave-uniform
            unsigned j = base + t * 64 + lane;
            PT dx = pk[t].x - pi.x, dy = pk[t].y - pi.y, dz = pk[t].z - pi.z;
            min_image<PT>(dx, dy, dz, box);
            PT rsq = dx * dx + dy * dy + dz * dz;
            bool keep = (j < nn) && !(rsq > rmaxsq);
            unsigned long long m = __ballot(keep);
            unsigned q = Q + ballot_rank(m);
            Q += __popcll(m);
            DV out;
            out.x = (DT)dx; out.y = (DT)dy; out.z = (DT)dz;
            out.w = (DT)scalar_as_int(pk[t].w);
            if constexpr (!REPLAY) {
                if (keep && q < NN) store_stream(&row[q], out);
            } else {
                if (keep && q >= lo) store_stream(&row[q % NN], out);
            }
        }
    }
    return Q;
}

template <typename PT, typename DT>
__device__ __forceinline__ void build_row(
    const unsigned w, const unsigned lane, typename Vec4<DT>::type *__restrict__ dest,
    const typename Vec4<PT>::type *__restrict__ pos, unsigned N, unsigned NN, unsigned offset, const BoxT<PT> &box,
    const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist, const unsigned *__restrict__ head_list,
    PT rmaxsq, unsigned *__restrict__ max_count, float4 *__restrict__ positions_out, unsigned *__restrict__ counts_io) {
    using DV = typename Vec4<DT>::type;
    const unsigned idx = w + offset;
    if (idx >= N) return;
    const unsigned nn = n_neigh[idx];
    const unsigned *nl = nlist + head_list[idx];
    const auto pi = pos[idx];
    DV *row = dest + (size_t)w * NN;
    // m_positions_comm.receiveArray(..., unstuff4=true) (TensorflowCompute.cc:172) for free:
    // this wave already holds pos[idx]
    if (positions_out != nullptr && lane == 0)
        positions_out[w] = make_float4((float)pi.x, (float)pi.y, (float)pi.z, (float)scalar_as_int(pi.w));

    unsigned Q = nn ? sweep<PT, DT, false>(row, pos, nl, nn, pi, box, rmaxsq, NN, lane, 0u) : 0u;

    DV zero;
    zero.x = zero.y = zero.z = zero.w = (DT)0;
    const unsigned filled = Q < NN ? Q : NN;
    // Context-owned scratch (counts_io != null): the row held counts_io[w] live slots after the
    // previous call and zeros behind them, so only slots [filled, previous) need re-zeroing --
    // the padding (26 % of the tensor at C3) is not rewritten every step.
    const unsigned zero_end = counts_io != nullptr ? counts_io[w] : NN;
    for (unsigned s = filled + lane; s < zero_end; s += 64) store_stream(&row[s], zero);
    if (counts_io != nullptr && lane == 0) counts_io[w] = filled;

    if (Q > NN) {
        // Overflow (an error condition upstream, caught by check_nlist): reproduce the
        // reference's wrap exactly.  Entry q lands in slot q % NN and survives iff no
        // later entry maps to the same slot, i.e. q >= Q - NN.  Slots already hold
        // entries q < NN; replay, in order, the survivors with q >= max(NN, Q - NN)
        // after the first sweep's stores have retired.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned lo = Q - NN > NN ? Q - NN : NN;
        sweep<PT, DT, true>(row, pos, nl, nn, pi, box, rmaxsq, NN, lane, lo);
    }
    if (max_count != nullptr && lane == 0 && Q > *(volatile unsigned *)max_count) atomicMax(max_count, Q);
}

// Two rows per wave with ALL their index loads, then all their gathers, issued before any store (the
// structure of fused_forces_rows2_kernel, fused_eval.hip, which builds AND evaluates in 63 us).  Fast
// path for the common case (both rows have 1..192 list entries and do not overflow NN); anything else
// is redone by build_row.  R = 1 is the plain wave-per-row kernel.
// Measured at C3 with streaming stores (run-to-run spread ~10 %): one row per wave 67-75 us; 2 / 4 / 8
// rows per wave taken one after the other 74 / 74 / 77 us; loads up front, 2 / 4 / 8 rows: 64-71 /
// 60-68 / 72 us.
// (Round 3, measured and removed: the merged-tail form of fused_rows_group_tails as a pure builder -- two straight-line trips per
//  row and ONE shared trip for the tails of a wave's rows.  Standalone with the full zero fill it won 5 % at two rows per wave
//  (68.4-69.2 -> 64.6-65.4 us at C3, 344 MB of true traffic = 54.7 us at the sustainable rate; three / four rows 75.5 / 76.8),
//  but through the context, where counts_io bounds the zero fill, it LOST 5 % (build + evaluator 112.0-112.2 against
//  106.5-107.1 us on an equilibrated liquid) and in the pair-MLP step 85.7-86.7 against 81.5-81.6.  Tensor digests identical.)
template <typename PT, typename DT, int R>
__global__ __launch_bounds__(256) void build_pair_vectors_kernel(
    typename Vec4<DT>::type *__restrict__ dest, const typename Vec4<PT>::type *__restrict__ pos,
    unsigned N, unsigned NN, unsigned offset, unsigned batch, BoxT<PT> box,
    const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist,
    const unsigned *__restrict__ head_list, PT rmaxsq, unsigned *__restrict__ max_count,
    float4 *__restrict__ positions_out, unsigned *__restrict__ counts_io) {
    using PV = typename Vec4<PT>::type;
    using DV = typename Vec4<DT>::type;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wv = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const unsigned w0 = wv * R;
    if (w0 >= batch) return;
    unsigned nn[R];
    bool fast = R > 1 && w0 + R <= batch && w0 + R - 1 + offset < N;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        nn[r] = fast ? n_neigh[w0 + r + offset] : 0u;
        fast = fast && nn[r] != 0 && nn[r] <= 64 * kChunk;
    }
    if (!fast) {
#pragma unroll 1
        for (unsigned r = 0; r < (unsigned)R && w0 + r < batch; ++r)
            build_row<PT, DT>(w0 + r, lane, dest, pos, N, NN, offset, box, n_neigh, nlist, head_list, rmaxsq, max_count,
                              positions_out, counts_io);
        return;
    }
    const bool simple_box = box.ortho && box.periodic[0] && box.periodic[1] && box.periodic[2];
    PV pi[R];
    unsigned k[R][kChunk];
    PV pk[R][kChunk];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned *nl = nlist + head_list[w0 + r + offset];
        pi[r] = pos[w0 + r + offset];
#pragma unroll
        for (int t = 0; t < kChunk; ++t) {
            const unsigned j = t * 64 + lane;
            k[r][t] = nl[min(j, nn[r] - 1u)];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < kChunk; ++t) pk[r][t] = load_neighbor(pos, k[r][t]); // (fp64: 28 of the 32 bytes; build + evaluator 122.1 -> 120.4 us)
    unsigned redo = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned w = w0 + r;
        if (positions_out != nullptr && lane == 0)
            positions_out[w] = make_float4((float)pi[r].x, (float)pi[r].y, (float)pi[r].z, (float)scalar_as_int(pi[r].w));
        DV *row = dest + (size_t)w * NN;
        unsigned Q = 0;
#pragma unroll
        for (int t = 0; t < kChunk; ++t) {
            if ((unsigned)t * 64 >= nn[r]) break; // wave-uniform
            // (as in fused_eval.hip: the common box on a 12-instruction minimum image behind a wave-uniform branch, the
            //  live-entry mask from scalar arithmetic, the lane's predicate read back from the scalar mask)
            const unsigned left = nn[r] - (unsigned)t * 64;
            const unsigned long long valid = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
            PT dx, dy, dz, rsq;
            if (simple_box) {
                asm volatile("" ::: "memory");
                rsq = pair_vector_simple<PT>(pk[r][t], pi[r], box, dx, dy, dz);
            } else {
                rsq = pair_vector<PT>(pk[r][t], pi[r], box, dx, dy, dz);
            }
            unsigned long long m = ballot64(!(rsq > rmaxsq)) & valid;
            const unsigned q = Q + ballot_rank(m);
            Q += __popcll(m);
            if (Q > NN) { // (wave-uniform) about to overflow: slots bounded lane by lane; the row is redone below
                asm volatile("" ::: "memory");
                m &= ballot64(q < NN);
            }
            DV out;
            out.x = (DT)dx; out.y = (DT)dy; out.z = (DT)dz;
            out.w = (DT)scalar_as_int(pk[r][t].w);
            if (inverse_ballot64(m)) store_stream(&row[q], out);
        }
        if (Q > NN) { // overflow (an error upstream): build_row redoes the whole row, slot wrap included
            redo |= 1u << r;
            continue;
        }
        DV zero;
        zero.x = zero.y = zero.z = zero.w = (DT)0;
        const unsigned zero_end = counts_io != nullptr ? counts_io[w] : NN;
        for (unsigned sl = Q + lane; sl < zero_end; sl += 64) store_stream(&row[sl], zero);
        if (counts_io != nullptr && lane == 0) counts_io[w] = Q;
        if (max_count != nullptr && lane == 0 && Q > *(volatile unsigned *)max_count) atomicMax(max_count, Q);
    }
#pragma unroll 1
    for (unsigned r = 0; r < (unsigned)R; ++r)
        if ((redo >> r) & 1u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            build_row<PT, DT>(w0 + r, lane, dest, pos, N, NN, offset, box, n_neigh, nlist, head_list, rmaxsq, max_count,
                              positions_out, counts_io);
        }
}

template <typename PT, typename DT>
static int launch_build(void *dest, const void *pos, unsigned N, unsigned NN, unsigned offset,
                        unsigned batch, const htf_box *hb, const unsigned *n_neigh,
                        const unsigned *nlist, const unsigned *head_list, double rmax,
                        unsigned *max_count, float4 *positions_out, unsigned *counts_io, hipStream_t stream) {
    BoxT<PT> b = make_boxt<PT>(hb);
    PT rc = (PT)rmax;
    PT rmaxsq = rc * rc;
    // rows per wave: 4 for fp32 positions, 2 for fp64 (4 rows spill SGPRs).  The other geometries lost their A/Bs (header of this
    // file) and exist in variants builds only (make CXXFLAGS_EXTRA=-DHTF_AB_VARIANTS: HTF_BUILD_ROWS = 1 | 2 | 4 | 8).
#define HTF_BUILD_LAUNCH(RR)                                                                                           \
    hipLaunchKernelGGL((build_pair_vectors_kernel<PT, DT, RR>), dim3((batch + 4 * RR - 1) / (4 * RR)), dim3(256), 0, stream, \
                       (typename Vec4<DT>::type *)dest, (const typename Vec4<PT>::type *)pos, N, NN, offset, batch, b, \
                       n_neigh, nlist, head_list, rmaxsq, max_count, positions_out, counts_io)
#ifdef HTF_AB_VARIANTS
    static const char *rows_env = getenv("HTF_BUILD_ROWS");
    const int rows = rows_env ? atoi(rows_env) : (sizeof(PT) == 4 ? 4 : 2);
    if (rows == 1) HTF_BUILD_LAUNCH(1);
    else if (rows == 4) HTF_BUILD_LAUNCH(4);
    else if (rows == 8) HTF_BUILD_LAUNCH(8);
    else HTF_BUILD_LAUNCH(2);
#else
    if constexpr (sizeof(PT) == 4) HTF_BUILD_LAUNCH(4); else HTF_BUILD_LAUNCH(2);
#endif
#undef HTF_BUILD_LAUNCH
    return check_launch("build_pair_vectors_kernel");
}

} // namespace htf

namespace htf {
int build_pair_vectors_impl(void *dest, int dest_dtype, const void *d_pos, int pos_dtype, unsigned N, unsigned NN,
                            unsigned offset, unsigned batch_size, const htf_box *box, const unsigned *d_n_neigh,
                            const unsigned *d_nlist, const unsigned *d_head_list, double rmax,
                            unsigned *d_max_count, float4 *positions_out, unsigned *counts_io, hipStream_t s) {
    HTF_REQUIRE(dest && d_pos && d_n_neigh && d_nlist && d_head_list && box, "htf_build_pair_vectors: null pointer");
    HTF_REQUIRE(NN > 0, "htf_build_pair_vectors: NN must be > 0");
    HTF_REQUIRE(offset <= N && batch_size <= N - offset, "htf_build_pair_vectors: batch [%u, %u) exceeds N=%u", offset, offset + batch_size, N);
    HTF_REQUIRE(rmax > 0, "htf_build_pair_vectors: rmax must be > 0");
    for (int d = 0; d < 3; ++d)
        HTF_REQUIRE(box->hi[d] > box->lo[d], "htf_build_pair_vectors: empty box along %d", d);
    if (batch_size == 0) return HTF_OK;
    if (pos_dtype == HTF_F32 && dest_dtype == HTF_F32)
        return launch_build<float, float>(dest, d_pos, N, NN, offset, batch_size, box, d_n_neigh, d_nlist, d_head_list, rmax, d_max_count, positions_out, counts_io, s);
    if (pos_dtype == HTF_F64 && dest_dtype == HTF_F32)
        return launch_build<double, float>(dest, d_pos, N, NN, offset, batch_size, box, d_n_neigh, d_nlist, d_head_list, rmax, d_max_count, positions_out, counts_io, s);
    if (pos_dtype == HTF_F64 && dest_dtype == HTF_F64)
        return launch_build<double, double>(dest, d_pos, N, NN, offset, batch_size, box, d_n_neigh, d_nlist, d_head_list, rmax, d_max_count, positions_out, counts_io, s);
    if (pos_dtype == HTF_F32 && dest_dtype == HTF_F64)
        return launch_build<float, double>(dest, d_pos, N, NN, offset, batch_size, box, d_n_neigh, d_nlist, d_head_list, rmax, d_max_count, positions_out, counts_io, s);
    set_error("htf_build_pair_vectors: bad dtype (%d, %d)", pos_dtype, dest_dtype);
    return HTF_ERR_INVALID;
}
} // namespace htf

extern "C" int htf_build_pair_vectors(void *dest, int dest_dtype, const void *d_pos, int pos_dtype,
                                      unsigned N, unsigned NN, unsigned offset, unsigned batch_size,
                                      unsigned n_ghost, const htf_box *box, const unsigned *d_n_neigh,
                                      const unsigned *d_nlist, const unsigned *d_head_list, double rmax,
                                      unsigned *d_max_count, htf_stream stream) {
    (void)n_ghost; // ghosts are addressed through the index list (k >= N); nothing to size
    return htf::build_pair_vectors_impl(dest, dest_dtype, d_pos, pos_dtype, N, NN, offset, batch_size, box, d_n_neigh,
                                        d_nlist, d_head_list, rmax, d_max_count, nullptr, nullptr, (hipStream_t)stream);
}
