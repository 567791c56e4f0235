// Fused gather-evaluate: HOOMD index neighbor list + positions -> forces, WITHOUT
// materialising the [N, NN, 4] pair-vector tensor.
//
// The contract path writes 268 MB of pair vectors per step and reads them straight back
// (build_pair_vectors -> eval_forces, ~621 MB of algorithmic traffic at C3).  When nothing
// needs the tensor itself -- a traced declarative model, nobody calling get_nlist_array --
// the same arithmetic can run on the pair vector while it is still in registers: per
// particle the kernel reads its index row (4 B per neighbor) and gathers 16-B positions
// from the L2-resident position table, i.e. ~76 MB per step.  Semantics are those of the
// two-kernel path: identical dx (box_math.h pins contraction off), keep unless
// rsq > r_cut^2, NN-slot wrap on overflow (only the last NN kept neighbors contribute),
// zero padding contributes nothing, same per-slot math (pair_math.h).
// htf_config.fused = 1: the nlist side buffer is not filled (opt-in, reported separately).
// htf_config.fused = 2 (STORE): the SAME kernel also writes the pair vectors it has in
// registers -- rows, zero tails and live counts exactly as build_pair_vectors_kernel does
// (bit-identical tensor) -- so the tensor exists for get_nlist_array / training / observables
// but is never read back: the evaluator's 268 MB re-read and its launch disappear from the step.
#ifndef __HIPCC_RTC__
#include <cstdlib>

#include <cstring>
#include <hip/hip_ext.h>
#include <mutex>
#include <vector>
#endif

#include "box_math.h"
#include "htf_common.h"
#include "htf_internal.h"
#include "pair_math.h"

namespace htf {

#ifndef __HIPCC_RTC__
LaunchEvents &launch_events() {
    static thread_local LaunchEvents e;
    return e;
}
const void *&step_epilogue_request() {
    static thread_local const void *r = nullptr;
    return r;
}
int &step_epilogue_level() {
    static thread_local int l = 0;
    return l;
}

#endif

// hipLaunchKernelGGL, or -- when the profiler has handed over a pair of events -- the launch that stamps them with the kernel's
// own begin and end
#define HTF_LAUNCH_TIMED(kernel, grid, block, stream, ...)                                                            \
    do {                                                                                                               \
        LaunchEvents &le_ = launch_events();                                                                           \
        if (le_.start != nullptr && !le_.used) {                                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, le_.start, le_.stop, 0, __VA_ARGS__);                \
            le_.used = true;                                                                                           \
        } else {                                                                                                       \
            hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                           \
        }                                                                                                              \
    } while (0)

constexpr int kFChunk = 3; // index loads hoisted per lane per trip (as pair_vectors.hip: n_neigh <= 192 in one trip)

// A finished row: lanes 0 / 16 / 32 / 48 hold fx, fz, fy, e in `tot` (wave_sum4) and write one component each -- and, when the
// launch carries a step epilogue (htf_internal.h StepEpilogue), go on with the integrator's update of that component.
template <typename PT, bool HALO = true>
__device__ __forceinline__ void store_row_sums(void *__restrict__ force, int out_f64, unsigned w, unsigned idx, unsigned lane, float tot,
                                               const typename Vec4<PT>::type *__restrict__ pos, const StepEpilogue<PT> *ep) {
    if ((lane & 15u) == 0u) {
        const unsigned comp = ((lane >> 4) & 1u) * 2u + (lane >> 5); // 0, 2, 1, 3
        if (out_f64)
            ((double *)force)[(size_t)w * 4 + comp] = (double)tot;
        else
            ((float *)force)[(size_t)w * 4 + comp] = tot;
        // (the row's own position is read again here -- one word per lane, an L2 hit -- rather than kept in registers to the row's
        //  end: holding pi.w and the late pi.xyz cost the four-row form 14 VGPRs, 7 -> 5 waves per SIMD)
        if (ep != nullptr) {
            asm volatile("" ::: "memory"); // (the epilogue's loads stay HERE: hoisted over the rows' trips they cost 16 VGPRs)
            step_epilogue_lane<PT, HALO>(*ep, idx, comp, tot, reinterpret_cast<const PT *>(pos)[(size_t)idx * 4 + comp]);
        }
    }
}

// The step epilogue of a wave's R rows at once (the straight-line row-group forms): lane 4 r + c (< 4 R) owns component c of row r.
// Its velocity and own-position words are loaded at the HEAD of the wave's work (group_epilogue_load: one coalesced 16-lane load
// each, two registers for the whole kernel) and its force component comes out of the rows' wave_sum4 results by a cross-lane read
// (rows 0..3 of `tot`: fx, fz, fy, e), so that behind the last row there is arithmetic and stores, no trip to memory.
template <typename PT, int R, bool HALO>
__device__ __forceinline__ void group_epilogue_load(const StepEpilogue<PT> *ep, const typename Vec4<PT>::type *__restrict__ pos, unsigned idx0,
                                                    unsigned lane, PT &v, PT &own, uint4 &s0, uint4 &s1, unsigned skip_rows = 0u) {
    v = own = (PT)0;
    s0 = s1 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    if (ep != nullptr && lane < 4u * R && !((skip_rows >> (lane >> 2)) & 1u)) {
        const size_t o = (size_t)idx0 * 4 + lane; // rows idx0 .. idx0 + R - 1 are consecutive: 4 R consecutive words
        v = reinterpret_cast<const PT *>(ep->vel)[o];
        own = reinterpret_cast<const PT *>(pos)[o];
        if constexpr (HALO) step_epilogue_slots<PT>(*ep, idx0 + (lane >> 2), s0, s1);
    }
}
template <typename PT, int R, bool HALO>
__device__ __forceinline__ void group_epilogue(const StepEpilogue<PT> *ep, unsigned idx0, unsigned lane, const float (&tot)[R], unsigned skip_rows,
                                               PT v, PT own, const uint4 &s0, const uint4 &s1) {
    if (ep == nullptr) return;
    const unsigned er = lane >> 2, ec = lane & 3u;
    // component c of a row's sums sits in the 16-lane row {0: fx, 2: fz, 1: fy, 3: e} = lanes +0 / +32 / +16 / +48 of this one
    const int src = (int)((lane & 15u) + (ec == 1u ? 32u : (ec == 2u ? 16u : (ec == 3u ? 48u : 0u))));
    float f = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float t = __shfl(tot[r], src);
        f = er == (unsigned)r ? t : f;
    }
    if (lane < 4u * R && !((skip_rows >> er) & 1u)) {
        const PT xn = step_epilogue_core<PT>(*ep, idx0 + er, ec, f, own, v);
        if constexpr (HALO) step_epilogue_halo<PT>(*ep, s0, s1, ec, xn);
    }
}

struct FusedAcc {
    float fx = 0.f, fy = 0.f, fz = 0.f, en = 0.f;
    Virial6 v;
    unsigned npos = 0; // slots with dx > 0: SimModel check_nlist's count (simmodel.py:214-219)
};

// STORE: 0 none; 1 first pass (slot q for q < NN); 2 overflow replay (slot q % NN for q >= s_lo)
template <int KIND, bool VIRIAL, int STORE, typename PT>
__device__ __forceinline__ unsigned fused_sweep(FusedAcc &acc, const typename Vec4<PT>::type *__restrict__ pos,
                                                const unsigned *__restrict__ nl, unsigned nn,
                                                const typename Vec4<PT>::type pi, const BoxT<PT> &box, PT rmaxsq,
                                                unsigned lane, unsigned q_lo, unsigned q_hi, const PotParams &p,
                                                float4 *__restrict__ row, unsigned NN, unsigned s_lo) {
    using PV = typename Vec4<PT>::type;
    unsigned Q = 0;
    for (unsigned base = 0; base < nn; base += 64 * kFChunk) {
        unsigned k[kFChunk];
        PV pk[kFChunk];
#pragma unroll
        for (int t = 0; t < kFChunk; ++t) {
            unsigned j = base + t * 64 + lane;
            k[t] = nl[j < nn ? j : nn - 1];
        }
#pragma unroll
        for (int t = 0; t < kFChunk; ++t) pk[t] = pos[k[t]];
#pragma unroll
        for (int t = 0; t < kFChunk; ++t) {
            if (base + t * 64 >= nn) break; // wave-uniform
            unsigned j = base + t * 64 + lane;
            PT dx, dy, dz;
            PT rsq = pair_vector<PT>(pk[t], pi, box, dx, dy, dz);
            bool keep = (j < nn) && !(rsq > rmaxsq);
            unsigned long long m = __ballot(keep);
            unsigned q = Q + ballot_rank(m);
            Q += __popcll(m);
            if constexpr (STORE != 0) {
                // the tensor row, as build_pair_vectors_kernel writes it (fp32 wire, type as a float)
                const float4 out = make_float4((float)dx, (float)dy, (float)dz, (float)scalar_as_int(pk[t].w));
                if constexpr (STORE == 1) {
                    if (keep && q < NN) store_stream(&row[q], out);
                } else {
                    if (keep && q >= s_lo) store_stream(&row[q % NN], out);
                }
            }
            if (keep && q >= q_lo && q < q_hi) {
                // the model sees the pair vector after tf.cast to fp32 (simmodel.py:226-227)
                const float x = (float)dx, y = (float)dy, z = (float)dz;
                float e, ax, ay, az;
                pair_eval<KIND>(x, y, z, p, e, ax, ay, az, (float)scalar_as_int(pk[t].w), (float)scalar_as_int(pi.w));
                acc.fx += ax;
                acc.fy += ay;
                acc.fz += az;
                acc.en += e;
                acc.npos += x > 0.f ? 1u : 0u;
                if constexpr (VIRIAL) acc.v.add(x, y, z, ax, ay, az);
            }
        }
    }
    return Q;
}

template <int KIND, bool VIRIAL, bool STORE, typename PT, int EPL = 2>
__device__ __forceinline__ void fused_row(const unsigned w, const unsigned lane, const typename Vec4<PT>::type *__restrict__ pos,
                                          unsigned N, unsigned NN, unsigned offset, const BoxT<PT> &box,
                                          const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist,
                                          const unsigned *__restrict__ head_list, PT rmaxsq, void *__restrict__ force,
                                          void *__restrict__ virial9, int out_f64, const PotParams &p,
                                          unsigned *__restrict__ check_count, float4 *__restrict__ positions_out,
                                          float4 *__restrict__ dest, unsigned *__restrict__ counts_io,
                                          const StepEpilogue<PT> *ep = nullptr, float *tot_out = nullptr) {
    // (tot_out: the caller finishes the row itself -- the row-group forms run ONE epilogue for their rows, redone ones included,
    //  so that this routine's copy of it does not set their register count)
    const unsigned idx = w + offset;
    if (idx >= N) return;
    const unsigned nn = n_neigh[idx];
    const unsigned *nl = nlist + head_list[idx];
    const auto pi = pos[idx];
    if (positions_out != nullptr && lane == 0)
        positions_out[w] = make_float4((float)pi.x, (float)pi.y, (float)pi.z, (float)scalar_as_int(pi.w));

    FusedAcc acc;
    float4 *row = STORE ? dest + (size_t)w * NN : nullptr;
    unsigned Q = nn ? fused_sweep<KIND, VIRIAL, STORE ? 1 : 0, PT>(acc, pos, nl, nn, pi, box, rmaxsq, lane, 0u, NN, p, row, NN, 0u) : 0u;
    if constexpr (STORE) {
        // zero tail / delta zero-fill / live count: pair_vectors.hip, same bookkeeping
        const unsigned filled = Q < NN ? Q : NN;
        const unsigned zero_end = counts_io != nullptr ? counts_io[w] : NN;
        for (unsigned sl = filled + lane; sl < zero_end; sl += 64) store_stream(&row[sl], make_float4(0.f, 0.f, 0.f, 0.f));
        if (counts_io != nullptr && lane == 0) counts_io[w] = filled;
    }
    if (Q > NN) {
        // overflow: the reference's slot wrap leaves exactly the last NN kept neighbors
        acc = FusedAcc();
        if constexpr (STORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // first-pass stores retire before the wrap overwrites them
        const unsigned s_lo = Q - NN > NN ? Q - NN : NN;
        fused_sweep<KIND, VIRIAL, STORE ? 2 : 0, PT>(acc, pos, nl, nn, pi, box, rmaxsq, lane, Q - NN, Q, p, row, NN, s_lo);
    }
    // the four row sums as the row-group forms take them (wave_sum4: rows 0..3 of `tot` hold fx, fz, fy, e), so that a row's
    // force does not depend on which routine computed it
    float vscale;
    const float tot = finish_row_sums<KIND>(p, wave_sum4(acc.fx, acc.fy, acc.fz, acc.en), lane, &vscale);
    if (tot_out != nullptr) *tot_out = tot;
    store_row_sums<PT, (EPL >= 2)>(force, out_f64, w, idx, lane, tot, pos, EPL ? ep : nullptr);
    float v6[6];
    if constexpr (VIRIAL) {
        v6[0] = group_sum<64>(acc.v.xx);
        v6[1] = group_sum<64>(acc.v.xy);
        v6[2] = group_sum<64>(acc.v.xz);
        v6[3] = group_sum<64>(acc.v.yy);
        v6[4] = group_sum<64>(acc.v.yz);
        v6[5] = group_sum<64>(acc.v.zz);
        if constexpr (HasRowFn<KIND>::value) {
            // (simmodel.py:509-523 builds the virial from the NORMS of the pair forces: a row function scales it by |F'(rho)|)
#pragma unroll
            for (int c = 0; c < 6; ++c) v6[c] *= vscale;
        }
    }
    unsigned npos = 0;
    if (check_count != nullptr) npos = group_sum_u<64>(acc.npos);
    if (lane == 0) {
        if constexpr (VIRIAL) {
            const float v9[9] = {v6[0], v6[1], v6[2], v6[1], v6[3], v6[4], v6[2], v6[4], v6[5]};
            if (out_f64) {
                double *o = (double *)virial9 + (size_t)w * 9;
#pragma unroll
                for (int c = 0; c < 9; ++c) o[c] = v9[c];
            } else {
                float *o = (float *)virial9 + (size_t)w * 9;
#pragma unroll
                for (int c = 0; c < 9; ++c) o[c] = v9[c];
            }
        }
        if (check_count != nullptr && npos > *(volatile unsigned *)check_count) atomicMax(check_count, npos);
    }
}

template <int KIND, bool VIRIAL, bool STORE, typename PT>
__device__ __forceinline__ void fused_forces_body(const typename Vec4<PT>::type *__restrict__ pos, unsigned N,
                                                           unsigned NN, unsigned offset, unsigned batch,
                                                           BoxT<PT> box, const unsigned *__restrict__ n_neigh,
                                                           const unsigned *__restrict__ nlist,
                                                           const unsigned *__restrict__ head_list, PT rmaxsq,
                                                           void *__restrict__ force, void *__restrict__ virial9,
                                                           int out_f64, PotParams pin, unsigned *__restrict__ check_count,
                                                           float4 *__restrict__ positions_out,
                                                           float4 *__restrict__ dest, unsigned *__restrict__ counts_io) {    const PotParams p = resolve_theta<KIND>(pin);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= batch) return;
    fused_row<KIND, VIRIAL, STORE, PT>(w, lane, pos, N, NN, offset, box, n_neigh, nlist, head_list, rmaxsq, force, virial9,
                                       out_f64, p, check_count, positions_out, dest, counts_io);
}

// (bodies are device functions so that a generated unit -- csrc/jit_unit.hip -- can give its instantiations C names)
template <int KIND, bool VIRIAL, bool STORE, typename PT>
__global__ __launch_bounds__(256) void fused_forces_kernel(const typename Vec4<PT>::type *__restrict__ pos, unsigned N,
                                                           unsigned NN, unsigned offset, unsigned batch,
                                                           BoxT<PT> box, const unsigned *__restrict__ n_neigh,
                                                           const unsigned *__restrict__ nlist,
                                                           const unsigned *__restrict__ head_list, PT rmaxsq,
                                                           void *__restrict__ force, void *__restrict__ virial9,
                                                           int out_f64, PotParams pin, unsigned *__restrict__ check_count,
                                                           float4 *__restrict__ positions_out,
                                                           float4 *__restrict__ dest, unsigned *__restrict__ counts_io) {
    fused_forces_body<KIND, VIRIAL, STORE, PT>(pos, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmaxsq, force, virial9, out_f64,
                                               pin, check_count, positions_out, dest, counts_io);
}

// Measured alternatives for the fast path below (C3, tensor written / not written), with ordinary
// tensor stores: one row per wave 102 / 65 us; two rows 92 / 53 us (this kernel); four rows
// 97 / 66 us; survivors compacted
// into a wave-private LDS row first and evaluated from there in two trips instead of three, with
// full-width tensor stores, 94 / 57 us; evaluation made branch-free so that the two rows' arithmetic
// can pack into v_pk_* instructions 96 / 56 us.  With streaming tensor stores (store_stream,
// htf_common.h) the written variant takes 63 us for one, two and four rows per wave alike; a
// persistent kernel software-pipelined over rows (next row's metadata and list entries in flight
// during the current row's evaluation -- what took the C4 sweep below from 241 to 217 us) 67-77 /
// 63-70 us with 8192-2048 workgroups: here the hardware's own wave turnover does better.
// Workgroups of one or two waves instead of four: 64.5-66 / 65 us against 62.4.
// Round 2 (profiles/r02_fused_kernel_ab.txt, r02_fused_kernel_pmc.txt; written variant, 60.6-64.8 us by box).  A first reading
// of the counters blamed the 16-B position gathers (texture-address path busy ~85 %); serving every gather from an LDS table
// changed nothing (60.5 vs 60.9 us), and SQ_ACTIVE_INST_VALU, which counts QUAD-cycles, says the VALU was 99 % busy: the
// kernel was VALU-issue bound.  Measured while the gathers were the suspect, and NOT faster: 8 waves/SIMD by capping
// SGPRs at 80 (+3 %: 62.5); occupancy capped at 6 / 5 / 4 / 3 waves per SIMD through an LDS pad (63.8 / 66.7 / 72.8 /
// 87.8); workgroups of 8 and 16 waves for L1 sharing between more adjacent rows (63.3 / 75.0); particles renumbered
// in cell order (61.1 vs 60.6; a RANDOM order costs 90.6: locality matters, the lattice order already has it);
// survivors compacted through a wave-private LDS row, evaluated in two trips instead of three and stored as two
// full-width lines per row (66.0-66.4 vs 64.2-64.8 on the same box: fewer TA and VALU instructions, but the LDS
// round trip sits in the middle of every row's dependent chain); the candidate lists of 2 / 3 / 4 rows laid end to end
// and walked as ONE stream, 64 entries per trip (9 trips instead of 12 for four rows: a quarter fewer index loads,
// gathers and evaluation trips; per-row scalars read back with v_readlane, two accumulator sets, a row reduced when
// its last entry has passed): 81.6 / 78.8 / 77.8 us against 61.5, and 67.7 against 48.5 without the tensor -- the
// wave-uniform but data-dependent bookkeeping between trips (current row, split lane, running counts, a branch per
// trip) serialises what the straight-line two-row form lets the hardware overlap; the neighbor-index rows (read once,
// 73 MB) loaded with the nontemporal hint so that they leave the 32-KB L1 to the position lines: 74.5-77.0 us against
// 60.9-63.6 with the tensor written, 48.8-50.2 against 47.2-48.7 without.
// Two rows per wave with ALL their index loads, then all their gathers, issued before any
// arithmetic: twice the bytes in flight per wave slot while the evaluator's VALU work (which,
// unlike the plain build, this kernel has plenty of) runs under the other row's memory latency.
// Fast path for the common case (every row of the pair has 1..192 list entries and does not
// overflow NN); anything else is redone by the generic single-row routine.
template <int KIND, bool STORE, int R, typename PT, int EPL = 2>
__device__ __forceinline__ void fused_rows_group(
    const unsigned w0, const unsigned lane, const typename Vec4<PT>::type *__restrict__ pos, unsigned N, unsigned NN,
    unsigned offset, unsigned batch, const BoxT<PT> &box, const unsigned *__restrict__ n_neigh,
    const unsigned *__restrict__ nlist, const unsigned *__restrict__ head_list, PT rmaxsq, void *__restrict__ force,
    int out_f64, const PotParams &p, unsigned *__restrict__ check_count, float4 *__restrict__ positions_out,
    float4 *__restrict__ dest, unsigned *__restrict__ counts_io, const StepEpilogue<PT> *ep = nullptr, float *tot_out = nullptr) {
    using PV = typename Vec4<PT>::type;
    unsigned nn[R];
    bool fast = w0 + R <= batch;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        nn[r] = n_neigh[(w0 + r < batch ? w0 + r : w0) + offset];
        fast = fast && nn[r] != 0 && nn[r] <= 64 * kFChunk;
    }
    if (!fast) {
        float tl[R];
#pragma unroll
        for (int q_ = 0; q_ < R; ++q_) tl[q_] = 0.f;
#pragma unroll 1
        for (unsigned r = 0; r < (unsigned)R && w0 + r < batch; ++r) {
            float t = 0.f; // (a scalar and a select chain: an array indexed by the loop counter would live in scratch / LDS)
            fused_row<KIND, false, STORE, PT, EPL>(w0 + r, lane, pos, N, NN, offset, box, n_neigh, nlist, head_list, rmaxsq,
                                              force, nullptr, out_f64, p, check_count, positions_out, dest, counts_io, ep, &t);
#pragma unroll
            for (int q_ = 0; q_ < R; ++q_) tl[q_] = r == (unsigned)q_ ? t : tl[q_];
        }
        if (tot_out != nullptr) {
#pragma unroll
            for (int q_ = 0; q_ < R; ++q_) tot_out[q_] = tl[q_];
        }
        return;
    }
    const bool simple_box = box.ortho && box.periodic[0] && box.periodic[1] && box.periodic[2];
    PT ep_v = (PT)0, ep_own = (PT)0;
    uint4 ep_s0 = make_uint4(0, 0, 0, 0), ep_s1 = make_uint4(0, 0, 0, 0);
    if constexpr (EPL != 0) group_epilogue_load<PT, R, (EPL >= 2)>(ep, pos, w0 + offset, lane, ep_v, ep_own, ep_s0, ep_s1);
    float tot_r[R];
    PV pi[R];
    unsigned k[R][kFChunk];
    PV q[R][kFChunk];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned *nl = nlist + head_list[w0 + r + offset];
        pi[r] = pos[w0 + r + offset];
#pragma unroll
        for (int t = 0; t < kFChunk; ++t) {
            const unsigned j = t * 64 + lane;
            k[r][t] = nl[min(j, nn[r] - 1u)];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < kFChunk; ++t) q[r][t] = pos[k[r][t]];
    unsigned redo = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned w = w0 + r;
        if (positions_out != nullptr && lane == 0)
            positions_out[w] = make_float4((float)pi[r].x, (float)pi[r].y, (float)pi[r].z, (float)scalar_as_int(pi[r].w));
        float4 *row = STORE ? dest + (size_t)w * NN : nullptr;
        float fx = 0.f, fy = 0.f, fz = 0.f, en = 0.f;
        unsigned npos = 0, Q = 0;
#pragma unroll
        for (int t = 0; t < kFChunk; ++t) {
            if ((unsigned)t * 64 >= nn[r]) break; // wave-uniform
            // (as in fused_rows_group_tails: live-entry mask from scalar arithmetic, the lane's predicate read back from the
            //  scalar mask, evaluation without a branch -- a dropped candidate contributes exact zeros from x = 1e18)
            const unsigned left = nn[r] - (unsigned)t * 64;
            const unsigned long long valid = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
            const PV pk = q[r][t];
            PT dx, dy, dz;
            PT rsq;
            if (simple_box) { // wave-uniform: orthorhombic, periodic in x, y and z
                asm volatile("" ::: "memory");
                rsq = pair_vector_simple<PT>(pk, pi[r], box, dx, dy, dz);
            } else {
                rsq = pair_vector<PT>(pk, pi[r], box, dx, dy, dz);
            }
            const unsigned long long m = ballot64(!(rsq > rmaxsq)) & valid;
            const unsigned qq = Q + ballot_rank(m);
            Q += __popcll(m);
            const bool keep = inverse_ballot64(m);
            const float x = (float)dx, y = (float)dy, z = (float)dz;
            if constexpr (STORE) {
                unsigned long long ms = m;
                if (Q > NN) { // (wave-uniform) about to overflow: slots bounded lane by lane; the row is redone below
                    asm volatile("" ::: "memory");
                    ms &= ballot64(qq < NN);
                }
                if (inverse_ballot64(ms)) store_stream(&row[qq], make_float4(x, y, z, (float)scalar_as_int(pk.w)));
            }
            float e, ax, ay, az;
            pair_eval_if<KIND>(keep, x, y, z, p, e, ax, ay, az, (float)scalar_as_int(pk.w), (float)scalar_as_int(pi[r].w));
            fx += ax;
            fy += ay;
            fz += az;
            en += e;
            if (check_count != nullptr) { // wave-uniform
                asm volatile("" ::: "memory");
                npos += (keep && x > 0.f) ? 1u : 0u;
            }
        }
        const unsigned filled = Q < NN ? Q : NN;
        if constexpr (STORE) {
            const unsigned zero_end = counts_io != nullptr ? counts_io[w] : NN;
            for (unsigned sl = filled + lane; sl < zero_end; sl += 64) store_stream(&row[sl], make_float4(0.f, 0.f, 0.f, 0.f));
            if (counts_io != nullptr && lane == 0) counts_io[w] = filled;
        }
        tot_r[r] = 0.f;
        if (Q > NN) { // overflow (an error upstream): the generic routine reproduces the slot wrap
            redo |= 1u << r;
            continue;
        }
        const float tot = finish_row_sums<KIND>(p, wave_sum4(fx, fy, fz, en), lane); // rows 0..3: fx, fz, fy, e
        tot_r[r] = tot;
        store_row_sums<PT>(force, out_f64, w, w + offset, lane, tot, pos, nullptr);
        if (check_count != nullptr) {
            npos = group_sum_u<64>(npos);
            if (lane == 0 && npos > *(volatile unsigned *)check_count) atomicMax(check_count, npos);
        }
    }
    if constexpr (EPL != 0) group_epilogue<PT, R, (EPL >= 2)>(ep, w0 + offset, lane, tot_r, redo, ep_v, ep_own, ep_s0, ep_s1);
#pragma unroll 1
    for (unsigned r = 0; r < (unsigned)R; ++r) // ONE code copy: a row's result must not depend on its place in the group
        if ((redo >> r) & 1u) {
            if constexpr (STORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float t = 0.f;
            fused_row<KIND, false, STORE, PT, EPL>(w0 + r, lane, pos, N, NN, offset, box, n_neigh, nlist, head_list, rmaxsq,
                                              force, nullptr, out_f64, p, check_count, positions_out, dest, counts_io, ep, &t);
#pragma unroll
            for (int q_ = 0; q_ < R; ++q_) tot_r[q_] = r == (unsigned)q_ ? t : tot_r[q_];
        }
    if (tot_out != nullptr) {
#pragma unroll
        for (int r = 0; r < R; ++r) tot_out[r] = tot_r[r];
    }
}

// Tails merged: a row of ~139 candidates is two full trips and a third with ~11 live lanes, and a gather or an
// evaluation trip costs the same whether 64 or 11 of its lanes are live.  Here the first 128 entries of each of the R
// rows are walked as above (straight-line code, compile-time structure), and the R tails share ONE trip: lane l
// belongs to the row whose tail covers it (prefix sums of the tail lengths; fast path: they fit 64 lanes together).
// R = 4: 9 trips instead of 12.  Only the shared trip pays for per-lane row selection (list head, own position,
// rank base, tensor row) and for splitting its contributions over R accumulator sets.
template <int KIND, bool STORE, int R, typename PT, int EPL = 2>
__device__ __forceinline__ void fused_rows_group_tails(
    const unsigned w0, const unsigned lane, const typename Vec4<PT>::type *__restrict__ pos, unsigned N, unsigned NN,
    unsigned offset, unsigned batch, const BoxT<PT> &box, const unsigned *__restrict__ n_neigh,
    const unsigned *__restrict__ nlist, const unsigned *__restrict__ head_list, PT rmaxsq, void *__restrict__ force,
    int out_f64, const PotParams &p, unsigned *__restrict__ check_count, float4 *__restrict__ positions_out,
    float4 *__restrict__ dest, unsigned *__restrict__ counts_io, const StepEpilogue<PT> *ep = nullptr) {
    using PV = typename Vec4<PT>::type;
    unsigned nn[R], S[R + 1];
    // the straight-line path is the common case only: an orthorhombic box periodic in x, y and z, no check_nlist count.
    // Everything else takes the two-row form below, which carries the general BoxDim arithmetic.
    bool fast = w0 + R <= batch && box.ortho && box.periodic[0] && box.periodic[1] && box.periodic[2] && check_count == nullptr;
    S[0] = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        nn[r] = n_neigh[(w0 + r < batch ? w0 + r : w0) + offset];
        fast = fast && nn[r] != 0 && nn[r] <= 192u;
        S[r + 1] = S[r] + (nn[r] > 128u ? nn[r] - 128u : 0u);
    }
    // The step epilogue of the group's rows is ONE piece of code at the end of this routine (group_epilogue), whichever routine
    // computed a row: the fallbacks below are compiled WITHOUT one (EPL = 0) and hand their row sums back.  With their own copies
    // the kernel needed 77 (integrator) / 82 (+ halo) VGPRs against 53 -- six / five waves per SIMD instead of eight -- although
    // its straight-line path needs no more than without an epilogue (round 6, tools: a 9-second probe unit of this kernel alone).
    PT ep_v, ep_own;
    uint4 ep_s0, ep_s1;
    float tot_r[R];
    if (!fast || S[R] > 64u) { // (wave-uniform) the plain two-rows-at-a-time form handles everything else
        unsigned absent = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            tot_r[r] = 0.f;
            absent |= (w0 + r < batch && w0 + r + offset < N) ? 0u : (1u << r);
        }
#pragma unroll
        for (int r = 0; r < R; r += 2)
            if (w0 + r < batch) {
                if (r + 1 < R)
                    fused_rows_group<KIND, STORE, 2, PT, 0>(w0 + r, lane, pos, N, NN, offset, batch, box, n_neigh, nlist, head_list,
                                                       rmaxsq, force, out_f64, p, check_count, positions_out, dest, counts_io, nullptr, &tot_r[r]);
                else // (an odd group's last row: the row after it belongs to the next wave)
                    fused_row<KIND, false, STORE, PT, 0>(w0 + r, lane, pos, N, NN, offset, box, n_neigh, nlist, head_list, rmaxsq,
                                                    force, nullptr, out_f64, p, check_count, positions_out, dest, counts_io, nullptr, &tot_r[r]);
            }
        if constexpr (EPL != 0) {
            if (ep != nullptr) { // (the rare path: its loads sit where they are needed)
                group_epilogue_load<PT, R, (EPL >= 2)>(ep, pos, w0 + offset, lane, ep_v, ep_own, ep_s0, ep_s1, absent);
                group_epilogue<PT, R, (EPL >= 2)>(ep, w0 + offset, lane, tot_r, absent, ep_v, ep_own, ep_s0, ep_s1);
            }
        }
        return;
    }
    PV pi[R];
    unsigned head[R];
    unsigned k[R][2], kt;
    PV q[R][2], qt;
    // which row this lane's tail entry belongs to, and its entry index there
    unsigned rl = 0;
#pragma unroll
    for (int r = 1; r < R; ++r) rl += lane >= S[r] ? 1u : 0u;
    unsigned head_l = 0, s_l = 0, nn_l = 0;
    unsigned zero_end_r[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        zero_end_r[r] = NN; // (delta zero fill: where the previous call's row ended -- read with the row's other scalars, used last)
        if constexpr (STORE)
            if (counts_io != nullptr) zero_end_r[r] = counts_io[w0 + r];
        head[r] = head_list[w0 + r + offset];
        pi[r] = pos[w0 + r + offset];
        const unsigned *nl = nlist + head[r];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned j = t * 64 + lane;
            k[r][t] = nl[min(j, nn[r] - 1u)];
        }
        head_l = rl == (unsigned)r ? head[r] : head_l;
        s_l = rl == (unsigned)r ? S[r] : s_l;
        nn_l = rl == (unsigned)r ? nn[r] : nn_l;
    }
    const bool tail_live = lane < S[R];
    const unsigned jt = 128u + (lane - s_l);
    kt = nlist[head_l + (tail_live ? jt : nn_l - 1)];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) q[r][t] = load_neighbor(pos, k[r][t]);
    qt = load_neighbor(pos, kt);
    float fx[R], fy[R], fz[R], en[R];
    unsigned npos[R], Q[R];
    // the first 128 entries of every row.  The VALU is the unit this kernel saturates, so: the minimum image without the
    // tilt and non-periodic cases (12 instructions), the live-entry mask of a trip from scalar arithmetic, and the evaluation
    // WITHOUT a branch -- a dropped candidate is evaluated at x = 1e18, where s^6 underflows and energy and force are
    // exact zeros (pair_eval_if) -- so that only the tensor store is predicated.
    // (the box lengths in vector registers -- a vector instruction with a scalar source issues at half the rate when two or more
    //  waves share the SIMD, tools/valu_cost_probe.hip -- bought nothing here: 55.4 / 37.5 us against 55.4 / 37.8, DESIGN A.0)
    auto pair_vec = [&](const PV &pk, const PV &pc, PT &dx, PT &dy, PT &dz) { return pair_vector_simple<PT>(pk, pc, box, dx, dy, dz); };
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float4 *row = STORE ? dest + (size_t)(w0 + r) * NN : nullptr;
        fx[r] = fy[r] = fz[r] = en[r] = 0.f;
        npos[r] = 0;
        Q[r] = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if ((unsigned)t * 64 >= nn[r]) break; // wave-uniform
            const unsigned left = nn[r] - (unsigned)t * 64; // >= 1
            const unsigned long long valid = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
            const PV pk = q[r][t];
            PT dx, dy, dz;
            const PT rsq = pair_vec(pk, pi[r], dx, dy, dz);
            const unsigned long long m = ballot64(!(rsq > rmaxsq)) & valid;
            const unsigned qq = Q[r] + ballot_rank(m);
            Q[r] += __popcll(m);
            const bool keep = inverse_ballot64(m); // the scalar mask read as this lane's predicate: no VALU
            const float x = (float)dx, y = (float)dy, z = (float)dz;
            if constexpr (STORE) {
                unsigned long long ms = m;
                if (Q[r] > NN) { // (wave-uniform) a row about to overflow: its slots are bounded lane by lane, and it is redone below
                    asm volatile("" ::: "memory");
                    ms &= ballot64(qq < NN);
                }
                if (inverse_ballot64(ms)) store_stream(&row[qq], make_float4(x, y, z, (float)scalar_as_int(pk.w)));
            }
            float e, ax, ay, az;
            pair_eval_if<KIND>(keep, x, y, z, p, e, ax, ay, az, (float)scalar_as_int(pk.w), (float)scalar_as_int(pi[r].w));
            fx[r] += ax;
            fy[r] += ay;
            fz[r] += az;
            en[r] += e;
        }
    }
    // the epilogue's inputs (velocity, own position, message slots of lane 4 r + c): loaded HERE -- behind the straight-line part,
    // whose gathered positions are the kernel's register peak, ahead of the tail trip and the reductions that hide the trip to
    // memory (at the head of the wave's work they cost 10 VGPRs; behind the last row, 10 % of the kernel's time)
    asm volatile("" ::: "memory");
    if constexpr (EPL != 0) group_epilogue_load<PT, R, (EPL >= 2)>(ep, pos, w0 + offset, lane, ep_v, ep_own, ep_s0, ep_s1, 0u);
    // the shared tail trip
    if (S[R] != 0) {
        PV pil = pi[0];
#pragma unroll
        for (int r = 1; r < R; ++r) {
            pil.x = rl == (unsigned)r ? pi[r].x : pil.x;
            pil.y = rl == (unsigned)r ? pi[r].y : pil.y;
            pil.z = rl == (unsigned)r ? pi[r].z : pil.z;
            pil.w = rl == (unsigned)r ? pi[r].w : pil.w; // (read by generated bodies only: dead code for the closed forms)
        }
        PT dx, dy, dz;
        const PT rsq = pair_vec(qt, pil, dx, dy, dz);
        const unsigned long long m = ballot64(!(rsq > rmaxsq)) & (S[R] >= 64u ? ~0ull : ((1ull << S[R]) - 1ull));
        // rank inside its own row: kept lanes below me, minus those that belong to earlier rows, plus the row's count so far
        unsigned base_l = Q[0];
#pragma unroll
        for (int r = 1; r < R; ++r) {
            const unsigned before = (unsigned)__popcll(S[r] >= 64u ? m : (m & ((1ull << S[r]) - 1ull)));
            base_l = rl == (unsigned)r ? Q[r] - before : base_l;
        }
        const unsigned qq = base_l + ballot_rank(m);
        const bool keep = inverse_ballot64(m);
        const float xt = (float)dx, yt = (float)dy, zt = (float)dz;
        if constexpr (STORE)
            if (keep && qq < NN) store_stream(dest + (size_t)(w0 + rl) * NN + qq, make_float4(xt, yt, zt, (float)scalar_as_int(qt.w)));
        float e, ax, ay, az;
        pair_eval_if<KIND>(keep, xt, yt, zt, p, e, ax, ay, az, (float)scalar_as_int(qt.w), (float)scalar_as_int(pil.w));
        const unsigned px = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool mine = rl == (unsigned)r;
            fx[r] += mine ? ax : 0.f;
            fy[r] += mine ? ay : 0.f;
            fz[r] += mine ? az : 0.f;
            en[r] += mine ? e : 0.f;
            npos[r] += mine ? px : 0u;
            const unsigned long long seg = (S[r + 1] >= 64u ? ~0ull : ((1ull << S[r + 1]) - 1ull)) & ~(S[r] >= 64u ? ~0ull : ((1ull << S[r]) - 1ull));
            Q[r] += (unsigned)__popcll(m & seg);
        }
    }
    unsigned redo = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned w = w0 + r;
        if (positions_out != nullptr && lane == 0)
            positions_out[w] = make_float4((float)pi[r].x, (float)pi[r].y, (float)pi[r].z, (float)scalar_as_int(pi[r].w));
        const unsigned filled = Q[r] < NN ? Q[r] : NN;
        if constexpr (STORE) {
            float4 *row = dest + (size_t)w * NN;
            const unsigned zero_end = zero_end_r[r];
            for (unsigned sl = filled + lane; sl < zero_end; sl += 64) store_stream(&row[sl], make_float4(0.f, 0.f, 0.f, 0.f));
            if (counts_io != nullptr && lane == 0) counts_io[w] = filled;
        }
        tot_r[r] = 0.f;
        if (Q[r] > NN) {
            redo |= 1u << r;
            continue;
        }
        // the row's four sums together: rows 0..3 of `tot` hold fx, fz, fy, e; lanes 0 / 16 / 32 / 48 write one component each
        const float tot = finish_row_sums<KIND>(p, wave_sum4(fx[r], fy[r], fz[r], en[r]), lane);
        tot_r[r] = tot;
        store_row_sums<PT>(force, out_f64, w, w + offset, lane, tot, pos, nullptr);
        if (check_count != nullptr) {
            const unsigned np = group_sum_u<64>(npos[r]);
            if (lane == 0 && np > *(volatile unsigned *)check_count) atomicMax(check_count, np);
        }
    }
#pragma unroll 1
    for (unsigned r = 0; r < (unsigned)R; ++r)
        if ((redo >> r) & 1u) {
            if constexpr (STORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float t = 0.f;
            fused_row<KIND, false, STORE, PT, 0>(w0 + r, lane, pos, N, NN, offset, box, n_neigh, nlist, head_list, rmaxsq,
                                            force, nullptr, out_f64, p, check_count, positions_out, dest, counts_io, nullptr, &t);
#pragma unroll
            for (int q_ = 0; q_ < R; ++q_) tot_r[q_] = r == (unsigned)q_ ? t : tot_r[q_];
        }
    if constexpr (EPL != 0) group_epilogue<PT, R, (EPL >= 2)>(ep, w0 + offset, lane, tot_r, 0u, ep_v, ep_own, ep_s0, ep_s1);
}

// (Measured and removed, commit 6b65261: FOUR INDICES PER LANE -- the wave's four rows own 16 lanes each and a lane reads the
//  16-B blocks [4s, 4s+4) and [64+4s, 64+4s+4) of its row's index list, two index-load instructions for the first 128
//  entries of four rows instead of eight; slots from the ballots of the eight evaluation steps, same tensor bit for bit.
//  A lane's four survivors then land in consecutive slots and the lanes of one store instruction are ~43 B apart:
//  245 us with streaming stores, 97 us with ordinary ones, against 61-66; and without the tensor 48.7 against 44.9 --
//  the index-load instruction count is not what the kernel waits for.  profiles/r02_fused_kernel_ab.txt, batch 11.)
#ifndef HTF_TAILS_MINB_F64
#define HTF_TAILS_MINB_F64 1
#endif
// EP: the launch carries a step epilogue (htf_internal.h StepEpilogue).  A template parameter, not a null test: the epilogue's
// address arithmetic and loads cost the four-row form 16 VGPRs (69 -> 85: five waves per SIMD instead of seven, 57 -> 60 us at C3
// with the epilogue switched OFF), so launches without one run the code they always ran.
template <int KIND, bool STORE, int R, typename PT, int EP = 0>
__global__ __launch_bounds__(256, sizeof(PT) == 8 ? HTF_TAILS_MINB_F64 : 1) void fused_forces_tails_kernel(
    const typename Vec4<PT>::type *__restrict__ pos, unsigned N, unsigned NN, unsigned offset, unsigned batch,
    BoxT<PT> box, const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist,
    const unsigned *__restrict__ head_list, PT rmaxsq, void *__restrict__ force, int out_f64, PotParams pin,
    unsigned *__restrict__ check_count, float4 *__restrict__ positions_out, float4 *__restrict__ dest,
    unsigned *__restrict__ counts_io, const StepEpilogue<PT> *__restrict__ ep) {
    const PotParams p = resolve_theta<KIND>(pin);
    const unsigned lane = threadIdx.x & 63u;
    // (one wave per group of R rows.  Round 3, measured: the same body under a grid-stride loop, waves persistent at 4 ... 32
    //  workgroups per CU: 60.3-71.4 us against 57.2, and 61.2 with the loop running once -- the loop alone costs the straight-line
    //  code its schedule; fp64 positions 88.7-93.8 against 67.)
    const unsigned w0 = R * __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w0 >= batch) return;
    fused_rows_group_tails<KIND, STORE, R, PT, EP>(w0, lane, pos, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmaxsq,
                                                   force, out_f64, p, check_count, positions_out, dest, counts_io, EP ? ep : nullptr);
}

// The kernel strides over the row groups so that HTF_FUSED_GRID=<workgroups per CU> can launch it
// persistently for A/B runs.  tools/store_probe.hip says a persistent grid helps a bare
// load-chain + streaming-store kernel (65 -> 46 us); this kernel, whose rows differ in length
// and which has arithmetic to hide its loads under, is best with one group per wave
// (C3, tensor written: 62.5 us; 4 / 8 / 12 / 16 workgroups per CU: 84 / 78 / 69 / 68 us), the default.
template <int KIND, bool STORE, int R, typename PT, int EPL = 2>
__device__ __forceinline__ void fused_forces_rows2_body(
    const typename Vec4<PT>::type *__restrict__ pos, unsigned N, unsigned NN, unsigned offset, unsigned batch,
    BoxT<PT> box, const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist,
    const unsigned *__restrict__ head_list, PT rmaxsq, void *__restrict__ force, int out_f64, PotParams pin,
    unsigned *__restrict__ check_count, float4 *__restrict__ positions_out, float4 *__restrict__ dest,
    unsigned *__restrict__ counts_io, const StepEpilogue<PT> *ep = nullptr) {
    const PotParams p = resolve_theta<KIND>(pin);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wv = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const unsigned nw = (gridDim.x * blockDim.x) >> 6;
#pragma unroll 1
    for (unsigned w0 = R * wv; w0 < batch; w0 += R * nw)
        fused_rows_group<KIND, STORE, R, PT, EPL>(w0, lane, pos, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmaxsq,
                                             force, out_f64, p, check_count, positions_out, dest, counts_io, ep);
}

template <int KIND, bool STORE, int R, typename PT, int EP = 0>
__global__ __launch_bounds__(256) void fused_forces_rows2_kernel(
    const typename Vec4<PT>::type *__restrict__ pos, unsigned N, unsigned NN, unsigned offset, unsigned batch,
    BoxT<PT> box, const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist,
    const unsigned *__restrict__ head_list, PT rmaxsq, void *__restrict__ force, int out_f64, PotParams pin,
    unsigned *__restrict__ check_count, float4 *__restrict__ positions_out, float4 *__restrict__ dest,
    unsigned *__restrict__ counts_io, const StepEpilogue<PT> *__restrict__ ep) {
    fused_forces_rows2_body<KIND, STORE, R, PT, EP>(pos, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmaxsq, force, out_f64, pin,
                                                    check_count, positions_out, dest, counts_io, EP ? ep : nullptr);
}

#ifdef HTF_JIT_UNIT
} // namespace htf  (a generated unit takes the templates above and nothing else of this file)
#else
template <int KIND, bool VIRIAL, typename PT>
static int launch_fused(const void *pos, unsigned N, unsigned NN, unsigned offset, unsigned batch, const htf_box *hb,
                        const unsigned *n_neigh, const unsigned *nlist, const unsigned *head_list, double rmax,
                        void *force, void *virial9, int out_f64, const PotParams &p, unsigned *check_count,
                        float4 *positions_out, float4 *dest, unsigned *counts_io, hipStream_t s) {
    BoxT<PT> b = make_boxt<PT>(hb);
    PT rc = (PT)rmax;
    // the stand-in integrator as this launch's epilogue: a descriptor in device memory, built for this context's Scalar when it was
    // registered (context.hip htfs_set_step_epilogue sets the request around the call; a virial request -- the one-row kernel --
    // carries none).  A POINTER, not a by-value struct: the row routines take its address, and an address-taken kernel argument is
    // copied to scratch (312 bytes per lane of private memory for a kernel that had none).
    const StepEpilogue<PT> *ep = VIRIAL ? nullptr : static_cast<const StepEpilogue<PT> *>(step_epilogue_request());
    // (the epilogue forms exist for the two potentials BASELINE's configurations time; htfs_set_step_epilogue says so)
    // (fp32 positions only: under a HOOMD DOUBLE build the epilogue took the four-row form from 65 to 87 us -- 9.7 k against
    //  11.1 k steps/s at C3 -- so the fp64 wire keeps the integrator's own launch and no fp64 epilogue form is compiled)
    constexpr bool kEpilogueKind = !VIRIAL && (sizeof(PT) == 4 || HTF_EPILOGUE_F64) && (KIND == HTF_POT_LJ || KIND == HTF_POT_WCA);
    // level 1: the integrator alone (77 VGPRs in the four-row form, six waves per SIMD); level 2: + a brick's halo messages (82: five)
    const int ep_level = step_epilogue_level();
    if constexpr (!VIRIAL) {
#ifdef HTF_AB_VARIANTS // A/B builds only (CXXFLAGS_EXTRA=-DHTF_AB_VARIANTS): every form stays selectable
        static const char *rows_env = getenv("HTF_FUSED_ROWS"); // 1 | 2 | 4 rows per wave
        const int rows = rows_env ? atoi(rows_env) : 2;
        static const char *grid_env = getenv("HTF_FUSED_GRID"); // workgroups per CU, 0 = one wave per group
        static const int per_cu = grid_env ? atoi(grid_env) : 0;
        static const char *tails_env = getenv("HTF_FUSED_TAILS");
#else
        constexpr int rows = 2, per_cu = 0;
        constexpr const char *tails_env = nullptr;
#endif
        static const int n_cu = device_cu_count();
#define HTF_ROWS_LAUNCH_EP(ST, RR, EPV, EPP)                                                                           \
    HTF_LAUNCH_TIMED((fused_forces_rows2_kernel<KIND, ST, RR, PT, kEpilogueKind ? EPV : 0>), dim3(grid), dim3(256), s,  \
                     (const typename Vec4<PT>::type *)pos, N, NN, offset, batch, b, n_neigh, nlist, head_list,         \
                     (PT)(rc * rc), force, out_f64, p, check_count, positions_out, dest, counts_io, EPP)
#define HTF_ROWS_LAUNCH(ST, RR)                                                                                        \
    const unsigned full = ((batch + RR - 1) / RR + 3) / 4;                                                             \
    const unsigned grid = per_cu > 0 && (unsigned)(per_cu * n_cu) < full ? (unsigned)(per_cu * n_cu) : full;           \
    do {                                                                                                               \
        if (kEpilogueKind && ep != nullptr && ep_level >= 2) {                                                         \
            HTF_ROWS_LAUNCH_EP(ST, RR, 2, ep);                                                                         \
        } else if (kEpilogueKind && ep != nullptr) {                                                                   \
            HTF_ROWS_LAUNCH_EP(ST, RR, 1, ep);                                                                         \
        } else {                                                                                                       \
            HTF_ROWS_LAUNCH_EP(ST, RR, 0, (const StepEpilogue<PT> *)nullptr);                                          \
        }                                                                                                              \
    } while (0)
        // default for fp32 positions and batches of >= 16 384 rows (65 536 until the VALU diet of round 2: at 32 000 rows the
        // four-row form now takes 17.8 us against 19.0-19.4): four rows per wave with their tails in one trip
        // (58.0 us against 60.5 for the two-row form at C3 in isolation, 61.5-62.1 against 62.5-63.3 inside the MD loop,
        // 47.1 against 48.6 without the tensor, 80.7 against 85.4 beside the training stream; two rows with a shared
        // tail: no gain; at 32 768 rows the four-row form LOSES, 24.0 against 22.8: a quarter of the waves on a grid that
        // barely fills the chip).  HTF_FUSED_TAILS=0 selects the two-row form everywhere, whose forces do not depend on
        // how a step is cut into batches / row ranges, bit for bit.
        // fp64 positions (HOOMD in double precision): two rows per wave with a shared tail trip -- 66.8-69.8 us at C3 against
        // 78.1-78.4 for the plain two-row form and 68.7-72.9 for four rows (82 VGPRs, 83 spilled SGPRs); same-box A/B in
        // profiles/r03_f64_kernel_ab.txt.  Every fp64 VALU instruction of the kernel issues at the fp32 rate
        // (tools/valu_rate_probe.hip); what the fp64 wire costs is the second 16-B gather instruction per candidate.
        // Round 3, 32 000 rows (C2): two rows 18.5 us, four rows 19.1, plain two-row form 19.4; 62 500 rows: 31.7 / 31.6 -- half
        // as many waves on a grid that fills the chip only twice is what the four-row form loses there, so it starts at 49 152.
        // fp64 positions, after the 28-byte neighbor loads (one register less per gathered position): two / three / four rows
        // 68.0-68.8 / 66.1-67.2 / 66.6-67.0 us on one box -- the four-row form for both precisions.
        // Shipped forms: the plain two-row form for every closed form and batch size (the generic fallback), and the merged-tail
        // forms -- two rows from 16 384, four rows from 49 152 rows; ~100 KB of straight-line code per instantiation -- for the two
        // potentials BASELINE's configurations time (LJModel, WCARepulsion); the polynomial, the trainable LJ, the Gaussian and
        // SimplePotential take the two-row form at every size (~4 % slower at 131 072 rows).  Three rows per wave, four plain rows,
        // persistent grids and the one-row kernel lost their A/Bs and are compiled in variants builds only.
        constexpr bool kTails = KIND == HTF_POT_LJ || KIND == HTF_POT_WCA;
        int tails = batch >= 16384u ? (batch >= 49152u ? 4 : 2) : 0;
        // (until the epilogue became ONE copy at the end of a row group the four-row form with a halo-packing epilogue needed 82
        //  VGPRs -- five waves per SIMD -- and lost to the two-row form, 71.5 us against 60.4 at C3; now 58: -DHTF_HALO_FOUR_ROWS=0
        //  restores the old rule for A/B runs)
#ifndef HTF_HALO_FOUR_ROWS
#define HTF_HALO_FOUR_ROWS 1
#endif
        if (!HTF_HALO_FOUR_ROWS && kEpilogueKind && ep != nullptr && ep_level >= 2 && tails == 4) tails = 2;
        if (tails_env) tails = atoi(tails_env);
#ifndef HTF_AB_VARIANTS
        if (!kTails) tails = 0;
#endif
        if (tails == 2 || tails == 3 || tails == 4) {
#define HTF_TAILS_LAUNCH_EP(ST, RR, EPV, EPP)                                                                          \
    HTF_LAUNCH_TIMED((fused_forces_tails_kernel<KIND, ST, RR, PT, kEpilogueKind ? EPV : 0>), grid_t, dim3(256), s,      \
                     (const typename Vec4<PT>::type *)pos, N, NN, offset, batch, b, n_neigh, nlist, head_list,         \
                     (PT)(rc * rc), force, out_f64, p, check_count, positions_out, dest, counts_io, EPP)
#define HTF_TAILS_LAUNCH(ST, RR)                                                                                       \
    do {                                                                                                               \
        const dim3 grid_t(((batch + RR - 1) / RR + 3) / 4);                                                            \
        if (kEpilogueKind && ep != nullptr && ep_level >= 2) {                                                         \
            HTF_TAILS_LAUNCH_EP(ST, RR, 2, ep);                                                                        \
        } else if (kEpilogueKind && ep != nullptr) {                                                                   \
            HTF_TAILS_LAUNCH_EP(ST, RR, 1, ep);                                                                        \
        } else {                                                                                                       \
            HTF_TAILS_LAUNCH_EP(ST, RR, 0, (const StepEpilogue<PT> *)nullptr);                                         \
        }                                                                                                              \
    } while (0)
#ifdef HTF_AB_VARIANTS
            if (dest != nullptr) {
                if (tails == 2) HTF_TAILS_LAUNCH(true, 2); else if (tails == 3) HTF_TAILS_LAUNCH(true, 3); else HTF_TAILS_LAUNCH(true, 4);
            } else {
                if (tails == 2) HTF_TAILS_LAUNCH(false, 2); else if (tails == 3) HTF_TAILS_LAUNCH(false, 3); else HTF_TAILS_LAUNCH(false, 4);
            }
#else
            if constexpr (kTails) {
                if (dest != nullptr) {
                    if (tails == 2) HTF_TAILS_LAUNCH(true, 2); else HTF_TAILS_LAUNCH(true, 4);
                } else {
                    if (tails == 2) HTF_TAILS_LAUNCH(false, 2); else HTF_TAILS_LAUNCH(false, 4);
                }
            }
#endif
#undef HTF_TAILS_LAUNCH
#undef HTF_TAILS_LAUNCH_EP
            return check_launch("fused_forces_tails_kernel");
        }
#ifdef HTF_AB_VARIANTS
        if (rows == 2 || rows == 4) {
            if (rows == 2) {
                if (dest != nullptr) { HTF_ROWS_LAUNCH(true, 2); } else { HTF_ROWS_LAUNCH(false, 2); }
            } else {
                if (dest != nullptr) { HTF_ROWS_LAUNCH(true, 4); } else { HTF_ROWS_LAUNCH(false, 4); }
            }
            return check_launch("fused_forces_rows2_kernel");
        }
#else
        if (dest != nullptr) { HTF_ROWS_LAUNCH(true, 2); } else { HTF_ROWS_LAUNCH(false, 2); }
        return check_launch("fused_forces_rows2_kernel");
#endif
#undef HTF_ROWS_LAUNCH
#undef HTF_ROWS_LAUNCH_EP
    }
#ifndef HTF_AB_VARIANTS
    if constexpr (VIRIAL) // (the one-row kernel: every virial request; without a virial only in variants builds, HTF_FUSED_ROWS=1)
#endif
    {
        if (dest != nullptr)
            HTF_LAUNCH_TIMED((fused_forces_kernel<KIND, VIRIAL, true, PT>), dim3((batch + 3) / 4), dim3(256), s,
                               (const typename Vec4<PT>::type *)pos, N, NN, offset, batch, b, n_neigh, nlist, head_list,
                               (PT)(rc * rc), force, virial9, out_f64, p, check_count, positions_out, dest, counts_io);
        else
            HTF_LAUNCH_TIMED((fused_forces_kernel<KIND, VIRIAL, false, PT>), dim3((batch + 3) / 4), dim3(256), s,
                               (const typename Vec4<PT>::type *)pos, N, NN, offset, batch, b, n_neigh, nlist, head_list,
                               (PT)(rc * rc), force, virial9, out_f64, p, check_count, positions_out, dest, counts_io);
    }
    return check_launch("fused_forces_kernel");
}

template <int KIND>
static int launch_fused_k(const void *pos, int pos_dtype, unsigned N, unsigned NN, unsigned offset, unsigned batch,
                          const htf_box *hb, const unsigned *n_neigh, const unsigned *nlist,
                          const unsigned *head_list, double rmax, void *force, void *virial9, int out_f64,
                          const PotParams &p, unsigned *check_count, float4 *positions_out, float4 *dest,
                          unsigned *counts_io, hipStream_t s) {
#define HTF_FUSED(V, T) launch_fused<KIND, V, T>(pos, N, NN, offset, batch, hb, n_neigh, nlist, head_list, rmax, force, virial9, out_f64, p, check_count, positions_out, dest, counts_io, s)
    if (pos_dtype == HTF_F32) return virial9 ? HTF_FUSED(true, float) : HTF_FUSED(false, float);
    return virial9 ? HTF_FUSED(true, double) : HTF_FUSED(false, double);
#undef HTF_FUSED
}

// ---------------------------------------------------------------------------- config C4 in ONE kernel
// htf_build_pair_vectors + htf_eval_forces2 fused: base potential A, Gaussian CV channel B,
// the per-block CV partial sums and the compute_rdf histogram, all from the pair vectors while
// they are in registers; the tensor is written too when `dest` is given (bit-identical).
// Rows with <= 192 list entries keep their pair vectors in registers between the counting pass
// and the evaluating pass; longer rows re-gather.  Overflow (Q > NN, an error upstream) needs no
// replay here: only the NN survivors q >= Q - NN are stored / evaluated, each into its own slot
// q % NN.  Persistent blocks: the LDS histogram is flushed once per block.
constexpr unsigned kRdfMaxBins2 = 1024;
#ifndef HTF_FUSED2_ROWS_DEFAULT
#define HTF_FUSED2_ROWS_DEFAULT 2
#endif

struct Rdf2 {
    float r0, r1;
    unsigned nb;
    unsigned *hist;
    const float *edges; // [nb + 1] bin thresholds on the SQUARED norm (rdf_edges below)
};

// tf.histogram_fixed_width's bin of a pair vector,  clamp(floor(nb * ((sqrt_rn(fl(fl(x x + y y) + z z)) - r0) / (r1 - r0)))),
// is a non-decreasing step function of the squared norm s: every operation in it is monotone.  edges[b] = the smallest
// fp32 s whose bin is >= b (b = 1 .. nb-1; edges[0] = -1, edges[nb] = +inf), found by bisection over the fp32 bit
// patterns with the SAME host arithmetic (IEEE sqrtf, division, floorf).  The sweep then needs no correctly rounded
// square root and no division per slot: a 1-ulp v_sqrt_f32 guess g, then  g - (s < edges[g]) + (s >= edges[g+1]).
static float rdf_bin_exact_host(float s, float r0, float r1, unsigned nb) {
    const volatile float r = sqrtf(s);
    const volatile float d = r - r0;
    const volatile float w = r1 - r0;
    const volatile float q = d / w;
    const volatile float pqr = (float)nb * q;
    const float fi = floorf(pqr);
    return fi < 0.f ? 0.f : (fi > (float)(nb - 1) ? (float)(nb - 1) : fi);
}

// One immutable device table per (device, range, bins), built the first time those parameters are seen and never rewritten:
// a sweep of an earlier call -- on another stream, or replayed from a captured graph -- may still be reading its table when the
// parameters change (the first version kept ONE table and overwrote it in place).  The build (a bisection on the
// host, one H2D copy, one synchronize) therefore happens once per parameter set, before any kernel that uses the table is
// enqueued; calls with known parameters do no host work here and are legal under stream capture.  The cache is process-wide
// (a handful of 4-KB tables for any realistic run); entries are never freed.
struct RdfEdgeTable {
    float r0, r1;
    unsigned nb;
    int dev;
    float *d;
};

static int rdf_edges(float r0, float r1, unsigned nb, hipStream_t stream, const float **out) {
    static std::mutex mu;
    static std::vector<RdfEdgeTable> tables;
    int dev = 0;
    HTF_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    for (const RdfEdgeTable &t : tables)
        if (t.dev == dev && t.r0 == r0 && t.r1 == r1 && t.nb == nb) {
            *out = t.d;
            return HTF_OK;
        }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        set_error("compute_rdf range (%g, %g) / %u bins first seen during stream capture: run one eager step with these parameters first", r0, r1, nb);
        return HTF_ERR_INVALID;
    }
    std::vector<float> h((size_t)nb + 1, 0.f);
    h[0] = -1.f;
    h[nb] = __builtin_inff();
    for (unsigned b = 1; b < nb; ++b) {
        // smallest non-negative finite fp32 (by bit pattern) whose bin is >= b; none: +inf
        unsigned lo = 0u, hi = 0x7f800000u; // hi is never a candidate: answer in [lo, hi]
        while (lo < hi) {
            const unsigned mid = lo + (hi - lo) / 2u;
            float x;
            std::memcpy(&x, &mid, 4);
            if (rdf_bin_exact_host(x, r0, r1, nb) >= (float)b) hi = mid; else lo = mid + 1u;
        }
        std::memcpy(&h[b], &lo, 4);
    }
    RdfEdgeTable t{r0, r1, nb, dev, nullptr};
    HTF_CHECK_HIP(hipMalloc((void **)&t.d, ((size_t)nb + 1) * sizeof(float)));
    HTF_CHECK_HIP(hipMemcpy(t.d, h.data(), ((size_t)nb + 1) * sizeof(float), hipMemcpyHostToDevice)); // blocking: complete for every stream
    tables.push_back(t);
    *out = t.d;
    return HTF_OK;
}

// COMPACT (NN <= 128): the survivors of a row are first written to a wave-private LDS row at their final slot, then
// both potentials, the CV term and the histogram bin are evaluated slot by slot -- two trips over the ~95 survivors
// instead of three over the ~139 candidates -- and the tensor leaves as full-width lines.  This sweep is VALU-bound
// (two potentials, an exp, a correctly rounded sqrt and a histogram update per slot), unlike the LJ step, which is
// bound by its gathers and for which the same restructuring lost 2 % (see fused_rows_group).
template <int KA, bool STORE, bool COMPACT, typename PT>
__global__ __launch_bounds__(256) void fused_forces2_kernel(
    const typename Vec4<PT>::type *__restrict__ pos, unsigned N, unsigned NN, unsigned offset, unsigned batch,
    BoxT<PT> box, const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist,
    const unsigned *__restrict__ head_list, PT rmaxsq, void *__restrict__ forceA, void *__restrict__ forceB,
    int out_f64, PotParams pa_in, PotParams pb, float *__restrict__ partials, Rdf2 rdf,
    float4 *__restrict__ dest, unsigned *__restrict__ counts_io) {
    using PV = typename Vec4<PT>::type;
    const PotParams pa = resolve_theta<KA>(pa_in);
    // ONE shared object, the edge table first: its LDS address is then the constant 0 and a bin's two thresholds are read at
    // idx * 4 with the instruction's own offsets (separate __shared__ arrays cost an address add per slot)
    struct Lds {
        float edge[kRdfMaxBins2 + 4];
        unsigned hist[kRdfMaxBins2];
        float4 rows[COMPACT ? 4 * 128 : 1];
        float part[4];
    };
    __shared__ __attribute__((aligned(16))) Lds lds;
    float (&s_edge)[kRdfMaxBins2 + 4] = lds.edge;
    unsigned (&s_hist)[kRdfMaxBins2] = lds.hist;
    float (&s_part)[4] = lds.part;
    float4 *mine = lds.rows + (COMPACT ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 128 : 0);
    const bool do_rdf = rdf.hist != nullptr;
    if (do_rdf) {
        for (unsigned i = threadIdx.x; i < rdf.nb; i += blockDim.x) s_hist[i] = 0;
        for (unsigned i = threadIdx.x; i <= rdf.nb; i += blockDim.x) s_edge[i] = rdf.edges[i];
        __syncthreads();
    }
    // (wb as a SCALAR: a row's count, list head and own position are then scalar loads and the live-entry masks of its trips
    //  scalar arithmetic -- derived from threadIdx they were vector loads, and every trip built its mask with a dozen VALU
    //  instructions and two v_readfirstlane on the unit this kernel saturates)
    const unsigned lane = threadIdx.x & 63u, wb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool simple_box = box.ortho && box.periodic[0] && box.periodic[1] && box.periodic[2];
    const float rdf_scale = do_rdf ? (float)rdf.nb / (rdf.r1 - rdf.r0) : 0.f;
    const float rdf_bias = -rdf.r0 * rdf_scale; // (the guess may be off by one either way: the edge table settles it)
    // (through readfirstlane: a float compare leaves a LANE MASK, and branching on it cost a v_cndmask + v_cmp per slot)
    const bool coarse_bins = __builtin_amdgcn_readfirstlane((int)(do_rdf && (rdf.r1 - rdf.r0) >= 1e-4f * (float)rdf.nb)) != 0;
    // bin of a zero pair vector (the padded slots of the tensor are part of compute_rdf's input)
    int pad_bin = 0;
    if (do_rdf) {
        const float fi = floorf((float)rdf.nb * ((0.f - rdf.r0) / (rdf.r1 - rdf.r0)));
        pad_bin = fi < 0.f ? 0 : (fi > (float)(rdf.nb - 1) ? (int)(rdf.nb - 1) : (int)fi);
    }
    unsigned n_lo = 0, n_hi = 0;
    float cv_wave = 0.f; // sum of the B energy column over this wave's rows, in row order

    auto eval_slot = [&](float x, float y, float z, float &ax, float &ay, float &az, float &ae, float &bx, float &by,
                         float &bz, float &be) {
        float e, fx, fy, fz;
        const RinvFwd f = rinv_fwd(x, y, z); // t = x + 1e-7, r', 1 / r', s: once for both potentials
        pair_eval_f<KA>(f, x, y, z, pa, e, fx, fy, fz);
        ax += fx; ay += fy; az += fz; ae += e;
        pair_eval_f<HTF_POT_GAUSS>(f, x, y, z, pb, e, fx, fy, fz);
        bx += fx; by += fy; bz += fz; be += e;
        if (do_rdf) {
            // tf.histogram_fixed_width's bin from the squared norm and the threshold table (rdf_edges): the evaluator's r'
            // and a multiply give a guess that is off by at most one, two LDS words settle it -- no correctly
            // rounded sqrt, no division (33 M slots at C4 agree with the oracle bin for bin: test_full_size_c4_eds_sweep)
            const float sq = plain_sq3(x, y, z);
            float rg = f.rp; // r' is within 2e-7 of the plain norm: with bins no narrower than 1e-4 a guess off by at most one
            if (!coarse_bins) { // wave-uniform
                asm volatile("" ::: "memory");
                rg = __builtin_amdgcn_sqrtf(sq);
            }
            // guess = clamp(floor((r' - r0) scale)): one fma (r0 scale folded), v_cvt_flr_i32_f32 (floor and convert in one
            // instruction; out-of-range values saturate), one v_med3_i32; the two edge compares feed add-with-carry forms
            int idx;
            {
                const float qf = fmaf(rg, rdf_scale, rdf_bias);
                asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(idx) : "v"(qf));
                asm("v_med3_i32 %0, %1, 0, %2" : "=v"(idx) : "v"(idx), "s"((int)rdf.nb - 1));
            }
            // (byte offsets from here on: the corrections are then two selects of +-4 and one three-operand add)
            const unsigned off = (unsigned)idx << 2;
            const float e0 = *(const float *)((const char *)s_edge + off), e1 = *(const float *)((const char *)s_edge + off + 4);
            const unsigned off2 = off + (sq >= e1 ? 4u : 0u) + (sq < e0 ? (unsigned)-4 : 0u);
            atomicAdd((unsigned *)((char *)s_hist + off2), 1u); // live slots rarely land in the end bins (the padding, counted per row below, does)
        }
    };
    auto one = [&](PT dx, PT dy, PT dz, const PV &pk, bool keep, unsigned q, unsigned lo, unsigned Q, float4 *row, float &ax,
                   float &ay, float &az, float &ae, float &bx, float &by, float &bz, float &be) {
        if (keep && q >= lo) {
            const float x = (float)dx, y = (float)dy, z = (float)dz;
            if constexpr (STORE) {
                unsigned slot = q;
                if (Q > NN) { // wave-uniform
                    asm volatile("" ::: "memory");
                    slot = q % NN;
                }
                store_stream(&row[slot], make_float4(x, y, z, (float)scalar_as_int(pk.w)));
            }
            eval_slot(x, y, z, ax, ay, az, ae, bx, by, bz, be);
        }
    };

    const unsigned ngroups = (batch + 3) / 4;
    // The persistent loop is software-pipelined over its rows: while row i is evaluated, the metadata
    // (count, list head, own position) and then the first 192 list entries of the row this wave takes
    // NEXT are already on their way -- PMC had the waves of the unpipelined loop waiting on memory
    // 48 % of the time (three dependent loads at the head of every row, eight waves per SIMD).
    struct Meta {
        unsigned nn, head;
        PV pi;
        bool have;
    };
    auto fetch_meta = [&](unsigned grp) {
        Meta m;
        m.have = grp < ngroups && grp * 4 + wb < batch; // wave-uniform
        m.nn = 0;
        m.head = 0;
        m.pi = pos[offset]; // any valid element
        if (m.have) {
            const unsigned idx = grp * 4 + wb + offset;
            m.nn = n_neigh[idx];
            m.head = head_list[idx];
            m.pi = pos[idx];
        }
        return m;
    };
    auto fetch_idx = [&](const Meta &m, unsigned (&k)[kFChunk]) {
        const unsigned *nl = nlist + m.head;
#pragma unroll
        for (int t = 0; t < kFChunk; ++t) {
            const unsigned j = t * 64 + lane;
            k[t] = (m.have && m.nn != 0 && m.nn <= 64 * kFChunk) ? nl[j < m.nn ? j : m.nn - 1] : 0u;
        }
    };
    Meta cur = fetch_meta(blockIdx.x);
    unsigned k_cur[kFChunk];
    fetch_idx(cur, k_cur);
#pragma unroll 1
    for (unsigned grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const unsigned w = grp * 4 + wb;
        Meta nxt;
        unsigned k_nxt[kFChunk];
        if (!cur.have) { // wave-uniform: this wave has no row in the group (tail of the batch)
            nxt = fetch_meta(grp + gridDim.x);
            fetch_idx(nxt, k_nxt);
            cur = nxt;
#pragma unroll
            for (int t = 0; t < kFChunk; ++t) k_cur[t] = k_nxt[t];
            continue;
        }
        const unsigned nn = cur.nn;
        const unsigned *nl = nlist + cur.head;
        const PV pi = cur.pi;
        float4 *row = STORE ? dest + (size_t)w * NN : nullptr;
        float ax = 0.f, ay = 0.f, az = 0.f, ae = 0.f, bx = 0.f, by = 0.f, bz = 0.f, be = 0.f;
        unsigned Q = 0;
        if (nn <= 64 * kFChunk) {
            // pair vectors stay in registers between the two passes
            unsigned k[kFChunk];
            PV pk[kFChunk];
            unsigned q[kFChunk];
            bool keep[kFChunk];
            PT vx[kFChunk], vy[kFChunk], vz[kFChunk];
#pragma unroll
            for (int t = 0; t < kFChunk; ++t) k[t] = k_cur[t];
#pragma unroll
            for (int t = 0; t < kFChunk; ++t) pk[t] = pos[k[t]];
            nxt = fetch_meta(grp + gridDim.x); // in flight beside this row's gathers
#pragma unroll
            for (int t = 0; t < kFChunk; ++t) {
                PT rsq;
                if (simple_box) { // wave-uniform
                    asm volatile("" ::: "memory");
                    rsq = pair_vector_simple<PT>(pk[t], pi, box, vx[t], vy[t], vz[t]);
                } else {
                    rsq = pair_vector<PT>(pk[t], pi, box, vx[t], vy[t], vz[t]);
                }
                const unsigned left = nn > (unsigned)t * 64 ? nn - (unsigned)t * 64 : 0u;
                const unsigned long long m = ballot64(!(rsq > rmaxsq)) & (left >= 64u ? ~0ull : ((1ull << left) - 1ull));
                keep[t] = inverse_ballot64(m);
                q[t] = Q + ballot_rank(m);
                Q += __popcll(m);
            }
            fetch_idx(nxt, k_nxt); // in flight under this row's evaluation
            const unsigned lo = Q > NN ? Q - NN : 0u;
            if constexpr (COMPACT) {
#pragma unroll
                for (int t = 0; t < kFChunk; ++t) {
                    if ((unsigned)t * 64 >= nn) break; // wave-uniform
                    unsigned slot = q[t]; // the slot this survivor ends up in (overflow: the reference's wrap, an integer division
                    if (Q > NN) {         // kept behind a wave-uniform branch -- if-converted it ran for every trip)
                        asm volatile("" ::: "memory");
                        slot = q[t] % NN;
                    }
                    if (keep[t] && q[t] >= lo)
                        mine[slot] = make_float4((float)vx[t], (float)vy[t], (float)vz[t], (float)scalar_as_int(pk[t].w));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const unsigned live = Q < NN ? Q : NN;
                const unsigned prev = (STORE && counts_io != nullptr) ? counts_io[w] : (STORE ? NN : 0u);
                const unsigned end = live > prev ? live : prev;
                for (unsigned sl = lane; sl < end; sl += 64) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (sl < live) {
                        v = mine[sl];
                        eval_slot(v.x, v.y, v.z, ax, ay, az, ae, bx, by, bz, be);
                    }
                    if constexpr (STORE) store_stream(&row[sl], v);
                }
                __builtin_amdgcn_wave_barrier(); // the row is consumed before the next one overwrites it
            } else {
#pragma unroll
                for (int t = 0; t < kFChunk; ++t) {
                    if ((unsigned)t * 64 >= nn) break; // wave-uniform
                    one(vx[t], vy[t], vz[t], pk[t], keep[t], q[t], lo, Q, row, ax, ay, az, ae, bx, by, bz, be);
                }
            }
        } else {
            nxt = fetch_meta(grp + gridDim.x);
            fetch_idx(nxt, k_nxt);
            for (unsigned base = 0; base < nn; base += 64) { // counting pass
                const unsigned j = base + lane;
                const PV pk = pos[nl[j < nn ? j : nn - 1]];
                PT dx, dy, dz;
                const PT rsq = pair_vector<PT>(pk, pi, box, dx, dy, dz);
                Q += __popcll(__ballot((j < nn) && !(rsq > rmaxsq)));
            }
            const unsigned lo = Q > NN ? Q - NN : 0u;
            unsigned Q2 = 0;
            for (unsigned base = 0; base < nn; base += 64) { // evaluating pass
                const unsigned j = base + lane;
                const PV pk = pos[nl[j < nn ? j : nn - 1]];
                PT dx, dy, dz;
                const PT rsq = pair_vector<PT>(pk, pi, box, dx, dy, dz);
                const bool kp = (j < nn) && !(rsq > rmaxsq);
                const unsigned long long m = __ballot(kp);
                const unsigned qq = Q2 + ballot_rank(m);
                Q2 += __popcll(m);
                one(dx, dy, dz, pk, kp, qq, lo, Q, row, ax, ay, az, ae, bx, by, bz, be);
            }
        }
        const unsigned filled = Q < NN ? Q : NN;
        if constexpr (STORE) {
            const bool tail_done = COMPACT && nn <= 64 * kFChunk; // the compacting path stored live slots and tail together
            const unsigned zero_end = counts_io != nullptr ? counts_io[w] : NN;
            if (!tail_done)
                for (unsigned sl = filled + lane; sl < zero_end; sl += 64) store_stream(&row[sl], make_float4(0.f, 0.f, 0.f, 0.f));
            if (counts_io != nullptr && lane == 0) counts_io[w] = filled;
        }
        if (do_rdf && lane == 0 && filled < NN) { // the row's zero padding
            const unsigned npad = NN - filled;
            if (pad_bin == 0) n_lo += npad;
            else if (pad_bin == (int)rdf.nb - 1) n_hi += npad;
            else atomicAdd(&s_hist[pad_bin], npad);
        }
        // eight row sums in 20 instructions (64 one by one): rows 0..3 of each result hold x, z, y, e
        const float ta = wave_sum4(ax, ay, az, ae), tb = wave_sum4(bx, by, bz, be);
        if ((lane & 15u) == 0u) {
            const unsigned comp = ((lane >> 4) & 1u) * 2u + (lane >> 5); // 0, 2, 1, 3
            if (out_f64) {
                ((double *)forceA)[(size_t)w * 4 + comp] = (double)ta;
                ((double *)forceB)[(size_t)w * 4 + comp] = (double)tb;
            } else {
                ((float *)forceA)[(size_t)w * 4 + comp] = ta;
                ((float *)forceB)[(size_t)w * 4 + comp] = tb;
            }
        }
        cv_wave += __shfl(tb, 48); // the B energy sum sits in row 3
        cur = nxt;
#pragma unroll
        for (int t = 0; t < kFChunk; ++t) k_cur[t] = k_nxt[t];
    }
    if (partials != nullptr) { // one partial per block, fixed order -> deterministic
        if (lane == 0) s_part[wb] = cv_wave;
        __syncthreads();
        if (threadIdx.x == 0) partials[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    }
    if (do_rdf) {
        n_lo = group_sum_u<64>(n_lo);
        n_hi = group_sum_u<64>(n_hi);
        if (lane == 0) {
            if (n_lo) atomicAdd(&s_hist[0], n_lo);
            if (n_hi) atomicAdd(&s_hist[rdf.nb - 1], n_hi);
        }
        __syncthreads();
        for (unsigned i = threadIdx.x; i < rdf.nb; i += blockDim.x)
            if (s_hist[i]) atomicAdd(&rdf.hist[i], s_hist[i]);
    }
}

// ---------------------------------------------------------------------------- the C4 sweep, R rows per wave, tails merged
// Round 3.  The structure of fused_rows_group_tails (3.1) with the second potential, the CV row sums and the histogram added:
// the first 128 list entries of each of the wave's R rows as straight-line code, the R tails in ONE trip -- 9 trips per four
// rows, every one of them doing the pair vector, both potentials from one rinv_fwd and the bin of its kept candidates, where
// the compacting form walks 12 gather trips and 8 evaluation trips through an LDS row.  (Round 2 built this form once and it
// LOST at 128 VGPRs -- before the sweep's instruction diet; after it the per-trip code is ~85 VALU instructions and the
// accumulators are what the registers go to.)  Persistent workgroups as the compacting kernel (the LDS histogram is flushed
// once per workgroup); one CV partial per workgroup in wave order -> deterministic.  Rows the straight-line path does not
// cover (> 192 list entries, NN overflow, a non-orthorhombic or non-periodic box, an incomplete group) go through ONE
// generic single-row routine.
struct Sweep2 {
    PotParams pa, pb;
    float *edge;      // LDS: bin thresholds on the squared norm (address 0 of the block's LDS)
    unsigned *hist;   // LDS histogram
    float r0, scale, bias;
    int nb, pad_bin;
    bool do_rdf, coarse;
    unsigned n_lo, n_hi; // padded slots that fall into the first / last bin, counted in registers
    float cv_wave;       // sum of the B energy column over this wave's rows, in row order
};

__device__ __forceinline__ void sweep2_bin(const Sweep2 &st, float x, float y, float z, float rp, unsigned delta = 1u) {
    const float sq = plain_sq3(x, y, z);
    float rg = rp;
    if (!st.coarse) { // wave-uniform
        asm volatile("" ::: "memory");
        rg = __builtin_amdgcn_sqrtf(sq);
    }
    int idx;
    const float qf = fmaf(rg, st.scale, st.bias);
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(idx) : "v"(qf));
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(idx) : "v"(idx), "s"(st.nb - 1));
    const unsigned off = (unsigned)idx << 2;
    const float e0 = *(const float *)((const char *)st.edge + off), e1 = *(const float *)((const char *)st.edge + off + 4);
    const unsigned off2 = off + (sq >= e1 ? 4u : 0u) + (sq < e0 ? (unsigned)-4 : 0u);
    // One LDS add per live slot into ONE 102-bin table.  Round 4, measured and not kept (profiles/r04_c4_kernel_ab.txt): eight
    // lane-striped replicas per bin, summed at the workgroup's flush -- 169.2 / 166.9 us against 167.9 / 171.0 on the same box:
    // the collisions of a wave's 64 adds on ~60 addresses are not what this sweep waits for.
    atomicAdd((unsigned *)((char *)st.hist + off2), delta);
}

// both potentials of one candidate from ONE rinv_fwd; a dropped candidate is evaluated far out, where both vanish exactly
template <int KA>
__device__ __forceinline__ void sweep2_eval(const Sweep2 &st, bool keep, float x, float y, float z, float &ax, float &ay, float &az,
                                            float &ae, float &bx, float &by, float &bz, float &be) {
    static_assert(KA == HTF_POT_LJ || KA == HTF_POT_WCA || KA == HTF_POT_LJ_PARAM, "base potentials that vanish identically far out");
    const float xe = keep ? x : 1e18f;
    float e, fx, fy, fz;
    if constexpr (KA == HTF_POT_LJ_PARAM) {
        pair_eval_f<KA>(RinvFwd(), xe, y, z, st.pa, e, fx, fy, fz);
        ax += fx; ay += fy; az += fz; ae += e;
        const RinvFwd f = rinv_fwd(xe, y, z);
        pair_eval_f<HTF_POT_GAUSS>(f, xe, y, z, st.pb, e, fx, fy, fz);
        bx += fx; by += fy; bz += fz; be += e;
        if (st.do_rdf && keep) sweep2_bin(st, x, y, z, f.rp);
    } else {
        const RinvFwd f = rinv_fwd(xe, y, z);
        pair_eval_f<KA>(f, xe, y, z, st.pa, e, fx, fy, fz);
        ax += fx; ay += fy; az += fz; ae += e;
        pair_eval_f<HTF_POT_GAUSS>(f, xe, y, z, st.pb, e, fx, fy, fz);
        bx += fx; by += fy; bz += fz; be += e;
        if (st.do_rdf && keep) sweep2_bin(st, x, y, z, f.rp);
    }
}

__device__ __forceinline__ void sweep2_finish_row(Sweep2 &st, unsigned w, unsigned lane, unsigned NN, unsigned filled, float ax, float ay,
                                                  float az, float ae, float bx, float by, float bz, float be, void *__restrict__ forceA,
                                                  void *__restrict__ forceB, int out_f64) {
    if (st.do_rdf && lane == 0 && filled < NN) { // the row's zero padding is part of compute_rdf's input
        const unsigned npad = NN - filled;
        if (st.pad_bin == 0) st.n_lo += npad;
        else if (st.pad_bin == st.nb - 1) st.n_hi += npad;
        else atomicAdd(&st.hist[st.pad_bin], npad);
    }
    const float ta = wave_sum4(ax, ay, az, ae), tb = wave_sum4(bx, by, bz, be); // rows 0..3 of each: x, z, y, e
    if ((lane & 15u) == 0u) {
        const unsigned comp = ((lane >> 4) & 1u) * 2u + (lane >> 5); // 0, 2, 1, 3
        if (out_f64) {
            ((double *)forceA)[(size_t)w * 4 + comp] = (double)ta;
            ((double *)forceB)[(size_t)w * 4 + comp] = (double)tb;
        } else {
            ((float *)forceA)[(size_t)w * 4 + comp] = ta;
            ((float *)forceB)[(size_t)w * 4 + comp] = tb;
        }
    }
    st.cv_wave += __shfl(tb, 48);
}

// any row: counting pass, then the evaluating pass over the NN survivors the reference's slot wrap keeps (fused_forces2_kernel's
// long-row path as a routine)
template <int KA, bool STORE, typename PT>
__device__ __forceinline__ void sweep2_row_generic(Sweep2 &st, const unsigned w, const unsigned lane,
                                                const typename Vec4<PT>::type *__restrict__ pos, unsigned NN, unsigned offset,
                                                const BoxT<PT> &box, const unsigned *__restrict__ n_neigh,
                                                const unsigned *__restrict__ nlist, const unsigned *__restrict__ head_list, PT rmaxsq,
                                                void *__restrict__ forceA, void *__restrict__ forceB, int out_f64,
                                                float4 *__restrict__ dest, unsigned *__restrict__ counts_io) {
    using PV = typename Vec4<PT>::type;
    const unsigned idx = w + offset;
    const unsigned nn = n_neigh[idx];
    const unsigned *nl = nlist + head_list[idx];
    const PV pi = pos[idx];
    float4 *row = STORE ? dest + (size_t)w * NN : nullptr;
    unsigned Q = 0;
    for (unsigned base = 0; base < nn; base += 64) {
        const unsigned j = base + lane;
        const PV pk = pos[nl[j < nn ? j : nn - 1]];
        PT dx, dy, dz;
        const PT rsq = pair_vector<PT>(pk, pi, box, dx, dy, dz);
        Q += __popcll(__ballot((j < nn) && !(rsq > rmaxsq)));
    }
    const unsigned lo = Q > NN ? Q - NN : 0u;
    float ax = 0.f, ay = 0.f, az = 0.f, ae = 0.f, bx = 0.f, by = 0.f, bz = 0.f, be = 0.f;
    unsigned Q2 = 0;
    for (unsigned base = 0; base < nn; base += 64) {
        const unsigned j = base + lane;
        const PV pk = pos[nl[j < nn ? j : nn - 1]];
        PT dx, dy, dz;
        const PT rsq = pair_vector<PT>(pk, pi, box, dx, dy, dz);
        const bool kp = (j < nn) && !(rsq > rmaxsq);
        const unsigned long long m = __ballot(kp);
        const unsigned qq = Q2 + ballot_rank(m);
        Q2 += __popcll(m);
        const bool use = kp && qq >= lo;
        const float x = (float)dx, y = (float)dy, z = (float)dz;
        if constexpr (STORE)
            if (use) store_stream(&row[qq % NN], make_float4(x, y, z, (float)scalar_as_int(pk.w)));
        sweep2_eval<KA>(st, use, x, y, z, ax, ay, az, ae, bx, by, bz, be);
    }
    const unsigned filled = Q < NN ? Q : NN;
    if constexpr (STORE) {
        const unsigned zero_end = counts_io != nullptr ? counts_io[w] : NN;
        for (unsigned sl = filled + lane; sl < zero_end; sl += 64) store_stream(&row[sl], make_float4(0.f, 0.f, 0.f, 0.f));
        if (counts_io != nullptr && lane == 0) counts_io[w] = filled;
    }
    sweep2_finish_row(st, w, lane, NN, filled, ax, ay, az, ae, bx, by, bz, be, forceA, forceB, out_f64);
}

template <int KA, bool STORE, int R, typename PT>
__device__ __forceinline__ void sweep2_rows_group_tails(Sweep2 &st, const unsigned w0, const unsigned lane,
                                                        const typename Vec4<PT>::type *__restrict__ pos, unsigned NN, unsigned offset,
                                                        unsigned batch, const BoxT<PT> &box, bool simple_box,
                                                        const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist,
                                                        const unsigned *__restrict__ head_list, PT rmaxsq, void *__restrict__ forceA,
                                                        void *__restrict__ forceB, int out_f64, float4 *__restrict__ dest,
                                                        unsigned *__restrict__ counts_io) {
    using PV = typename Vec4<PT>::type;
    unsigned nn[R], S[R + 1];
    bool fast = w0 + R <= batch && simple_box;
    S[0] = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        nn[r] = n_neigh[(w0 + r < batch ? w0 + r : w0) + offset];
        fast = fast && nn[r] != 0 && nn[r] <= 192u;
        S[r + 1] = S[r] + (nn[r] > 128u ? nn[r] - 128u : 0u);
    }
    if (!fast || S[R] > 64u) { // (wave-uniform)
#pragma unroll 1
        for (unsigned r = 0; r < (unsigned)R; ++r)
            if (w0 + r < batch)
                sweep2_row_generic<KA, STORE, PT>(st, w0 + r, lane, pos, NN, offset, box, n_neigh, nlist, head_list, rmaxsq, forceA, forceB,
                                                  out_f64, dest, counts_io);
        return;
    }
    PV pi[R];
    unsigned head[R];
    unsigned k[R][2], kt;
    PV q[R][2], qt;
    unsigned rl = 0;
#pragma unroll
    for (int r = 1; r < R; ++r) rl += lane >= S[r] ? 1u : 0u;
    unsigned head_l = 0, s_l = 0, nn_l = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        head[r] = head_list[w0 + r + offset];
        pi[r] = pos[w0 + r + offset];
        const unsigned *nl = nlist + head[r];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned j = t * 64 + lane;
            k[r][t] = nl[min(j, nn[r] - 1u)];
        }
        head_l = rl == (unsigned)r ? head[r] : head_l;
        s_l = rl == (unsigned)r ? S[r] : s_l;
        nn_l = rl == (unsigned)r ? nn[r] : nn_l;
    }
    const bool tail_live = lane < S[R];
    const unsigned jt = 128u + (lane - s_l);
    kt = nlist[head_l + (tail_live ? jt : nn_l - 1)];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t) q[r][t] = load_neighbor(pos, k[r][t]);
    qt = load_neighbor(pos, kt);
    float ax[R], ay[R], az[R], ae[R], bx[R], by[R], bz[R], be[R];
    unsigned Q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float4 *row = STORE ? dest + (size_t)(w0 + r) * NN : nullptr;
        ax[r] = ay[r] = az[r] = ae[r] = bx[r] = by[r] = bz[r] = be[r] = 0.f;
        Q[r] = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if ((unsigned)t * 64 >= nn[r]) break; // wave-uniform
            const unsigned left = nn[r] - (unsigned)t * 64;
            const unsigned long long valid = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
            const PV pk = q[r][t];
            PT dx, dy, dz;
            const PT rsq = pair_vector_simple<PT>(pk, pi[r], box, dx, dy, dz);
            const unsigned long long m = ballot64(!(rsq > rmaxsq)) & valid;
            const unsigned qq = Q[r] + ballot_rank(m);
            Q[r] += __popcll(m);
            const bool keep = inverse_ballot64(m);
            const float x = (float)dx, y = (float)dy, z = (float)dz;
            if constexpr (STORE) {
                unsigned long long ms = m;
                if (Q[r] > NN) { // (wave-uniform) about to overflow: slots bounded lane by lane; the row is redone below
                    asm volatile("" ::: "memory");
                    ms &= ballot64(qq < NN);
                }
                if (inverse_ballot64(ms)) store_stream(&row[qq], make_float4(x, y, z, (float)scalar_as_int(pk.w)));
            }
            sweep2_eval<KA>(st, keep, x, y, z, ax[r], ay[r], az[r], ae[r], bx[r], by[r], bz[r], be[r]);
        }
    }
    if (S[R] != 0) { // the shared tail trip
        PV pil = pi[0];
#pragma unroll
        for (int r = 1; r < R; ++r) {
            pil.x = rl == (unsigned)r ? pi[r].x : pil.x;
            pil.y = rl == (unsigned)r ? pi[r].y : pil.y;
            pil.z = rl == (unsigned)r ? pi[r].z : pil.z;
        }
        PT dx, dy, dz;
        const PT rsq = pair_vector_simple<PT>(qt, pil, box, dx, dy, dz);
        const unsigned long long m = ballot64(!(rsq > rmaxsq)) & (S[R] >= 64u ? ~0ull : ((1ull << S[R]) - 1ull));
        unsigned base_l = Q[0];
#pragma unroll
        for (int r = 1; r < R; ++r) {
            const unsigned before = (unsigned)__popcll(S[r] >= 64u ? m : (m & ((1ull << S[r]) - 1ull)));
            base_l = rl == (unsigned)r ? Q[r] - before : base_l;
        }
        const unsigned qq = base_l + ballot_rank(m);
        const bool keep = inverse_ballot64(m);
        const float xt = (float)dx, yt = (float)dy, zt = (float)dz;
        if constexpr (STORE)
            if (keep && qq < NN) store_stream(dest + (size_t)(w0 + rl) * NN + qq, make_float4(xt, yt, zt, (float)scalar_as_int(qt.w)));
        float tax = 0.f, tay = 0.f, taz = 0.f, tae = 0.f, tbx = 0.f, tby = 0.f, tbz = 0.f, tbe = 0.f;
        sweep2_eval<KA>(st, keep, xt, yt, zt, tax, tay, taz, tae, tbx, tby, tbz, tbe);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool mine = rl == (unsigned)r;
            ax[r] += mine ? tax : 0.f; ay[r] += mine ? tay : 0.f; az[r] += mine ? taz : 0.f; ae[r] += mine ? tae : 0.f;
            bx[r] += mine ? tbx : 0.f; by[r] += mine ? tby : 0.f; bz[r] += mine ? tbz : 0.f; be[r] += mine ? tbe : 0.f;
            const unsigned long long seg = (S[r + 1] >= 64u ? ~0ull : ((1ull << S[r + 1]) - 1ull)) & ~(S[r] >= 64u ? ~0ull : ((1ull << S[r]) - 1ull));
            Q[r] += (unsigned)__popcll(m & seg);
        }
    }
    unsigned redo = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned w = w0 + r;
        const unsigned filled = Q[r] < NN ? Q[r] : NN;
        if (Q[r] > NN) { // overflow (an error upstream): the generic routine reproduces the slot wrap -- and its histogram
            redo |= 1u << r;
            continue;
        }
        if constexpr (STORE) {
            float4 *row = dest + (size_t)w * NN;
            const unsigned zero_end = counts_io != nullptr ? counts_io[w] : NN;
            for (unsigned sl = filled + lane; sl < zero_end; sl += 64) store_stream(&row[sl], make_float4(0.f, 0.f, 0.f, 0.f));
            if (counts_io != nullptr && lane == 0) counts_io[w] = filled;
        }
        sweep2_finish_row(st, w, lane, NN, filled, ax[r], ay[r], az[r], ae[r], bx[r], by[r], bz[r], be[r], forceA, forceB, out_f64);
    }
    if (redo != 0) {
#pragma unroll 1
        for (unsigned r = 0; r < (unsigned)R; ++r)
            if ((redo >> r) & 1u) {
                if constexpr (STORE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (st.do_rdf) {
                    // the straight-line path has binned EVERY kept candidate of this row before its overflow was known: take them
                    // out again (the generic routine then bins the NN survivors of the slot wrap)
                    const unsigned idx = w0 + r + offset;
                    const unsigned nnr = n_neigh[idx];
                    const unsigned *nl = nlist + head_list[idx];
                    const PV pc = pos[idx];
                    for (unsigned base = 0; base < nnr; base += 64) {
                        const unsigned j = base + lane;
                        const PV pk = pos[nl[j < nnr ? j : nnr - 1]];
                        PT dx, dy, dz;
                        const PT rsq = pair_vector_simple<PT>(pk, pc, box, dx, dy, dz);
                        if ((j < nnr) && !(rsq > rmaxsq)) {
                            const float x = (float)dx, y = (float)dy, z = (float)dz;
                            sweep2_bin(st, x, y, z, rinv_fwd(x, y, z).rp, 0xFFFFFFFFu);
                        }
                    }
                }
                sweep2_row_generic<KA, STORE, PT>(st, w0 + r, lane, pos, NN, offset, box, n_neigh, nlist, head_list, rmaxsq, forceA, forceB,
                                                  out_f64, dest, counts_io);
            }
    }
}

template <int KA, bool STORE, int R, typename PT>
__global__ __launch_bounds__(256) void fused_forces2_tails_kernel(
    const typename Vec4<PT>::type *__restrict__ pos, unsigned N, unsigned NN, unsigned offset, unsigned batch,
    BoxT<PT> box, const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist,
    const unsigned *__restrict__ head_list, PT rmaxsq, void *__restrict__ forceA, void *__restrict__ forceB,
    int out_f64, PotParams pa_in, PotParams pb, float *__restrict__ partials, Rdf2 rdf,
    float4 *__restrict__ dest, unsigned *__restrict__ counts_io) {
    struct Lds {
        float edge[kRdfMaxBins2 + 4];
        unsigned hist[kRdfMaxBins2];
        float part[4];
    };
    __shared__ __attribute__((aligned(16))) Lds lds;
    Sweep2 st;
    st.pa = resolve_theta<KA>(pa_in);
    st.pb = pb;
    st.edge = lds.edge;
    st.hist = lds.hist;
    st.do_rdf = rdf.hist != nullptr;
    if (st.do_rdf) {
        for (unsigned i = threadIdx.x; i < rdf.nb; i += blockDim.x) lds.hist[i] = 0;
        for (unsigned i = threadIdx.x; i <= rdf.nb; i += blockDim.x) lds.edge[i] = rdf.edges[i];
        __syncthreads();
    }
    st.r0 = rdf.r0;
    st.nb = (int)rdf.nb;
    st.scale = st.do_rdf ? (float)rdf.nb / (rdf.r1 - rdf.r0) : 0.f;
    st.bias = -rdf.r0 * st.scale;
    // (through readfirstlane: a float compare leaves a LANE MASK, and branching on it cost a v_cndmask + v_cmp per slot)
    st.coarse = __builtin_amdgcn_readfirstlane((int)(st.do_rdf && (rdf.r1 - rdf.r0) >= 1e-4f * (float)rdf.nb)) != 0;
    st.pad_bin = 0;
    if (st.do_rdf) {
        const float fi = floorf((float)rdf.nb * ((0.f - rdf.r0) / (rdf.r1 - rdf.r0)));
        st.pad_bin = fi < 0.f ? 0 : (fi > (float)(rdf.nb - 1) ? (int)(rdf.nb - 1) : (int)fi);
    }
    st.n_lo = st.n_hi = 0;
    st.cv_wave = 0.f;
    const unsigned lane = threadIdx.x & 63u, wb = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool simple_box = box.ortho && box.periodic[0] && box.periodic[1] && box.periodic[2];
    const unsigned ngroups = (batch + R - 1) / R;
    const unsigned wv = blockIdx.x * 4 + wb, nw = gridDim.x * 4;
#pragma unroll 1
    for (unsigned g = wv; g < ngroups; g += nw)
        sweep2_rows_group_tails<KA, STORE, R, PT>(st, R * g, lane, pos, NN, offset, batch, box, simple_box, n_neigh, nlist, head_list,
                                                  rmaxsq, forceA, forceB, out_f64, dest, counts_io);
    if (partials != nullptr) { // one partial per workgroup, waves in a fixed order -> deterministic
        if (lane == 0) lds.part[wb] = st.cv_wave;
        __syncthreads();
        if (threadIdx.x == 0) partials[blockIdx.x] = (lds.part[0] + lds.part[1]) + (lds.part[2] + lds.part[3]);
    }
    if (st.do_rdf) {
        const unsigned n_lo = group_sum_u<64>(st.n_lo), n_hi = group_sum_u<64>(st.n_hi);
        if (lane == 0) {
            if (n_lo) atomicAdd(&lds.hist[0], n_lo);
            if (n_hi) atomicAdd(&lds.hist[rdf.nb - 1], n_hi);
        }
        __syncthreads();
        for (unsigned i = threadIdx.x; i < rdf.nb; i += blockDim.x)
            if (lds.hist[i]) atomicAdd(&rdf.hist[i], lds.hist[i]);
    }
}

// (Measured and removed, commit 17bc8a6: this sweep in the four-rows-per-wave form of fused_rows_group_tails -- all pair vectors
//  and ballots of a group first, so that an overflowing row is known before anything is stored or binned, then 9 evaluation
//  trips per four rows instead of 12 and no LDS round trip; same tensor, histogram and forces (the 16 C4 / EDS / RDF parity
//  tests pass).  Eight accumulators for each of four rows beside eight gathered positions: 128 VGPRs, 4 waves per SIMD,
//  and 229.9 us against 206.2 on the same box (tensor-less and without the histogram: 155.5 against 150.9).)
unsigned fused_forces2_num_partials(unsigned batch) {
    const unsigned ngroups = (batch + 3) / 4;
    // persistent workgroups: 4096 = 16 per CU, about twice what is resident -- measured at C4
    // 1024 / 2048 / 4096 / 8192 / 16384 workgroups: 244 / 220 / 204 / 212 / 372 us (every workgroup ends with a
    // flush of its LDS histogram); the rows-per-wave form, round 3: 1536 / 2560 / 3072 / 4096 / 8192 -> 165 / 165 / 166 /
    // 169 / 203 us (1792 and 2048, one more workgroup per CU than is resident: 191 / 177) -- flat around the default.
    // (Also measured and not kept, profiles/r03_c4_kernel_ab.txt: the group loop as a software pipeline -- scalars, list
    //  entries and gathers of the next one or two groups in flight; 85 / 108 VGPRs, 175 / 187 us against 167.)
#ifdef HTF_AB_VARIANTS
    static const char *env = getenv("HTF_FUSED2_GRID");
    const unsigned cap = env ? (unsigned)atoi(env) : 4096u;
#else
    constexpr unsigned cap = 4096u;
#endif
    return ngroups < cap ? ngroups : cap;
}

template <int KA, typename PT>
static int launch_fused2(const PotParams &pa, const PotParams &pb, const void *pos, unsigned N, unsigned NN, unsigned offset,
                         unsigned batch, const htf_box *hb, const unsigned *n_neigh, const unsigned *nlist,
                         const unsigned *head_list, double rmax, void *fa, void *fb, int out_f64, float *partials,
                         const Rdf2 &rdf, float4 *dest, unsigned *counts_io, hipStream_t s) {
    BoxT<PT> b = make_boxt<PT>(hb);
    PT rc = (PT)rmax;
    const unsigned grid = fused_forces2_num_partials(batch);
    // Shipped forms: two rows per wave with merged tails (fp32 positions and the base potentials that vanish far out), the
    // compacting persistent kernel for everything else.  Variants builds: HTF_FUSED2_ROWS = 4 | 2 | 0, HTF_FUSED2_COMPACT = 0
    // (evaluate the candidates in place).
#ifdef HTF_AB_VARIANTS
    static const char *cenv = getenv("HTF_FUSED2_COMPACT");
    const bool compact = NN <= 128 && (cenv ? atoi(cenv) != 0 : true);
    static const char *renv = getenv("HTF_FUSED2_ROWS");
    const int rows = renv ? atoi(renv) : HTF_FUSED2_ROWS_DEFAULT;
#else
    constexpr int rows = HTF_FUSED2_ROWS_DEFAULT;
#endif
    if constexpr (KA == HTF_POT_LJ || KA == HTF_POT_WCA || KA == HTF_POT_LJ_PARAM) {
        if (rows == 2 || rows == 4) {
#define HTF_F2T_LAUNCH(ST, RR)                                                                                         \
    hipLaunchKernelGGL((fused_forces2_tails_kernel<KA, ST, RR, PT>), dim3(grid), dim3(256), 0, s,                      \
                       (const typename Vec4<PT>::type *)pos, N, NN, offset, batch, b, n_neigh, nlist, head_list,       \
                       (PT)(rc * rc), fa, fb, out_f64, pa, pb, partials, rdf, dest, counts_io)
#ifdef HTF_AB_VARIANTS
            if (dest != nullptr) {
                if (rows == 2) HTF_F2T_LAUNCH(true, 2); else HTF_F2T_LAUNCH(true, 4);
            } else {
                if (rows == 2) HTF_F2T_LAUNCH(false, 2); else HTF_F2T_LAUNCH(false, 4);
            }
#else
            if (dest != nullptr) HTF_F2T_LAUNCH(true, HTF_FUSED2_ROWS_DEFAULT); else HTF_F2T_LAUNCH(false, HTF_FUSED2_ROWS_DEFAULT);
#endif
#undef HTF_F2T_LAUNCH
            return check_launch("fused_forces2_tails_kernel");
        }
    }
#define HTF_F2_LAUNCH(ST, CP)                                                                                          \
    hipLaunchKernelGGL((fused_forces2_kernel<KA, ST, CP, PT>), dim3(grid), dim3(256), 0, s,                            \
                       (const typename Vec4<PT>::type *)pos, N, NN, offset, batch, b, n_neigh, nlist, head_list,       \
                       (PT)(rc * rc), fa, fb, out_f64, pa, pb, partials, rdf, dest, counts_io)
#ifdef HTF_AB_VARIANTS
    if (dest != nullptr) {
        if (compact) HTF_F2_LAUNCH(true, true); else HTF_F2_LAUNCH(true, false);
    } else {
        if (compact) HTF_F2_LAUNCH(false, true); else HTF_F2_LAUNCH(false, false);
    }
#else
    // (rows longer than the compaction buffer: NN <= 128 is what the compacting form holds; beyond it the in-place form)
    if (NN <= 128) {
        if (dest != nullptr) HTF_F2_LAUNCH(true, true); else HTF_F2_LAUNCH(false, true);
    } else {
        if (dest != nullptr) HTF_F2_LAUNCH(true, false); else HTF_F2_LAUNCH(false, false);
    }
#endif
#undef HTF_F2_LAUNCH
    return check_launch("fused_forces2_kernel");
}

int fused_forces2_impl(const PotParams &pa, const PotParams &pb, const void *pos, int pos_dtype, unsigned N, unsigned NN,
                       unsigned offset, unsigned batch, const htf_box *box, const unsigned *n_neigh,
                       const unsigned *nlist, const unsigned *head_list, double rmax, void *fa, void *fb,
                       int force_dtype, float *partials, float rdf_r0, float rdf_r1, unsigned rdf_nb,
                       unsigned *rdf_hist, float4 *dest, unsigned *counts_io, hipStream_t s) {
    HTF_REQUIRE(pos && n_neigh && nlist && head_list && box && fa && fb, "htf_build_eval_forces2: null pointer");
    HTF_REQUIRE(NN > 0 && rmax > 0, "htf_build_eval_forces2: NN and rmax must be > 0");
    HTF_REQUIRE(offset <= N && batch <= N - offset, "htf_build_eval_forces2: batch [%u, %u) exceeds N=%u", offset, offset + batch, N);
    HTF_REQUIRE(pos_dtype == HTF_F32 || pos_dtype == HTF_F64, "htf_build_eval_forces2: bad position dtype %d", pos_dtype);
    HTF_REQUIRE(pb.kind == HTF_POT_GAUSS, "htf_build_eval_forces2: potB must be HTF_POT_GAUSS");
    if (rdf_hist != nullptr)
        HTF_REQUIRE(rdf_nb >= 3 && rdf_nb <= kRdfMaxBins2 && rdf_r1 > rdf_r0, "htf_build_eval_forces2: need 3 <= bins <= %u and r1 > r0", kRdfMaxBins2);
    if (batch == 0) return HTF_OK;
    const float *edges = nullptr;
    if (rdf_hist != nullptr) {
        const int rc_e = rdf_edges(rdf_r0, rdf_r1, rdf_nb, s, &edges);
        if (rc_e != HTF_OK) return rc_e;
    }
    const Rdf2 rdf{rdf_r0, rdf_r1, rdf_nb, rdf_hist, edges};
    const int out_f64 = force_dtype == HTF_F64;
#define HTF_F2(K)                                                                                                      \
    (pos_dtype == HTF_F32 ? launch_fused2<K, float>(pa, pb, pos, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmax, fa, fb, out_f64, partials, rdf, dest, counts_io, s) \
                          : launch_fused2<K, double>(pa, pb, pos, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmax, fa, fb, out_f64, partials, rdf, dest, counts_io, s))
    switch (pa.kind) {
    case HTF_POT_LJ: return HTF_F2(HTF_POT_LJ);
    case HTF_POT_WCA: return HTF_F2(HTF_POT_WCA);
    case HTF_POT_RINV_POLY: return HTF_F2(HTF_POT_RINV_POLY);
    case HTF_POT_LJ_PARAM: return HTF_F2(HTF_POT_LJ_PARAM);
    default:
        set_error("htf_build_eval_forces2: potential kind %d is not a closed-form base potential", pa.kind);
        return HTF_ERR_INVALID;
    }
#undef HTF_F2
}

int fused_forces_impl(const PotParams &p, const void *pos, int pos_dtype, unsigned N, unsigned NN, unsigned offset,
                      unsigned batch, const htf_box *box, const unsigned *n_neigh, const unsigned *nlist,
                      const unsigned *head_list, double rmax, void *force, int force_dtype, void *virial9,
                      unsigned *check_count, float4 *positions_out, float4 *dest, unsigned *counts_io, hipStream_t s) {
    HTF_REQUIRE(pos && n_neigh && nlist && head_list && box && force, "htf_fused_forces: null pointer");
    HTF_REQUIRE(NN > 0 && rmax > 0, "htf_fused_forces: NN and rmax must be > 0");
    HTF_REQUIRE(offset <= N && batch <= N - offset, "htf_fused_forces: batch [%u, %u) exceeds N=%u", offset, offset + batch, N);
    HTF_REQUIRE(pos_dtype == HTF_F32 || pos_dtype == HTF_F64, "htf_fused_forces: bad position dtype %d", pos_dtype);
    if (batch == 0) return HTF_OK;
    const int out_f64 = force_dtype == HTF_F64;
#define HTF_FK(K) launch_fused_k<K>(pos, pos_dtype, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmax, force, virial9, out_f64, p, check_count, positions_out, dest, counts_io, s)
    switch (p.kind) {
    case HTF_POT_LJ: return HTF_FK(HTF_POT_LJ);
    case HTF_POT_WCA: return HTF_FK(HTF_POT_WCA);
    case HTF_POT_RINV_POLY: return HTF_FK(HTF_POT_RINV_POLY);
    case HTF_POT_GAUSS: return HTF_FK(HTF_POT_GAUSS);
    case HTF_POT_LJ_PARAM: return HTF_FK(HTF_POT_LJ_PARAM);
    case HTF_POT_SIMPLE:
        HTF_REQUIRE(virial9 == nullptr, "htf_fused_forces: SimplePotential has no virial");
        return HTF_FK(HTF_POT_SIMPLE);
    case HTF_POT_JIT: // a generated unit's instantiations of the kernels above (csrc/jit.hip)
        return jit_launch_fused(p, pos, pos_dtype, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmax, force, virial9, out_f64,
                                check_count, positions_out, dest, counts_io, s);
    default:
        set_error("htf_fused_forces: potential kind %d has no fused form (the pair-MLP is MFMA-bound, not traffic-bound)", p.kind);
        return HTF_ERR_INVALID;
    }
#undef HTF_FK
}

} // namespace htf
#endif // HTF_JIT_UNIT
