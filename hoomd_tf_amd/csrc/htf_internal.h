// Cross-translation-unit declarations (internal; not part of the ABI).
#pragma once
#include "htf_common.h"

namespace htf {

struct PotParams {
    int kind;
    float sigma;   // WCA sigma (fp32, as the Keras weight is)
    float wca_cut; // fp32(sigma) * fp32(2^(1/3))   layers.py:97
    float wca_cut_r2; // smallest fp32 t with sqrtf(t) >= wca_cut: (r2 < t) == (sqrtf(r2) < wca_cut), exactly
    float gauss_r0, gauss_ginv, gauss_coef; // HTF_POT_GAUSS: centre, 1/gap, coefficient
    float gauss_k_exp, gauss_k_force;       // -log2(e) / gap and -4 coef / gap: the exponent and force chains as one multiply each
    float lj_w0, lj_w1;                     // HTF_POT_LJ_PARAM
    const float *theta;                     // device parameter vector of a trainable potential (nullable)
    int n_terms;
    float coef[HTF_MAX_POLY_TERMS];
    int power[HTF_MAX_POLY_TERMS];
    float poly_cut_r2; // RINV_POLY mask: smallest fp32 t with sqrtf(t) >= poly_cut (0: no mask)
    const struct JitKernels *jit; // HTF_POT_JIT: the loaded code object's kernels (csrc/jit.hip); host-side only
    int jit_flags;                // HTF_JIT_READS_OWN_TYPE
    const void *own;              // streaming evaluator of a generated unit: the rows' positions ([B] x 4, w = type as a float) or null
    int own_f64;
};

// The stand-in integrator as the EPILOGUE of the one-kernel step (round 6; include/htf_standin.h htfs_step_epilogue): the lanes that
// hold a finished row's force components go on with v += f dt, x(t + dt) = wrap(x + v dt) -- htfs_nve_step's arithmetic,
// component by component -- written into the OTHER position buffer (every wave still reads x(t) of everybody), and, under a brick
// decomposition, into the halo messages that carry the row.  A step is then one launch where it was force kernel + integrator (+
// halo pack).  vel == nullptr: no epilogue.  All fields by value: a kernel argument.
// the epilogue under a HOOMD DOUBLE build: measured twice in round 6 and OFF -- with the first form of the epilogue 9.7 k against
// 11.1 k steps/s at C3, with the one-copy form (83 VGPRs against 79, the same six waves per SIMD) 10.55-10.59 k against 10.91-10.95 k:
// the fp64 four-row kernel takes 74-77 us with it and 65-66 without.  -DHTF_EPILOGUE_F64=1 (context.hip and fused_eval.hip) compiles it.
#ifndef HTF_EPILOGUE_F64
#define HTF_EPILOGUE_F64 0
#endif
template <typename T>
struct StepEpilogue {
    void *vel = nullptr;      // Scalar4[N], updated in place
    void *pos_next = nullptr; // Scalar4[N + ghosts]
    T dt = 0;
    T lo[3], L[3], Linv[3];   // the integrator's box (standin_gate.h SBox: the same three expressions)
    int periodic[3];
    // halo messages of the new positions (BrickDomain): row_slots[j][m] = slot of boundary row j in message m, 0xFFFFFFFF = none
    const unsigned *row_slots = nullptr;
    void *send = nullptr, *direct = nullptr; // Scalar4 rows: the send buffer / the ghost region of pos_next (own neighbor)
    unsigned cap_int = 0;
    int n_msg = 0, halo_wrap = 0;
    unsigned ghost_off[8], ghost_off_opp[8]; // (message m lands at ghost_off[n_msg - 1 - m] of a rank that is its own neighbor)
    T shift[8][3];
};

// `comp` (0..3) is the component this lane holds in `tot` (wave_sum4's lanes 0 / 16 / 32 / 48); `own` = x, y, z or w of the row's
// own position.  standin_gate.h nve_advance + wrap1 and brick.hip shifted(), one component at a time: the same bits.
// ``v``: the row's velocity component, loaded by the caller -- EARLY, at the head of the wave's work, by the row-group forms: a
// load behind the row's last instruction put a trip to memory at the end of every wave's critical path (+4.6 us on an 18.8 us
// kernel at 32 768 rows, +14 at 131 072: more than the integrator launch it replaces).  -> x(t + dt) of this component.
template <typename T>
__device__ __forceinline__ T step_epilogue_core(const StepEpilogue<T> &ep, unsigned idx, unsigned comp, float tot, T own, T v) {
    T *vel = reinterpret_cast<T *>(ep.vel) + (size_t)idx * 4;
    T *pn = reinterpret_cast<T *>(ep.pos_next) + (size_t)idx * 4;
    T xn = own; // (the type word travels as it is)
    if (comp < 3u) {
        const T lo = comp == 0u ? ep.lo[0] : (comp == 1u ? ep.lo[1] : ep.lo[2]);
        const T L = comp == 0u ? ep.L[0] : (comp == 1u ? ep.L[1] : ep.L[2]);
        const T Linv = comp == 0u ? ep.Linv[0] : (comp == 1u ? ep.Linv[1] : ep.Linv[2]);
        const int per = comp == 0u ? ep.periodic[0] : (comp == 1u ? ep.periodic[1] : ep.periodic[2]);
        const T f = (T)tot;
        v += ep.dt * f;
        xn = own + ep.dt * v;
        if (per) {
            const T fl = floor((xn - lo) * Linv);
            xn = xn - fl * L;
        }
        vel[comp] = v;
    }
    pn[comp] = xn;
    return xn;
}

// the new position into the halo messages that carry the row: slot `slot` of message M (0xFFFFFFFF: not in it), shifted as brick.hip
// shifted() shifts it (a wrap back into the global box -- halo_wrap, a replica brick on the global grid -- is not an epilogue's:
// htfs_set_step_epilogue declines)
template <typename T, int M>
__device__ __forceinline__ void step_epilogue_msg(const StepEpilogue<T> &ep, unsigned slot, unsigned comp, T xn) {
    if (slot == 0xFFFFFFFFu) return;
    T q = xn;
    if (comp < 3u) {
        const T sh = comp == 0u ? ep.shift[M][0] : (comp == 1u ? ep.shift[M][1] : ep.shift[M][2]);
        if (sh != (T)0) q = xn + sh;
    }
    if (ep.send != nullptr) reinterpret_cast<T *>(ep.send)[(size_t)(ep.ghost_off[M] + slot) * 4 + comp] = q;
    if (ep.direct != nullptr) reinterpret_cast<T *>(ep.direct)[(size_t)(ep.ghost_off_opp[M] + slot) * 4 + comp] = q;
}
template <typename T>
__device__ __forceinline__ void step_epilogue_halo(const StepEpilogue<T> &ep, const uint4 &s0, const uint4 &s1, unsigned comp, T xn) {
    step_epilogue_msg<T, 0>(ep, s0.x, comp, xn);
    step_epilogue_msg<T, 1>(ep, s0.y, comp, xn);
    if (ep.n_msg > 2) { // (wave-uniform: a 2-D brick's six further messages)
        step_epilogue_msg<T, 2>(ep, s0.z, comp, xn);
        step_epilogue_msg<T, 3>(ep, s0.w, comp, xn);
        step_epilogue_msg<T, 4>(ep, s1.x, comp, xn);
        step_epilogue_msg<T, 5>(ep, s1.y, comp, xn);
        step_epilogue_msg<T, 6>(ep, s1.z, comp, xn);
        step_epilogue_msg<T, 7>(ep, s1.w, comp, xn);
    }
}
// a row's slots in the halo messages (all 0xFFFFFFFF for an interior row / without a brick): loaded with the velocity, early
template <typename T>
__device__ __forceinline__ void step_epilogue_slots(const StepEpilogue<T> &ep, unsigned idx, uint4 &s0, uint4 &s1) {
    s0 = s1 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    if (ep.row_slots != nullptr && idx >= ep.cap_int) {
        const uint4 *rs = reinterpret_cast<const uint4 *>(ep.row_slots) + (size_t)(idx - ep.cap_int) * 2;
        s0 = rs[0];
        if (ep.n_msg > 4) s1 = rs[1];
    }
}

// one row at a time (the generic single-row routine: rows the straight-line forms hand back): everything loaded here
template <typename T, bool HALO = true>
__device__ __forceinline__ void step_epilogue_lane(const StepEpilogue<T> &ep, unsigned idx, unsigned comp, float tot, T own) {
    const T v = comp < 3u ? reinterpret_cast<const T *>(ep.vel)[(size_t)idx * 4 + comp] : (T)0;
    uint4 s0, s1;
    if constexpr (HALO) step_epilogue_slots<T>(ep, idx, s0, s1);
    const T xn = step_epilogue_core<T>(ep, idx, comp, tot, own, v);
    if constexpr (HALO) step_epilogue_halo<T>(ep, s0, s1, comp, xn);
}

#ifndef __HIPCC_RTC__ // (host-side dispatch declarations: nothing a generated unit needs)
// what the next fused-step launches of this thread carry as their epilogue (set by context.hip around fused_forces_impl)
struct htfs_step_epilogue_host; // (= htfs_step_epilogue of include/htf_standin.h)
const void *&step_epilogue_request();
int &step_epilogue_level(); // 1: the integrator alone, 2: + a brick's halo messages
// HTF_POT_JIT (csrc/jit.hip): a code object built by hoomd_tf_amd/codegen.py from csrc/jit_unit.hip
struct JitKernels;
int jit_create(const void *image, size_t bytes, JitKernels **out);
void jit_destroy(JitKernels *k);
int jit_launch_fused(const PotParams &p, const void *pos, int pos_dtype, unsigned N, unsigned NN, unsigned offset, unsigned batch,
                     const htf_box *box, const unsigned *n_neigh, const unsigned *nlist, const unsigned *head_list, double rmax,
                     void *force, void *virial9, int out_f64, unsigned *check_count, float4 *positions_out, float4 *dest,
                     unsigned *counts_io, hipStream_t s);
int jit_launch_eval(const PotParams &p, const void *nlist, int in_dtype, unsigned B, unsigned NN, void *force, void *virial9,
                    int out_f64, const unsigned *counts, hipStream_t s);
// the training sweep of a unit compiled with weights (train_pair.hip's row loop around the generated jets): block partials
int jit_launch_train(const PotParams &p, const void *nlist, int in_dtype, unsigned B, unsigned NN, const void *labels, int lab_f64,
                     void *pred, float *partials, unsigned grid, hipStream_t s);
int jit_launch_train_list(const PotParams &p, const void *pos, int pos_dtype, unsigned B, unsigned NN, const htf_box *box,
                          const unsigned *n_neigh, const unsigned *nlist, const unsigned *head_list, double rmax, const void *labels,
                          int lab_f64, void *pred, float *partials, unsigned grid, hipStream_t s);
int train_list_dispatch(const PotParams &p, const void *pos, int pos_dtype, unsigned B, unsigned NN, const htf_box *box,
                        const unsigned *n_neigh, const unsigned *nlist, const unsigned *head_list, double rmax, const void *labels,
                        int label_dtype, void *pred, float *accum, float *scratch, hipStream_t stream);
int jit_num_params(const JitKernels *k);

// counts (nullable): live slots per row; slots >= counts[row] are known zero padding and not loaded
int eval_pair_dispatch(const PotParams &p, const void *nlist, int in_dtype, unsigned B, unsigned NN,
                       void *force, int force_dtype, void *virial9, const unsigned *counts, hipStream_t stream);

// pair-vector build with the positions side buffer staged by the same kernel
int build_pair_vectors_impl(void *dest, int dest_dtype, const void *d_pos, int pos_dtype, unsigned N, unsigned NN,
                            unsigned offset, unsigned batch_size, const htf_box *box, const unsigned *d_n_neigh,
                            const unsigned *d_nlist, const unsigned *d_head_list, double rmax,
                            unsigned *d_max_count, float4 *positions_out, unsigned *counts_io, hipStream_t s);

// Kernel-exact timing for the profiler (htf_profile_enable): when `start` is set, the next fused-step launch of this thread goes
// through hipExtLaunchKernelGGL, which stamps the two events with the kernel's own begin and end (what rocprofv3 reports),
// instead of being bracketed by hipEventRecord calls, which add the command processor's event handling to the interval.
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
    bool used = false;
};
LaunchEvents &launch_events();

int fused_forces_impl(const PotParams &p, const void *pos, int pos_dtype, unsigned N, unsigned NN, unsigned offset,
                      unsigned batch, const htf_box *box, const unsigned *n_neigh, const unsigned *nlist,
                      const unsigned *head_list, double rmax, void *force, int force_dtype, void *virial9,
                      unsigned *check_count, float4 *positions_out, float4 *dest, unsigned *counts_io, hipStream_t s);

int eval_pair2_dispatch(const PotParams &pa, const PotParams &pb, const void *nlist, int in_dtype, unsigned B,
                        unsigned NN, void *forceA, void *forceB, int force_dtype, float *partials, float rdf_r0,
                        float rdf_r1, unsigned rdf_nbins_total, unsigned *rdf_hist, hipStream_t stream);
unsigned eval_pair2_num_partials(unsigned B, unsigned NN);

int potential_num_params(const PotParams &p);
size_t train_scratch_floats(const PotParams &p, unsigned B, unsigned NN);
int train_pair_dispatch(const PotParams &p, const void *nlist, int in_dtype, unsigned B, unsigned NN,
                        const void *labels, int label_dtype, void *pred, float *accum, float *scratch,
                        hipStream_t stream);

int fused_forces2_impl(const PotParams &pa, const PotParams &pb, const void *pos, int pos_dtype, unsigned N, unsigned NN,
                       unsigned offset, unsigned batch, const htf_box *box, const unsigned *n_neigh,
                       const unsigned *nlist, const unsigned *head_list, double rmax, void *fa, void *fb,
                       int force_dtype, float *partials, float rdf_r0, float rdf_r1, unsigned rdf_nb,
                       unsigned *rdf_hist, float4 *dest, unsigned *counts_io, hipStream_t s);
unsigned fused_forces2_num_partials(unsigned batch);

struct TopkDevice;
int topk_create(const htf_potential_desc *d, TopkDevice **out);
void topk_destroy(TopkDevice *m);
int topk_eval(const TopkDevice *m, const void *nlist, int in_dtype, unsigned B, unsigned NN, void *force,
              int force_dtype, void *virial9, hipStream_t stream);

struct MlpDevice;
int mlp_create(const htf_potential_desc *d, MlpDevice **out);
void mlp_destroy(MlpDevice *m);
int mlp_eval(const MlpDevice *m, const void *nlist, int in_dtype, unsigned B, unsigned NN, void *force,
             int force_dtype, void *virial9, hipStream_t stream);
#endif // __HIPCC_RTC__

} // namespace htf
