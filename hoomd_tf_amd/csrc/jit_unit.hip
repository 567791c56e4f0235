// The translation unit hoomd_tf_amd/codegen.py compiles for a TRACED pair energy (HTF_POT_JIT):
//
//     hipcc --genco --offload-arch=gfx950 -O3 -DHTF_JIT_BODY_FILE='"<body>.inc"' jit_unit.hip -o <hash>.hsaco
//
// -- pair_math.h's pair_eval_f<HTF_POT_JIT> with the generated body spliced in, and, around it, the SAME row loops the built-in
// closed forms run: the one-kernel step's plain two-row form (tensor written or not, fp32 / fp64 positions), its one-row form
// for virial requests, and the streaming evaluator over a pair-vector tensor.  The instantiations get C names so that
// csrc/jit.hip finds them with hipModuleGetFunction.  Nothing of this file is part of libhtf_amd.so.
#define HTF_JIT_UNIT 1
#include "htf_common.h"
// the generated statements: `e = ...; dedr = ...;` in terms of s, ds, r, x, y, z (see pair_math.h)
#include HTF_JIT_BODY_FILE
#define HTF_JIT_BODY HTF_JIT_BODY_TEXT
#if defined(HTF_JIT_NPARAMS) && HTF_JIT_NPARAMS > 0   // (before pair_math.h: its pair_eval_grad<HTF_POT_JIT> is conditional on it)
#define HTF_JIT_TRAIN_BODY HTF_JIT_TRAIN_BODY_TEXT
#endif
#ifdef HTF_JIT_ROW_TEXT // (a row function: energy_i = F(sum_j e_ij) -- pair_math.h row_function<HTF_POT_JIT>)
#define HTF_JIT_ROW_FN HTF_JIT_ROW_TEXT
#endif
#include "fused_eval.hip"
#include "eval_pair.hip"

using namespace htf;

#define HTF_JIT_ROWS2(NAME, STORE, PT)                                                                                              \
    extern "C" __global__ __launch_bounds__(256) void NAME(                                                                         \
        const typename Vec4<PT>::type *__restrict__ pos, unsigned N, unsigned NN, unsigned offset, unsigned batch, BoxT<PT> box,    \
        const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist, const unsigned *__restrict__ head_list,            \
        PT rmaxsq, void *__restrict__ force, int out_f64, PotParams pin, unsigned *__restrict__ check_count,                        \
        float4 *__restrict__ positions_out, float4 *__restrict__ dest, unsigned *__restrict__ counts_io) {                          \
        fused_forces_rows2_body<HTF_POT_JIT, STORE, 2, PT>(pos, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmaxsq, force, \
                                                           out_f64, pin, check_count, positions_out, dest, counts_io);             \
    }
HTF_JIT_ROWS2(htf_jit_rows2_f32_store, true, float)
HTF_JIT_ROWS2(htf_jit_rows2_f32_nostore, false, float)
HTF_JIT_ROWS2(htf_jit_rows2_f64_store, true, double)
HTF_JIT_ROWS2(htf_jit_rows2_f64_nostore, false, double)

// the four-row form with merged tails (fp32 positions, batches of >= 49 152 rows: launch_fused's rule for LJ and WCA) -- round 6
#define HTF_JIT_TAILS4(NAME, STORE)                                                                                                 \
    extern "C" __global__ __launch_bounds__(256) void NAME(                                                                         \
        const float4 *__restrict__ pos, unsigned N, unsigned NN, unsigned offset, unsigned batch, BoxT<float> box,                  \
        const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist, const unsigned *__restrict__ head_list,            \
        float rmaxsq, void *__restrict__ force, int out_f64, PotParams pin, unsigned *__restrict__ check_count,                     \
        float4 *__restrict__ positions_out, float4 *__restrict__ dest, unsigned *__restrict__ counts_io) {                          \
        const PotParams p = resolve_theta<HTF_POT_JIT>(pin);                                                                        \
        const unsigned lane = threadIdx.x & 63u;                                                                                    \
        const unsigned w0 = 4 * __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);                       \
        if (w0 >= batch) return;                                                                                                    \
        fused_rows_group_tails<HTF_POT_JIT, STORE, 4, float, 0>(w0, lane, pos, N, NN, offset, batch, box, n_neigh, nlist, head_list, \
                                                                rmaxsq, force, out_f64, p, check_count, positions_out, dest,       \
                                                                counts_io, nullptr);                                               \
    }
HTF_JIT_TAILS4(htf_jit_tails4_f32_store, true)
HTF_JIT_TAILS4(htf_jit_tails4_f32_nostore, false)

#define HTF_JIT_ROW1V(NAME, STORE, PT)                                                                                              \
    extern "C" __global__ __launch_bounds__(256) void NAME(                                                                         \
        const typename Vec4<PT>::type *__restrict__ pos, unsigned N, unsigned NN, unsigned offset, unsigned batch, BoxT<PT> box,    \
        const unsigned *__restrict__ n_neigh, const unsigned *__restrict__ nlist, const unsigned *__restrict__ head_list,            \
        PT rmaxsq, void *__restrict__ force, void *__restrict__ virial9, int out_f64, PotParams pin,                                \
        unsigned *__restrict__ check_count, float4 *__restrict__ positions_out, float4 *__restrict__ dest,                          \
        unsigned *__restrict__ counts_io) {                                                                                         \
        fused_forces_body<HTF_POT_JIT, true, STORE, PT>(pos, N, NN, offset, batch, box, n_neigh, nlist, head_list, rmaxsq, force,   \
                                                        virial9, out_f64, pin, check_count, positions_out, dest, counts_io);       \
    }
HTF_JIT_ROW1V(htf_jit_row1v_f32_store, true, float)
HTF_JIT_ROW1V(htf_jit_row1v_f32_nostore, false, float)
HTF_JIT_ROW1V(htf_jit_row1v_f64_store, true, double)
HTF_JIT_ROW1V(htf_jit_row1v_f64_nostore, false, double)

#define HTF_JIT_EVAL(NAME, VIRIAL, IT)                                                                                              \
    extern "C" __global__ __launch_bounds__(256) void NAME(const typename Vec4<IT>::type *__restrict__ nlist, unsigned B,           \
                                                           unsigned NN, void *__restrict__ force, void *__restrict__ virial9,      \
                                                           int out_f64, PotParams pin, const unsigned *__restrict__ counts) {      \
        eval_pair_body<HTF_POT_JIT, 16, VIRIAL, IT>(nlist, B, NN, force, virial9, out_f64, pin, counts);                            \
    }
HTF_JIT_EVAL(htf_jit_eval_f32, false, float)
HTF_JIT_EVAL(htf_jit_eval_f32_virial, true, float)
HTF_JIT_EVAL(htf_jit_eval_f64, false, double)
HTF_JIT_EVAL(htf_jit_eval_f64_virial, true, double)

#if defined(HTF_JIT_NWEIGHTS) && !(defined(HTF_JIT_NPARAMS) && HTF_JIT_NPARAMS > 0)
// (a row-function unit that reads weights: no training sweep, but the library still wants to know how long p.theta is)
extern "C" __device__ const int htf_jit_nparams = HTF_JIT_NWEIGHTS;
#endif

// A traced energy WITH WEIGHTS (round 6): the body file also defines HTF_JIT_NPARAMS and HTF_JIT_TRAIN_BODY_TEXT -- the same
// expression as forward-mode jets over (r', w_k) -- and the unit carries the library's training sweep around it.
#if defined(HTF_JIT_NPARAMS) && HTF_JIT_NPARAMS > 0
#include "train_pair.hip"
extern "C" __device__ const int htf_jit_nparams = HTF_JIT_NPARAMS;
#define HTF_JIT_TRAIN(NAME, IT)                                                                                                     \
    extern "C" __global__ __launch_bounds__(256) void NAME(const typename Vec4<IT>::type *__restrict__ nlist, unsigned B, unsigned NN, \
                                                           const void *__restrict__ labels, int lab_f64, void *__restrict__ pred,   \
                                                           PotParams pin, float *__restrict__ partials) {                           \
        train_pair_body<HTF_POT_JIT, IT>(nlist, B, NN, labels, lab_f64, pred, pin, partials);                                       \
    }
HTF_JIT_TRAIN(htf_jit_train_f32, float)
HTF_JIT_TRAIN(htf_jit_train_f64, double)
// ... and its list form (no pair-vector tensor: train_pair.hip train_list_body)
#define HTF_JIT_TRAIN_LIST(NAME, PT)                                                                                                \
    extern "C" __global__ __launch_bounds__(256) void NAME(const typename Vec4<PT>::type *__restrict__ pos, unsigned B, unsigned NN, \
                                                           BoxT<PT> box, const unsigned *__restrict__ n_neigh,                     \
                                                           const unsigned *__restrict__ nlist, const unsigned *__restrict__ head_list, \
                                                           PT rmaxsq, const void *__restrict__ labels, int lab_f64,                 \
                                                           void *__restrict__ pred, PotParams pin, float *__restrict__ partials) {  \
        train_list_body<HTF_POT_JIT, PT>(pos, B, NN, box, n_neigh, nlist, head_list, rmaxsq, labels, lab_f64, pred, pin, partials); \
    }
HTF_JIT_TRAIN_LIST(htf_jit_train_list_f32, float)
HTF_JIT_TRAIN_LIST(htf_jit_train_list_f64, double)
#endif
