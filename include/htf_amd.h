/* htf_amd.h -- C ABI of the MI355X-native hoomd-tf force/energy evaluator.
 *
 * This is the drop-in boundary (DESIGN.md "Boundary"): plain C,
 * raw device pointers and sizes, no torch / pybind / HOOMD / TensorFlow types.
 * Every entry point names the reference interface it replaces (paths relative to
 * the hoomd-tf v2.4.0 tree).  All device work is enqueued on the caller's
 * hipStream_t; nothing here synchronises the device.  Caller-owned pointers are
 * never retained past the call (reference: valid only inside computeForces,
 * TensorflowCompute.cc:129-216).
 *
 * Array layouts are HOOMD-blue 2.x's:
 *   Scalar4 pos[N + n_ghost]   (x, y, z, w = int type bits)      ParticleData
 *   unsigned n_neigh[N], head_list[N], nlist[..]   FULL neighbor list
 *   Scalar4 force[N]           (fx, fy, fz, per-particle energy) ForceCompute::m_force
 *   Scalar  virial[6 * pitch]  (xx, xy, xz, yy, yz, zz SoA)       ForceCompute::m_virial
 * Scalar is float or double (htf_dtype); the evaluator computes in fp32 exactly as
 * the reference model does (simmodel.py:15, tf.cast at :226-227).
 */
#ifndef HTF_AMD_H_
#define HTF_AMD_H_

#ifndef __HIPCC_RTC__ /* (a generated unit compiled by hipRTC: the basic types are built in, there are no system headers) */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct of this header changes size or layout or an enum gains a value (2: htf_potential_desc gained
 * poly_cut, htf_mlp_precision gained HTF_MLP_SPLIT16; 3: the stand-in's neighbor-search entry points of htf_standin.h take the
 * caller's candidate-range table).  Every binding compares the value it was built against with
 * htf_abi_version() of the library it loaded and refuses a mismatch: hoomd_tf_amd/_lib.py, csrc/pybind_abi.cc,
 * integration/hoomd_shim/TensorflowComputeAMD.cc. */
#define HTF_AMD_ABI_VERSION 4

/* the library is built with -fvisibility=hidden (as the reference is,
 * htf/CMakeLists.txt:48); only these entry points are exported */
#if defined(__GNUC__)
#define HTF_API __attribute__((visibility("default")))
#else
#define HTF_API
#endif

typedef void *htf_stream; /* hipStream_t */

/* status codes (reference: C++ exceptions / tf.errors, SURVEY 8(b) "Errors") */
enum htf_status {
    HTF_OK = 0,
    HTF_ERR_INVALID = 1,        /* ValueError: bad argument / configuration       */
    HTF_ERR_DEVICE = 2,         /* RuntimeError: HIP runtime failure               */
    HTF_ERR_NLIST_OVERFLOW = 3, /* tf InvalidArgumentError 'Neighbor list is full!' simmodel.py:214-224 */
    HTF_ERR_SKEWED_BOX = 4,     /* tf InvalidArgumentError 'box is skewed'  simmodel.py:195 */
    HTF_ERR_NOMEM = 5
};

enum htf_dtype { HTF_F32 = 0, HTF_F64 = 1 };

/* TensorflowCompute.h:44-48 enum class FORCE_MODE */
enum htf_force_mode { HTF_TF2HOOMD = 0, HTF_HOOMD2TF = 1 };

/* HOOMD BoxDim as the plugin sees it: lo/hi/tilt as TensorflowCompute.cc:271-282
 * updateBox() lays them out, plus the periodic flags minImage needs. */
typedef struct htf_box {
    double lo[3];
    double hi[3];
    double tilt[3]; /* xy, xz, yz */
    int periodic[3];
} htf_box;

/* ---- potentials: the declarative models of build_examples.py / layers.py ---- */
enum htf_potential_kind {
    HTF_POT_NONE = 0,
    HTF_POT_LJ = 1,        /* build_examples.py:67-77 LJModel, :104-115 LJVirialModel */
    HTF_POT_WCA = 2,       /* layers.py:52-98 WCARepulsion + build_examples.py:221-228 */
    HTF_POT_RINV_POLY = 3, /* E_i = sum_j sum_k coef_k * rinv^power_k (BenchmarkPotential :25-30, example 01) */
    HTF_POT_SIMPLE = 4,    /* build_examples.py:9-22 SimplePotential (forward only) */
    HTF_POT_PAIR_MLP = 5,  /* safe_norm -> RBFExpansion -> Dense-Dense-Dense */
    HTF_POT_GAUSS = 6,     /* e = c * exp(-(r - r0)^2 / gap) * [r > 3e-6], r = safe_norm(x): one RBFExpansion
                            * channel (layers.py:46-49) as a pair energy -- the soft RDF bin of config C4 */
    HTF_POT_LJ_PARAM = 7,  /* trainable LJ of example 06 / build_examples.py:336-372 (LJLayer): r = safe_norm(x),
                            * q = w1^6 / r^6 (divide_no_nan), e = w0 * 4 (q^2 - q) / 2; params (w0, w1) */
    HTF_POT_TOPK_MLP = 8,  /* example 08 / build_examples.py:199-218 NlistNN, a per-PARTICLE network:
                            * top_n = tf.sort(nlist_rinv(nlist), DESCENDING)[:, :K] -> Dense(H1) -> Dense(H2) ->
                            * Dense(1) = E_i.  desc.K = top_neighs (<= 16), H1, H2 <= 64, weights as for PAIR_MLP */
    HTF_POT_JIT = 9        /* round 5: an elementwise pair energy e(rinv, r) the caller TRACED from model code (htf/simmodel.py:87-121: any
                            * compute() is legal upstream) and compiled for gfx950 -- hoomd_tf_amd/codegen.py emits the body of
                            * pair_math.h's pair_eval_f<HTF_POT_JIT> (energy + d/dr, forward-mode), htf_jit_compile (hipRTC) or `hipcc --genco` builds the
                            * kernels of csrc/jit_unit.hip around it, desc.jit_image hands the code object over.  Same row loops,
                            * same launch geometry as the built-in closed forms; no second backend. */
};

enum htf_activation { HTF_ACT_LINEAR = 0, HTF_ACT_TANH = 1 };
enum htf_mlp_precision {
    HTF_MLP_FP32 = 0,  /* v_mfma_f32_32x32x2_f32 on fp32 operands */
    HTF_MLP_BF16 = 1,  /* bf16 operands (weights and activations rounded), fp32 accumulation: reduced precision */
    HTF_MLP_SPLIT = 2, /* fp32-level results on the bf16 matrix pipeline: every fp32 operand is split EXACTLY
                        * into three bf16 values (8 + 8 + 8 significand bits) and a product is the six partial
                        * products down to 2^-16 of it; what is dropped (2^-24) is the size of fp32's own rounding */
    HTF_MLP_SPLIT16 = 3 /* fp32-level results on the fp16 matrix pipeline: every fp32 operand as hi + lo, both fp16 and
                        * rounded to nearest (11 + 11 significand bits: 2^-22 relative, never worse than 2^-25 absolute),
                        * a product as hi*hi + hi*lo + lo*hi -- three MFMAs where SPLIT issues six, and two VALU
                        * instructions per split element where SPLIT needs five.  Operands must stay inside fp16's
                        * range (|x| < 65504: tanh activations always do; weights are checked at creation) */
};

#define HTF_MAX_POLY_TERMS 8

typedef struct htf_potential_desc {
    int kind;   /* htf_potential_kind */
    /* WCA */
    double sigma;
    /* GAUSS */
    double gauss_r0, gauss_gap, gauss_coef;
    /* LJ_PARAM start values */
    double lj_w0, lj_w1;
    /* Trainable potentials (LJ_PARAM: w0, w1; WCA: sigma; RINV_POLY: coef[0..n_terms)): optional
     * DEVICE parameter vector, read by the kernels at launch instead of the host values above,
     * so that an optimizer step on the device is seen by the next evaluation.  Borrowed. */
    const float *d_theta;
    /* RINV_POLY */
    int n_terms;
    double coef[HTF_MAX_POLY_TERMS];
    int power[HTF_MAX_POLY_TERMS];
    /* PAIR_MLP and TOPK_MLP: host pointers, row-major [in, out] like Keras Dense kernels; copied */
    int K, H1, H2;
    int activation;    /* htf_activation */
    int mlp_precision; /* htf_mlp_precision: MFMA operand type; accumulation is fp32 */
    double rbf_low, rbf_high;
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    /* RINV_POLY: optional hard mask `tf.cast(tf.norm(nlist[:, :, :3], axis=2) < poly_cut, tf.float32) * energy`
     * (examples/01. Quickstart.ipynb cell 3: WCA as r^-12 inside 2^(1/6)); the mask carries no gradient.  0 = no mask.
     * (appended with ABI version 2: a caller built against the version-1 layout passes a SHORTER struct and must be rebuilt;
     *  the version check is what catches it) */
    double poly_cut;
    /* HTF_POT_JIT: the code object (appended with ABI version 3) */
    const void *jit_image;
    size_t jit_image_bytes;
    /* HTF_POT_JIT: HTF_JIT_READS_OWN_TYPE when the traced energy reads the ROW particle's own type (positions[i, 3]: a parameter
     * table by species pair) -- the streaming evaluator then needs the positions tensor beside the pair vectors
     * (htf_eval_forces_typed); the one-kernel step has it anyway.  The neighbor's type (nlist[i, j, 3]) is in the tensor. */
    int jit_flags;
} htf_potential_desc;
#define HTF_JIT_READS_OWN_TYPE 1

typedef struct htf_potential htf_potential; /* opaque; owns device copies of weights */

HTF_API const char *htf_last_error(void); /* thread-local message for the last non-OK status */
HTF_API int htf_abi_version(void);
HTF_API int htf_device_count(void);

HTF_API int htf_potential_create(const htf_potential_desc *desc, htf_potential **out);
HTF_API void htf_potential_destroy(htf_potential *pot);

/* ------------------------------------------------------------------------- *
 * Stateless kernels -- one per reference kernel / TF graph on the path.
 * ------------------------------------------------------------------------- */

/* Replaces htf_gpu_reshape_nlist (TensorflowCompute.cuh:32-48, .cu:80-209) and the
 * CPU prepareNeighbors (TensorflowCompute.cc:303-374), whose semantics it follows:
 * zero fill, neighbor kept unless rsq > rmax^2, slot index wraps modulo NN.
 * dest: [batch_size * NN] Scalar4 of dest_dtype.  d_max_count (nullable): device
 * unsigned, atomically max'ed with the largest kept-neighbor count of the batch
 * (> NN means the list overflowed). */
HTF_API int htf_build_pair_vectors(void *dest, int dest_dtype,
                           const void *d_pos, int pos_dtype,
                           unsigned N, unsigned NN, unsigned offset, unsigned batch_size,
                           unsigned n_ghost, const htf_box *box,
                           const unsigned *d_n_neigh, const unsigned *d_nlist,
                           const unsigned *d_head_list, double rmax,
                           unsigned *d_max_count, htf_stream stream);

/* Replaces the TF2 graph SimModel.compute -> compute_nlist_forces (simmodel.py:
 * 526-555) + compute_outputs / TfToHoomd (simmodel.py:240-255, tf2hoomd.cc:17-82):
 * nlist [B, NN, 4] (nlist_dtype) -> force [B] Scalar4 (force_dtype) written in
 * place (fx, fy, fz, energy).  virial9 (nullable): [B, 9] row-major 3x3 of
 * force_dtype, as _compute_virial returns (simmodel.py:509-523). */
HTF_API int htf_eval_forces(const htf_potential *pot,
                    const void *d_nlist, int nlist_dtype, unsigned B, unsigned NN,
                    void *d_force, int force_dtype, void *d_virial9, htf_stream stream);

/* htf_eval_forces for a potential that reads the row particle's own type: d_positions is the [B, 4] tensor compute() receives
 * beside nlist (simmodel.py:99-105: x, y, z, type as a float), positions_dtype its scalar type.  Any potential may be evaluated
 * through it (the positions are then ignored); a HTF_JIT_READS_OWN_TYPE potential handed to htf_eval_forces is HTF_ERR_INVALID. */
HTF_API int htf_eval_forces_typed(const htf_potential *pot,
                    const void *d_nlist, int nlist_dtype, unsigned B, unsigned NN,
                    const void *d_positions, int positions_dtype,
                    void *d_force, int force_dtype, void *d_virial9, htf_stream stream);

/* htf_build_pair_vectors + htf_eval_forces in one pass with the pair vectors kept in
 * registers (no [B, NN, 4] tensor): same semantics and per-slot arithmetic, ~8x less HBM
 * traffic.  Closed-form potentials only.  d_check_count (nullable, caller zeroes): max'ed
 * with max_i sum_j [dx_ij > 0], the quantity SimModel's check_nlist asserts on. */
HTF_API int htf_fused_forces(const htf_potential *pot, const void *d_pos, int pos_dtype,
                     unsigned N, unsigned NN, unsigned offset, unsigned batch_size, const htf_box *box,
                     const unsigned *d_n_neigh, const unsigned *d_nlist, const unsigned *d_head_list,
                     double rmax, void *d_force, int force_dtype, void *d_virial9,
                     unsigned *d_check_count, htf_stream stream);

/* htf_build_pair_vectors and htf_eval_forces as ONE kernel: d_dest ([batch, NN, 4] fp32) receives
 * exactly what htf_build_pair_vectors writes (bit-identical, zero padded) while the forces are
 * evaluated on the pair vectors still in registers -- the producing kernel carries the
 * evaluator as its epilogue, so the tensor is written once and not re-read. */
HTF_API int htf_build_eval_forces(const htf_potential *pot, void *d_dest, const void *d_pos, int pos_dtype,
                          unsigned N, unsigned NN, unsigned offset, unsigned batch_size, const htf_box *box,
                          const unsigned *d_n_neigh, const unsigned *d_nlist, const unsigned *d_head_list,
                          double rmax, void *d_force, int force_dtype, void *d_virial9,
                          unsigned *d_check_count, htf_stream stream);

/* Two potentials in ONE pass over the pair vectors: forceA <- potA, forceB <- potB (both [B]
 * Scalar4).  potB must be HTF_POT_GAUSS, potA a closed-form potential.  d_partials (nullable,
 * >= htf_eval2_num_partials(B, NN) floats): per-block sums of forceB[i].w, to be reduced with
 * htf_reduce_partials -- the collective-variable sum of an EDS-biased model (config C4)
 * without a second sweep over the 268 MB tensor and without float atomics.  d_rdf_hist
 * (nullable, [rdf_nbins_total], accumulated into): the untyped compute_rdf histogram
 * (htf_rdf_histogram semantics) of the same pair vectors, fused into the sweep. */
HTF_API int htf_eval_forces2(const htf_potential *potA, const htf_potential *potB,
                     const void *d_nlist, int nlist_dtype, unsigned B, unsigned NN,
                     void *d_forceA, void *d_forceB, int force_dtype,
                     float *d_partials,
                     float rdf_r0, float rdf_r1, unsigned rdf_nbins_total, unsigned *d_rdf_hist,
                     htf_stream stream);
HTF_API unsigned htf_eval2_num_partials(unsigned B, unsigned NN);

/* htf_build_pair_vectors + htf_eval_forces2 as ONE kernel (config C4's whole sweep): the pair
 * vectors are evaluated for both potentials, summed into the CV partials and binned into the
 * compute_rdf histogram while they are in registers; d_dest (nullable, fp32 [batch, NN, 4]) also
 * receives the tensor, bit-identical to htf_build_pair_vectors'.  d_partials: at least
 * htf_build_eval2_num_partials(batch_size) floats (one per persistent block). */
HTF_API int htf_build_eval_forces2(const htf_potential *potA, const htf_potential *potB, void *d_dest,
                           const void *d_pos, int pos_dtype, unsigned N, unsigned NN, unsigned offset,
                           unsigned batch_size, const htf_box *box, const unsigned *d_n_neigh,
                           const unsigned *d_nlist, const unsigned *d_head_list, double rmax,
                           void *d_forceA, void *d_forceB, int force_dtype, float *d_partials,
                           float rdf_r0, float rdf_r1, unsigned rdf_nbins_total, unsigned *d_rdf_hist,
                           htf_stream stream);
HTF_API unsigned htf_build_eval2_num_partials(unsigned batch_size);

/* *d_out = scale * sum(d_partials[0..n)) in a fixed order (one block; deterministic). */
HTF_API int htf_reduce_partials(const float *d_partials, unsigned n, float scale, float *d_out, htf_stream stream);

/* EDS-biased force assembly: force[i] = (base.xyz + alpha * bias.xyz, base.w + alpha * cv) with
 * alpha = *d_alpha, cv = *d_cv read on the device: compute_nlist_forces(nlist, E_base + alpha*cv)
 * for a scalar collective variable cv whose unit-alpha forces are `bias` (simmodel.py:526-578:
 * a rank-0 energy term is tiled into every particle's energy column). */
HTF_API int htf_bias_combine(void *d_force, const void *d_bias, const float *d_alpha, const float *d_cv,
                     int dtype, unsigned N, htf_stream stream);

/* ---- online training, FORCE_MODE::hoomd2tf (tensorflowcompute.py:347-370) ----
 * One pass over the pair vectors of a batch for a trainable closed-form potential:
 * predicted [F, E] per particle, residual against the labels (HOOMD net / reference forces,
 * Scalar4), and d(sum of squared residuals)/d(theta) through the force.  d_accum [1 + P] floats
 * (P = htf_potential_num_params) receives {sum_i sum_c res_ic^2, d/dtheta_0, ...}; it is
 * OVERWRITTEN.  d_pred (nullable): the predicted forces [B] Scalar4.  d_scratch: at least
 * htf_train_scratch_floats(pot, B, NN) floats.  Keras 'MeanSquaredError' over the [B, 4]
 * batch is accum[0] / (4 B). */
HTF_API int htf_potential_num_params(const htf_potential *pot);
HTF_API size_t htf_train_scratch_floats(const htf_potential *pot, unsigned B, unsigned NN);
HTF_API int htf_train_pair_grad(const htf_potential *pot, const void *d_nlist, int nlist_dtype,
                        unsigned B, unsigned NN, const void *d_labels, int label_dtype,
                        void *d_pred, float *d_accum, float *d_scratch, htf_stream stream);

/* The same sweep FROM HOOMD'S INDEX LIST (round 6): positions are gathered and the pair vectors formed in registers exactly as
 * htf_build_pair_vectors forms them (minimum image, r_cut mask, a row's first NN kept neighbors -- the last NN past an
 * overflow), so a training step neither writes nor re-reads the [B, NN, 4] tensor.  Rows 0 .. B-1 of the arrays (ghosts
 * behind them are gathered like any neighbor); d_accum, d_pred, d_scratch (htf_train_scratch_floats) as above.  Closed forms
 * and generated units with weights; a pair-MLP is HTF_ERR_INVALID (its sweep reads the tensor). */
HTF_API int htf_train_pair_grad_list(const htf_potential *pot, const void *d_pos, int pos_dtype, unsigned B, unsigned NN,
                             const htf_box *box, const unsigned *d_n_neigh, const unsigned *d_nlist,
                             const unsigned *d_head_list, double rmax, const void *d_labels, int label_dtype,
                             void *d_pred, float *d_accum, float *d_scratch, htf_stream stream);

enum htf_optimizer_kind { HTF_OPT_SGD = 0, HTF_OPT_ADAM = 1, HTF_OPT_NADAM = 2 };
typedef struct htf_optimizer_desc {
    int kind;                  /* tf.keras.optimizers.{SGD, Adam, Nadam} update rules */
    float lr, beta1, beta2, epsilon;
    unsigned nonneg_mask;      /* bit k: tf.keras.constraints.NonNeg on theta_k            */
    float l1_reg[8];           /* d(regulariser)/d(theta_k) added to the gradient (WCARepulsion: -strength) */
} htf_optimizer_desc;
#define HTF_OPT_STATE_FLOATS 24 /* m[8], v[8], t, nadam m_schedule, loss_sum, n_steps, last_loss, pad */

/* One optimizer step ON THE DEVICE: grad_k = scale * d_accum[1 + k] + l1_reg[k]; loss = scale *
 * d_accum[0] is added to the running metric in d_state.  d_state: HTF_OPT_STATE_FLOATS floats,
 * zero-initialised by the caller (m_schedule is set to 1 on the first step). */
HTF_API int htf_optimizer_step(float *d_theta, unsigned P, const float *d_accum, float scale,
                       float *d_state, const htf_optimizer_desc *desc, htf_stream stream);

/* Same rules for a parameter vector of any length (pair-MLP weights).  d_state:
 * HTF_OPT_STATE_FLOATS + 2 P floats, zero-initialised (header as above, then m[P], v[P]);
 * nonneg_mask / l1_reg are ignored. */
HTF_API int htf_optimizer_step_n(float *d_theta, unsigned P, const float *d_accum, float scale, float *d_state,
                         const htf_optimizer_desc *desc, htf_stream stream);

/* Pair-MLP potentials created with desc.d_theta (flat Keras get_weights() order: W1 [K][H1] |
 * b1 | W2 [H1][H2] | b2 | W3 [H2] | b3; the host weight pointers may then be NULL) keep reading
 * that caller-owned device vector: after it changes (optimizer step, set_weights), rebuild the
 * MFMA operand images on the device.  A no-op for closed-form potentials.
 * precision HTF_MLP_SPLIT16 carries every operand as two fp16 halves: each image build checks |2.885 w| < 6e4 and that w is a
 * number, and reports a violation through a host-mapped word the NEXT evaluation / training call on the potential reads without
 * synchronising -- i.e. the error (HTF_ERR_ARG, "left fp16's range") surfaces one call late, after a step that ran on saturated
 * images.  Every refresh judges the range afresh: repair d_theta, refresh, and the potential is usable again. */
HTF_API int htf_potential_refresh(htf_potential *pot, htf_stream stream);

/* HTF_POT_JIT: the run-time compiler for a generated unit.  hipRTC (libhiprtc, bound with dlopen when first asked for) compiles
 * `unit_source` -- csrc/jit_unit.hip's text -- for `arch` ("gfx950") with the headers it includes handed over BY TEXT (name as
 * written in the #include, content): the library's own kernels' sources and the generated body; no compiler driver, no temporary
 * files, no GPU needed.  On success *image / *image_bytes hold the code object (htf_jit_free releases it) for
 * htf_potential_desc.jit_image.  `log` (nullable) receives the compiler's diagnostics, truncated to log_bytes.
 * htf_jit_available() = 1 when libhiprtc could be loaded (HTF_HIPRTC_LIB names it explicitly). */
HTF_API int htf_jit_available(void);
HTF_API int htf_jit_compile(const char *unit_source, const char *arch, int n_headers, const char *const *header_names,
                            const char *const *header_texts, int n_options, const char *const *options,
                            void **image, size_t *image_bytes, char *log, size_t log_bytes);
HTF_API void htf_jit_free(void *image);


/* Replaces htf_gpu_add_virial (TensorflowCompute.cu:41-71; CPU .cc:284-301):
 * dest[c*pitch + i] += src[i*9 + {0,1,2,4,5,8}]. */
HTF_API int htf_add_virial(void *d_dest, const void *d_src9, int dtype, unsigned N, size_t pitch, htf_stream stream);

/* Replaces htf_gpu_add_scalar4 (TensorflowCompute.cu:11-39): dest[i] += src[i]. */
HTF_API int htf_add_scalar4(void *d_dest, const void *d_src, int dtype, unsigned N, htf_stream stream);

/* Replaces TFArrayComm::receiveArray(+unstuff4) (TFArrayComm.h:86-130,
 * TFArrayComm.cu:9-29): dest[i] = src[offset+i], w = (Scalar)int_bits(w) when
 * unstuff4 != 0.  dest may be of a different dtype than src (fused cast). */
HTF_API int htf_copy_positions(void *d_dest, int dest_dtype, const void *d_src, int src_dtype,
                       unsigned offset, unsigned N, int unstuff4, htf_stream stream);

/* ForceCompute::calcEnergySum behind getLogValue("tensorflow") (TensorflowCompute.cc:376-395):
 * *d_out = sum_i force[i].w, accumulated in double in a fixed order. */
HTF_API int htf_energy_sum(const void *d_force, int dtype, unsigned N, double *d_out, htf_stream stream);

/* Replaces htf_gpu_copy3 (TFArrayComm.cu:31-56; TFArrayComm::sendArray(..., copy3 = true),
 * TFArrayComm.h:171): dest[i].xyz = src[i].xyz, dest[i].w (HOOMD's stuffed type) untouched --
 * how the mapped (coarse-grained) bead positions of enable_mapped_nlist are written back into
 * HOOMD's position array before the neighbor search (TensorflowCompute.cc:228-241). */
HTF_API int htf_copy3(void *d_dest, int dest_dtype, const void *d_src, int src_dtype, unsigned N, htf_stream stream);

/* compute_positions_forces (simmodel.py:492-506) for the per-particle radial energies positions-only models
 * build from tf.norm(positions, axis=1): e_i = coef * |p_i|^power with the norm over the first ncomp (3 or 4)
 * columns of the [N, 4] positions side buffer (x, y, z, un-stuffed type) -- build_examples.py:59-64
 * BenchmarkNonlistModel is coef 1, power -1, ncomp 4 with divide_no_nan.  force[i] =
 * (-d(sum e)/d p_i[:3], e_i) written in place (Scalar4 of force_dtype). */
HTF_API int htf_positions_forces_radial(const void *d_positions, int dtype, unsigned N, int ncomp, int power,
                                double coef, void *d_force, int force_dtype, htf_stream stream);

/* SimModel.compute_inputs check_nlist (simmodel.py:214-219): *d_out =
 * max(*d_out, max_i sum_j [nlist[i,j,0] > 0]).  The caller zeroes *d_out. */
HTF_API int htf_check_nlist(const void *d_nlist, int nlist_dtype, unsigned B, unsigned NN,
                    unsigned *d_out, htf_stream stream);

/* nlist_rinv (simmodel.py:618-635): out[B*NN] fp32. */
HTF_API int htf_nlist_rinv(const void *d_nlist, int nlist_dtype, unsigned B, unsigned NN,
                   float *d_out, htf_stream stream);

/* tf.math.top_k (what tf.sort(..., direction='DESCENDING') runs on): the k largest entries of every row of
 * x [B, n] (fp32, n <= 256), largest first, equal values in order of their index; d_values [B, k],
 * d_indices [B, k]. */
HTF_API int htf_top_k(const float *d_x, unsigned B, unsigned n, unsigned k, float *d_values, int *d_indices,
              htf_stream stream);

/* compute_rdf + masked_nlist (simmodel.py:638-693).  Histogram of |nlist xyz| with
 * tf.histogram_fixed_width semantics over nbins_total = nbins + 2 bins (values clamp
 * into the end bins; padded slots land in bin 0).  type_tensor (nullable): one float
 * per row at stride type_stride floats (e.g. positions[:,3]); type_i >= 0 keeps only rows
 * of that type (boolean_mask), type_j >= 0 zeroes slots whose neighbor type differs.
 * d_hist [nbins_total] is ACCUMULATED into (caller zeroes). */
HTF_API int htf_rdf_histogram(const void *d_nlist, int nlist_dtype, unsigned B, unsigned NN,
                      float r0, float r1, unsigned nbins_total,
                      const float *d_type_tensor, unsigned type_stride, int type_i, int type_j,
                      unsigned *d_hist, htf_stream stream);

/* compute_rdf's tail (simmodel.py:663-668): rdf[b] = hist[b+1] / (shell[b+1]^3 - shell[b]^3),
 * rs[b] = (shell[b] + shell[b+1]) / 2 with shell = linspace(r0, r1, nbins + 1), fp32. */
HTF_API int htf_rdf_finalize(const unsigned *d_hist, unsigned nbins, float r0, float r1,
                     float *d_rdf, float *d_rs, htf_stream stream);

/* RBFExpansion.call (layers.py:46-49): out[i, k] = exp(-(x[i] - c_k)^2 / gap), c = fp32
 * linspace(low, high, count), gap = c[1] - c[0]. */
HTF_API int htf_rbf_expansion(const float *d_x, size_t n, double low, double high, unsigned count,
                      float *d_out, htf_stream stream);

/* EDSLayer.call (layers.py:159-195) as a device-resident state machine: one step of the
 * running-mean / ssd / TF1-Adam update on a scalar collective variable read from
 * *d_cv.  d_state: 8 floats {mean, ssd, alpha, adam_m, adam_v, n, t, 0}, zero-initialised by
 * the caller; alpha after the call is d_state[2].  No host round trip. */
HTF_API int htf_eds_update(float *d_state, const float *d_cv, float set_point, int period,
                   float learning_rate, float cv_scale, htf_stream stream);

/* wrap_vector (simmodel.py:606-615): out = r - round(r / bs) * bs, bs = hi - lo, on n
 * 3-vectors (orthorhombic; round half to even). */
HTF_API int htf_wrap_vector(const void *d_r, int dtype, size_t n, const htf_box *box, void *d_out,
                    htf_stream stream);

/* ------------------------------------------------------------------------- *
 * Context: TensorflowCompute<M> (TensorflowCompute.h:75-250, .cc:29-216).
 * ------------------------------------------------------------------------- */
typedef struct htf_config {
    double r_cut;        /* ctor arg r_cut            TensorflowCompute.h:85 */
    unsigned nneighs;    /* ctor arg nneighs          :86 */
    int force_mode;      /* FORCE_MODE                :87 */
    unsigned period;     /* ctor arg period           :88 */
    unsigned batch_size; /* 0 = all local particles   :89, .cc:143 */
    int scalar_dtype;    /* HOOMD Scalar: HTF_F32 / HTF_F64 (isDoublePrecision, .h:117-124) */
    int check_nlist;     /* SimModel(check_nlist=True) simmodel.py:15 */
    int virial;          /* SimModel(virial=True)      simmodel.py:15 */
    unsigned max_n;      /* m_pdata->getMaxN(): sizes the scratch, reallocate() .cc:91-121 */
    int fused;           /* 0: two kernels, build then evaluate (the reference's dataflow).
                          * 2: ONE kernel writes the pair-vector tensor AND evaluates it while it is
                          *    in registers (htf_build_eval_forces): the tensor is bit-identical and
                          *    available through htf_get_nlist_buffer, but never read back.
                          * 1: as 2 without writing the tensor (htf_fused_forces); the nlist side
                          *    buffer is then NOT filled.  Closed-form potentials; the pair-MLP
                          *    always takes the two-kernel route. */
} htf_config;

/* What HOOMD hands over each step (raw device pointers; see layouts above). */
typedef struct htf_hoomd_arrays {
    const void *pos;            /* Scalar4[N + n_ghost]                            */
    unsigned N;                 /* m_pdata->getN()                                  */
    unsigned n_ghost;           /* m_pdata->getNGhosts()                            */
    const unsigned *n_neigh;    /* NeighborList::getNNeighArray()                   */
    const unsigned *nlist;      /* NeighborList::getNListArray()                    */
    const unsigned *head_list;  /* NeighborList::getHeadList()                      */
    htf_box box;                /* m_pdata->getBox()                                */
    void *force;                /* Scalar4[N]  ForceCompute::m_force (written)      */
    void *virial;               /* Scalar[6*pitch] ForceCompute::m_virial (+=) or 0 */
    size_t virial_pitch;
} htf_hoomd_arrays;

typedef struct htf_ctx htf_ctx;

HTF_API int htf_create(const htf_config *cfg, htf_ctx **out);
HTF_API void htf_destroy(htf_ctx *ctx);
HTF_API int htf_set_potential(htf_ctx *ctx, const htf_potential *pot); /* borrowed; must outlive ctx use */
HTF_API int htf_resize(htf_ctx *ctx, unsigned max_n);                   /* reallocate(), .cc:88,91-121 */

/* TensorflowCompute<M>::computeForces(timestep) (.cc:129-216): period gate, batch
 * loop, pair-vector build, evaluation into force[offset..], virial fold-in.
 * Returns HTF_ERR_SKEWED_BOX / HTF_ERR_NLIST_OVERFLOW like the reference's asserts
 * (the overflow check reads one device word back and therefore synchronises the
 * stream; it only runs when cfg.check_nlist is set). */
HTF_API int htf_compute_forces(htf_ctx *ctx, unsigned timestep, const htf_hoomd_arrays *arrays, htf_stream stream);

/* The same call restricted to particle rows [row_begin, row_begin + row_count): lets a
 * domain-decomposed caller evaluate the rows that have no ghost neighbors while HOOMD's
 * Communicator (or hoomd_tf_amd/domain.py) is still refreshing ghost positions, and the
 * boundary rows afterwards.  With batch_size == 0 the context buffers (htf_get_*_buffer)
 * hold row i in slot i, so after the ranges of a step have been computed they describe the
 * whole step exactly as one htf_compute_forces call would. */
HTF_API int htf_compute_forces_rows(htf_ctx *ctx, unsigned timestep, const htf_hoomd_arrays *arrays,
                            unsigned row_begin, unsigned row_count, htf_stream stream);

/* Buffer getters (TensorflowCompute.cc:398-407 get*Buffer): device pointers of the
 * context-owned side buffers, for zero-copy views.  nlist: fp32 [B, NN, 4];
 * positions: fp32 [B, 4] (type un-stuffed); virial: Scalar [B, 9]. */
HTF_API void *htf_get_nlist_buffer(htf_ctx *ctx);
/* The nlist side buffer is READ-ONLY for the caller between steps: the context keeps every row as [live slots |
 * zeros] and re-zeroes only the slots a row has lost since the previous call (the padding, a quarter of the tensor
 * at 131 072 x 128, is not rewritten every step).  The reference lets a model write into this tensor (it is just a
 * TF tensor there); a caller that does so here must call htf_reset_nlist_buffer before the next htf_compute_forces:
 * the next build then rewrites every row's whole zero tail. */
HTF_API int htf_reset_nlist_buffer(htf_ctx *ctx, htf_stream stream);
HTF_API void *htf_get_positions_buffer(htf_ctx *ctx);
HTF_API void *htf_get_virial_buffer(htf_ctx *ctx);
HTF_API unsigned htf_get_batch_capacity(htf_ctx *ctx);

/* ------------------------------------------------------------------------- *
 * Ghost-position halo over RCCL (particle-domain decomposition).
 * Under HOOMD-blue the Communicator refreshes ghost positions every step and the plugin just reads them
 * (TensorflowCompute.cc:143-148); these entry points are that exchange for callers without HOOMD's
 * Communicator: one grouped ncclSend x2 / ncclRecv x2 per step on a halo stream of its own.
 * ------------------------------------------------------------------------- */
typedef struct htf_halo htf_halo;
#define HTF_HALO_ID_BYTES 128 /* sizeof(ncclUniqueId) */

HTF_API int htf_halo_available(void);          /* 1 when librccl could be loaded */
/* rank 0 obtains the id and ships its 128 bytes to every rank out of band (MPI_Bcast, torch.distributed ...) */
HTF_API int htf_halo_unique_id(void *id128);
/* collective over the `world` ranks: ncclCommInitRank on the calling thread's current device */
HTF_API int htf_halo_create(const void *id128, int rank, int world, htf_halo **out);
HTF_API void htf_halo_destroy(htf_halo *halo);
/* What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): the evidence a multi-rank
 * bench line carries that N ranks really sit on N devices (ABI 4). */
HTF_API int htf_halo_comm_info(htf_halo *halo, int *nranks, int *rank, int *device);
/* Send pos[send_left_first, +send_left_count) to rank `left` and pos[send_right_first, +count) to rank `right`;
 * receive the right neighbor's left-going message into pos[recv_right_first, +count) and the left neighbor's
 * right-going one into pos[recv_left_first, +count) (element = Scalar4 of `dtype`).  Starts after everything
 * already queued on `stream`; returns at once.  htf_halo_exchange_end makes `stream` wait for the arrival. */
HTF_API int htf_halo_exchange_begin(htf_halo *halo, void *d_pos, int dtype, int left, int right,
                            unsigned send_left_first, unsigned send_left_count,
                            unsigned send_right_first, unsigned send_right_count,
                            unsigned recv_left_first, unsigned recv_left_count,
                            unsigned recv_right_first, unsigned recv_right_count, htf_stream stream);
HTF_API int htf_halo_exchange_end(htf_halo *halo, htf_stream stream);

/* Any number of messages in one grouped exchange (2-D brick decompositions talk to 8 neighbors; the fixed-size migration
 * messages of a rebuild): sends posted in the order given, then receives in the order given -- the caller orders them so that
 * between any two ranks the k-th send meets the k-th receive.  async != 0: on the halo stream behind everything queued on
 * `stream`, ended by htf_halo_exchange_end; async == 0: on `stream` itself.  Plain stream work either way: a hipGraph capture of
 * `stream` records it. */
HTF_API int htf_halo_exchange_n(htf_halo *halo, int n_send, const void *const *send_ptrs, const size_t *send_bytes,
                                const int *send_peers, int n_recv, void *const *recv_ptrs, const size_t *recv_bytes,
                                const int *recv_peers, htf_stream stream, int async);
/* d_value[0..n) <- element-wise max over the ranks, in place, on `stream` (the all-reduced distance check). */
HTF_API int htf_halo_allreduce_max_f32(htf_halo *halo, float *d_value, unsigned n, htf_stream stream);

/* Profiler scopes (reference: HOOMD Profiler push/pop "TensorflowCompute::reshapeNeighbors"
 * and "TensorflowCompute::Force Update", TensorflowCompute.cc:164-168,196-206).  When
 * enabled, htf_compute_forces brackets the pair-vector build and the evaluator with
 * hipEvents on the caller's stream (`on` = k > 1: only every k-th batch, for loops where two event
 * records per step are a measurable share of the step; the one-kernel step is one scope, returned
 * as eval_ms); htf_profile_read synchronises on the recorded events, returns the summed kernel
 * milliseconds and the number of bracketed batches since the last read, and resets the accumulators. */
HTF_API int htf_profile_enable(htf_ctx *ctx, int on);
HTF_API int htf_profile_read(htf_ctx *ctx, double *build_ms, double *eval_ms, unsigned *n_calls);

#ifdef __cplusplus
}
#endif
#endif /* HTF_AMD_H_ */
