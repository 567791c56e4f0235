// Host side of the C ABI: error channel, potential handles, htf_eval_forces dispatch
// and the TensorflowCompute::computeForces driver (TensorflowCompute.cc:129-216)
// re-stated over raw HOOMD-layout device pointers with no Python, no TF and no
// device-wide synchronisation in the loop.
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <dlfcn.h>
#include <new>
#include <vector>

#include "htf_common.h"
#include "htf_internal.h"
#include "htf_standin.h"
#include "pair_mlp.h"
#include "pair_math.h"

namespace htf {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

} // namespace htf

// HOOMD Profiler scopes of the reference (TensorflowCompute.cc:139-140,164-168,196-206) as roctx
// ranges, visible in `rocprofv3 --marker-trace`.  librocprofiler-sdk-roctx is loaded on demand
// and only when HTF_ROCTX is set, so a normal run neither links nor calls it.
namespace htf {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        if (!getenv("HTF_ROCTX")) return;
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr;
    }
};
static Roctx &roctx() {
    static Roctx r;
    return r;
}
struct RoctxScope {
    bool on;
    explicit RoctxScope(const char *name) : on(roctx().push != nullptr) {
        if (on) roctx().push(name);
    }
    ~RoctxScope() {
        if (on) roctx().pop();
    }
};
} // namespace htf

struct htf_potential {
    htf::PotParams pp;
    htf::MlpDevice *mlp = nullptr;
    htf::TopkDevice *topk = nullptr;
    htf::JitKernels *jit = nullptr;
};

// potentials with an evaluator of their own (everything else goes through eval_pair_dispatch / the fused kernels)
static inline bool own_evaluator(const htf_potential *p) { return p->pp.kind == HTF_POT_PAIR_MLP || p->pp.kind == HTF_POT_TOPK_MLP; }

struct htf_ctx {
    htf_config cfg;
    const htf_potential *pot = nullptr;
    unsigned capacity = 0;     // rows the scratch can hold (batch_size or max_n)
    float4 *nlist = nullptr;   // [capacity, NN] fp32 pair vectors  (m_nlist_array, .cc:113)
    float4 *positions = nullptr; // [capacity] fp32 positions, type un-stuffed (m_positions_array, .cc:100)
    void *virial = nullptr;    // [capacity, 9] Scalar              (m_virial_array, .cc:117)
    unsigned *flag = nullptr;  // device word for check_nlist / overflow counts
    unsigned *counts = nullptr; // [capacity] live slots per scratch row (delta zero-fill + padding skip)
    // profiler scopes: event triples (before build, between, after eval) per batch
    unsigned profiling = 0;      // 0 = off, k = bracket every k-th htf_compute_forces batch
    unsigned prof_tick = 0;
    std::vector<hipEvent_t> ev_pool;
    std::vector<char> ev_one_scope; // per triple: the one-kernel step records no middle event
    std::vector<char> ev_complete;  // per triple: the closing event was recorded (an error return in between leaves it 0)
    size_t ev_used = 0;
    // the stand-in integrator as the one-kernel step's epilogue (htfs_set_step_epilogue)
    void *d_epilogue[HTFS_EPILOGUE_SLOTS] = {nullptr, nullptr}; // StepEpilogue<Scalar> descriptors in device memory
    bool epilogue_ok[HTFS_EPILOGUE_SLOTS] = {false, false};
    int epilogue_level[HTFS_EPILOGUE_SLOTS] = {0, 0};          // 1: the integrator alone, 2: + a brick's halo messages
    int epilogue_slot = -1;                                     // what the next compute call carries
};

static hipEvent_t next_event(htf_ctx *c) {
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev_pool.push_back(e);
    }
    return c->ev_pool[c->ev_used++];
}

extern "C" const char *htf_last_error(void) { return htf::g_last_error.c_str(); }
extern "C" int htf_abi_version(void) { return HTF_AMD_ABI_VERSION; }

extern "C" int htf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------ potentials
extern "C" int htf_potential_create(const htf_potential_desc *d, htf_potential **out) {
    using namespace htf;
    HTF_REQUIRE(d && out, "htf_potential_create: null pointer");
    htf_potential *p = new (std::nothrow) htf_potential();
    if (!p) {
        set_error("htf_potential_create: out of host memory");
        return HTF_ERR_NOMEM;
    }
    std::memset(&p->pp, 0, sizeof p->pp);
    p->pp.kind = d->kind;
    int rc = HTF_OK;
    switch (d->kind) {
    case HTF_POT_LJ:
    case HTF_POT_SIMPLE:
        break;
    case HTF_POT_WCA:
        if (!(d->sigma > 0)) {
            set_error("htf_potential_create: WCA sigma must be > 0 (got %g)", d->sigma);
            rc = HTF_ERR_INVALID;
        }
        p->pp.sigma = (float)d->sigma;
        // layers.py:97 `true_sig * 2**(1/3)`: fp32 weight times the python double rounded to fp32
        p->pp.wca_cut = p->pp.sigma * (float)1.2599210498948732;
        p->pp.wca_cut_r2 = sqrt_threshold(p->pp.wca_cut);
        break;
    case HTF_POT_RINV_POLY:
        if (d->n_terms < 1 || d->n_terms > HTF_MAX_POLY_TERMS) {
            set_error("htf_potential_create: n_terms %d outside [1, %d]", d->n_terms, HTF_MAX_POLY_TERMS);
            rc = HTF_ERR_INVALID;
            break;
        }
        p->pp.n_terms = d->n_terms;
        for (int k = 0; k < d->n_terms; ++k) {
            if (d->power[k] < 1 || d->power[k] > 64) {
                set_error("htf_potential_create: rinv power %d outside [1, 64]", d->power[k]);
                rc = HTF_ERR_INVALID;
            }
            p->pp.coef[k] = (float)d->coef[k];
            p->pp.power[k] = d->power[k];
        }
        if (d->poly_cut < 0) {
            set_error("htf_potential_create: poly_cut must be >= 0 (got %g)", d->poly_cut);
            rc = HTF_ERR_INVALID;
        }
        // `r < cut` on r = tf.norm (correctly rounded sqrt of the plain sum of squares) as a threshold on the squared norm
        p->pp.poly_cut_r2 = d->poly_cut > 0 ? sqrt_threshold((float)d->poly_cut) : 0.0f;
        break;
    case HTF_POT_LJ_PARAM:
        p->pp.lj_w0 = (float)d->lj_w0;
        p->pp.lj_w1 = (float)d->lj_w1;
        break;
    case HTF_POT_GAUSS:
        if (!(d->gauss_gap > 0)) {
            set_error("htf_potential_create: GAUSS gap must be > 0 (got %g)", d->gauss_gap);
            rc = HTF_ERR_INVALID;
        }
        p->pp.gauss_r0 = (float)d->gauss_r0;
        p->pp.gauss_ginv = 1.0f / (float)d->gauss_gap;
        p->pp.gauss_coef = (float)d->gauss_coef;
        p->pp.gauss_k_exp = -1.4426950408889634f * p->pp.gauss_ginv;
        p->pp.gauss_k_force = -4.0f * p->pp.gauss_coef * p->pp.gauss_ginv;
        break;
    case HTF_POT_PAIR_MLP:
        rc = mlp_create(d, &p->mlp);
        break;
    case HTF_POT_TOPK_MLP:
        rc = topk_create(d, &p->topk);
        break;
    case HTF_POT_JIT:
        rc = jit_create(d->jit_image, d->jit_image_bytes, &p->jit);
        p->pp.jit = p->jit;
        p->pp.jit_flags = d->jit_flags;
        p->pp.n_terms = rc == HTF_OK ? jit_num_params(p->jit) : 0; // (weights the unit was compiled for: 0 = a weight-free energy)
        break;
    default:
        set_error("htf_potential_create: unknown potential kind %d", d->kind);
        rc = HTF_ERR_INVALID;
    }
    if (rc != HTF_OK) {
        delete p;
        return rc;
    }
    if (d->d_theta != nullptr && d->kind != HTF_POT_PAIR_MLP && d->kind != HTF_POT_TOPK_MLP) { // (the pair-MLP keeps theta in its own record)
        if (potential_num_params(p->pp) == 0) {
            set_error("htf_potential_create: potential kind %d has no trainable parameters", d->kind);
            delete p;
            return HTF_ERR_INVALID;
        }
        p->pp.theta = d->d_theta;
    }
    *out = p;
    return HTF_OK;
}

extern "C" int htf_potential_num_params(const htf_potential *pot) {
    if (!pot) return 0;
    return pot->mlp ? pot->mlp->num_params() : htf::potential_num_params(pot->pp);
}

extern "C" size_t htf_train_scratch_floats(const htf_potential *pot, unsigned B, unsigned NN) {
    if (!pot) return 0;
    return pot->mlp ? htf::mlp_train_scratch_floats(pot->mlp, B, NN) : htf::train_scratch_floats(pot->pp, B, NN);
}

extern "C" int htf_potential_refresh(htf_potential *pot, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(pot, "htf_potential_refresh: no potential");
    if (!pot->mlp) return HTF_OK; // closed forms read theta directly
    return mlp_refresh(pot->mlp, (hipStream_t)stream);
}

extern "C" int htf_train_pair_grad(const htf_potential *pot, const void *d_nlist, int nlist_dtype, unsigned B,
                                   unsigned NN, const void *d_labels, int label_dtype, void *d_pred, float *d_accum,
                                   float *d_scratch, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(pot, "htf_train_pair_grad: no potential");
    HTF_REQUIRE(d_nlist && d_labels && d_accum && d_scratch, "htf_train_pair_grad: null pointer");
    HTF_REQUIRE(NN > 0 && B > 0, "htf_train_pair_grad: empty batch");
    HTF_REQUIRE(nlist_dtype == HTF_F32 || nlist_dtype == HTF_F64, "htf_train_pair_grad: bad nlist dtype %d", nlist_dtype);
    HTF_REQUIRE(label_dtype == HTF_F32 || label_dtype == HTF_F64, "htf_train_pair_grad: bad label dtype %d", label_dtype);
    if (pot->mlp)
        return mlp_train_grad(pot->mlp, d_nlist, nlist_dtype, B, NN, d_labels, label_dtype == HTF_F64, d_pred, d_accum,
                              d_scratch, (hipStream_t)stream);
    return train_pair_dispatch(pot->pp, d_nlist, nlist_dtype, B, NN, d_labels, label_dtype, d_pred, d_accum, d_scratch,
                               (hipStream_t)stream);
}

extern "C" int htf_train_pair_grad_list(const htf_potential *pot, const void *d_pos, int pos_dtype, unsigned B, unsigned NN,
                                        const htf_box *box, const unsigned *d_n_neigh, const unsigned *d_nlist,
                                        const unsigned *d_head_list, double rmax, const void *d_labels, int label_dtype,
                                        void *d_pred, float *d_accum, float *d_scratch, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(pot, "htf_train_pair_grad_list: no potential");
    HTF_REQUIRE(d_pos && box && d_n_neigh && d_nlist && d_head_list && d_labels && d_accum && d_scratch, "htf_train_pair_grad_list: null pointer");
    HTF_REQUIRE(NN > 0 && B > 0, "htf_train_pair_grad_list: empty batch");
    HTF_REQUIRE(pos_dtype == HTF_F32 || pos_dtype == HTF_F64, "htf_train_pair_grad_list: bad position dtype %d", pos_dtype);
    HTF_REQUIRE(label_dtype == HTF_F32 || label_dtype == HTF_F64, "htf_train_pair_grad_list: bad label dtype %d", label_dtype);
    HTF_REQUIRE(!pot->mlp, "htf_train_pair_grad_list: the pair-MLP's training sweep reads the pair-vector tensor (htf_train_pair_grad)");
    return train_list_dispatch(pot->pp, d_pos, pos_dtype, B, NN, box, d_n_neigh, d_nlist, d_head_list, rmax, d_labels, label_dtype,
                               d_pred, d_accum, d_scratch, (hipStream_t)stream);
}

extern "C" void htf_potential_destroy(htf_potential *pot) {
    if (!pot) return;
    if (pot->mlp) htf::mlp_destroy(pot->mlp);
    if (pot->topk) htf::topk_destroy(pot->topk);
    if (pot->jit) htf::jit_destroy(pot->jit);
    delete pot;
}

extern "C" int htf_eval_forces(const htf_potential *pot, const void *d_nlist, int nlist_dtype, unsigned B,
                               unsigned NN, void *d_force, int force_dtype, void *d_virial9, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(pot, "htf_eval_forces: no potential");
    HTF_REQUIRE(d_nlist && d_force, "htf_eval_forces: null pointer");
    HTF_REQUIRE(NN > 0, "htf_eval_forces: NN must be > 0");
    HTF_REQUIRE(nlist_dtype == HTF_F32 || nlist_dtype == HTF_F64, "htf_eval_forces: bad nlist dtype %d", nlist_dtype);
    HTF_REQUIRE(force_dtype == HTF_F32 || force_dtype == HTF_F64, "htf_eval_forces: bad force dtype %d", force_dtype);
    if (B == 0) return HTF_OK;
    if (pot->pp.kind == HTF_POT_PAIR_MLP)
        return mlp_eval(pot->mlp, d_nlist, nlist_dtype, B, NN, d_force, force_dtype, d_virial9, (hipStream_t)stream);
    if (pot->pp.kind == HTF_POT_TOPK_MLP)
        return topk_eval(pot->topk, d_nlist, nlist_dtype, B, NN, d_force, force_dtype, d_virial9, (hipStream_t)stream);
    HTF_REQUIRE(!(pot->pp.kind == HTF_POT_SIMPLE && d_virial9), "htf_eval_forces: SimplePotential has no energy, hence no virial");
    HTF_REQUIRE(!(pot->pp.kind == HTF_POT_JIT && (pot->pp.jit_flags & HTF_JIT_READS_OWN_TYPE)),
                "htf_eval_forces: this traced energy reads the row particle's own type: evaluate it with htf_eval_forces_typed "
                "(positions beside the pair vectors) or through the one-kernel step");
    return eval_pair_dispatch(pot->pp, d_nlist, nlist_dtype, B, NN, d_force, force_dtype, d_virial9, nullptr, (hipStream_t)stream);
}

extern "C" int htf_eval_forces_typed(const htf_potential *pot, const void *d_nlist, int nlist_dtype, unsigned B, unsigned NN,
                                     const void *d_positions, int positions_dtype, void *d_force, int force_dtype, void *d_virial9,
                                     htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(pot, "htf_eval_forces_typed: no potential");
    if (!(pot->pp.kind == HTF_POT_JIT && (pot->pp.jit_flags & HTF_JIT_READS_OWN_TYPE)))
        return htf_eval_forces(pot, d_nlist, nlist_dtype, B, NN, d_force, force_dtype, d_virial9, stream);
    HTF_REQUIRE(d_nlist && d_force && d_positions, "htf_eval_forces_typed: null pointer");
    HTF_REQUIRE(NN > 0, "htf_eval_forces_typed: NN must be > 0");
    HTF_REQUIRE(nlist_dtype == HTF_F32 || nlist_dtype == HTF_F64, "htf_eval_forces_typed: bad nlist dtype %d", nlist_dtype);
    HTF_REQUIRE(positions_dtype == HTF_F32 || positions_dtype == HTF_F64, "htf_eval_forces_typed: bad positions dtype %d", positions_dtype);
    HTF_REQUIRE(force_dtype == HTF_F32 || force_dtype == HTF_F64, "htf_eval_forces_typed: bad force dtype %d", force_dtype);
    if (B == 0) return HTF_OK;
    PotParams pp = pot->pp;
    pp.own = d_positions;
    pp.own_f64 = positions_dtype == HTF_F64 ? 1 : 0;
    return eval_pair_dispatch(pp, d_nlist, nlist_dtype, B, NN, d_force, force_dtype, d_virial9, nullptr, (hipStream_t)stream);
}

extern "C" int htf_fused_forces(const htf_potential *pot, const void *d_pos, int pos_dtype, unsigned N, unsigned NN,
                                unsigned offset, unsigned batch_size, const htf_box *box, const unsigned *d_n_neigh,
                                const unsigned *d_nlist, const unsigned *d_head_list, double rmax, void *d_force,
                                int force_dtype, void *d_virial9, unsigned *d_check_count, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(pot, "htf_fused_forces: no potential");
    HTF_REQUIRE(force_dtype == HTF_F32 || force_dtype == HTF_F64, "htf_fused_forces: bad force dtype %d", force_dtype);
    return fused_forces_impl(pot->pp, d_pos, pos_dtype, N, NN, offset, batch_size, box, d_n_neigh, d_nlist, d_head_list,
                             rmax, d_force, force_dtype, d_virial9, d_check_count, nullptr, nullptr, nullptr,
                             (hipStream_t)stream);
}

extern "C" int htf_build_eval_forces(const htf_potential *pot, void *d_dest, const void *d_pos, int pos_dtype, unsigned N,
                                     unsigned NN, unsigned offset, unsigned batch_size, const htf_box *box,
                                     const unsigned *d_n_neigh, const unsigned *d_nlist, const unsigned *d_head_list,
                                     double rmax, void *d_force, int force_dtype, void *d_virial9,
                                     unsigned *d_check_count, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(pot, "htf_build_eval_forces: no potential");
    HTF_REQUIRE(d_dest, "htf_build_eval_forces: null pair-vector tensor");
    HTF_REQUIRE(force_dtype == HTF_F32 || force_dtype == HTF_F64, "htf_build_eval_forces: bad force dtype %d", force_dtype);
    return fused_forces_impl(pot->pp, d_pos, pos_dtype, N, NN, offset, batch_size, box, d_n_neigh, d_nlist, d_head_list,
                             rmax, d_force, force_dtype, d_virial9, d_check_count, nullptr, (float4 *)d_dest, nullptr,
                             (hipStream_t)stream);
}

extern "C" int htf_eval_forces2(const htf_potential *potA, const htf_potential *potB, const void *d_nlist,
                                int nlist_dtype, unsigned B, unsigned NN, void *d_forceA, void *d_forceB,
                                int force_dtype, float *d_partials, float rdf_r0, float rdf_r1,
                                unsigned rdf_nbins_total, unsigned *d_rdf_hist, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(potA && potB, "htf_eval_forces2: no potential");
    HTF_REQUIRE(d_nlist && d_forceA && d_forceB, "htf_eval_forces2: null pointer");
    HTF_REQUIRE(NN > 0, "htf_eval_forces2: NN must be > 0");
    HTF_REQUIRE(nlist_dtype == HTF_F32 || nlist_dtype == HTF_F64, "htf_eval_forces2: bad nlist dtype %d", nlist_dtype);
    HTF_REQUIRE(force_dtype == HTF_F32 || force_dtype == HTF_F64, "htf_eval_forces2: bad force dtype %d", force_dtype);
    HTF_REQUIRE(potB->pp.kind == HTF_POT_GAUSS, "htf_eval_forces2: potB must be HTF_POT_GAUSS");
    if (B == 0) return HTF_OK;
    return eval_pair2_dispatch(potA->pp, potB->pp, d_nlist, nlist_dtype, B, NN, d_forceA, d_forceB, force_dtype,
                               d_partials, rdf_r0, rdf_r1, rdf_nbins_total, d_rdf_hist, (hipStream_t)stream);
}

extern "C" unsigned htf_eval2_num_partials(unsigned B, unsigned NN) { return htf::eval_pair2_num_partials(B, NN); }

extern "C" int htf_build_eval_forces2(const htf_potential *potA, const htf_potential *potB, void *d_dest, const void *d_pos,
                                      int pos_dtype, unsigned N, unsigned NN, unsigned offset, unsigned batch_size,
                                      const htf_box *box, const unsigned *d_n_neigh, const unsigned *d_nlist,
                                      const unsigned *d_head_list, double rmax, void *d_forceA, void *d_forceB,
                                      int force_dtype, float *d_partials, float rdf_r0, float rdf_r1,
                                      unsigned rdf_nbins_total, unsigned *d_rdf_hist, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(potA && potB, "htf_build_eval_forces2: no potential");
    HTF_REQUIRE(force_dtype == HTF_F32 || force_dtype == HTF_F64, "htf_build_eval_forces2: bad force dtype %d", force_dtype);
    return fused_forces2_impl(potA->pp, potB->pp, d_pos, pos_dtype, N, NN, offset, batch_size, box, d_n_neigh, d_nlist,
                              d_head_list, rmax, d_forceA, d_forceB, force_dtype, d_partials, rdf_r0, rdf_r1,
                              rdf_nbins_total, d_rdf_hist, (float4 *)d_dest, nullptr, (hipStream_t)stream);
}

extern "C" unsigned htf_build_eval2_num_partials(unsigned batch_size) { return htf::fused_forces2_num_partials(batch_size); }

// ------------------------------------------------------------------------------ context
static void ctx_free(htf_ctx *c) {
    if (c->nlist) (void)hipFree(c->nlist);
    if (c->positions) (void)hipFree(c->positions);
    if (c->virial) (void)hipFree(c->virial);
    if (c->counts) (void)hipFree(c->counts);
    c->counts = nullptr;
    c->nlist = nullptr;
    c->positions = nullptr;
    c->virial = nullptr;
    c->capacity = 0;
}

// TensorflowCompute::reallocate (.cc:91-121): side buffers sized by batch_size, or by
// getMaxN() when unbatched; virial zeroed once (memsetArray(0), :120).
static int ctx_alloc(htf_ctx *c, unsigned max_n) {
    using namespace htf;
    ctx_free(c);
    unsigned cap = c->cfg.batch_size ? c->cfg.batch_size : max_n;
    if (cap == 0) cap = 1;
    size_t ssz = c->cfg.scalar_dtype == HTF_F64 ? 8 : 4;
    if (c->cfg.nneighs > 0) {
        // zeroed once: from then on every row is [live slots | zeros] and the build only
        // re-zeroes the slots a row has lost since the previous call (counts)
        HTF_CHECK_HIP(hipMalloc((void **)&c->nlist, (size_t)cap * c->cfg.nneighs * sizeof(float4)));
        HTF_CHECK_HIP(hipMemset(c->nlist, 0, (size_t)cap * c->cfg.nneighs * sizeof(float4)));
        HTF_CHECK_HIP(hipMalloc((void **)&c->counts, (size_t)cap * sizeof(unsigned)));
        HTF_CHECK_HIP(hipMemset(c->counts, 0, (size_t)cap * sizeof(unsigned)));
    }
    HTF_CHECK_HIP(hipMalloc((void **)&c->positions, (size_t)cap * sizeof(float4)));
    HTF_CHECK_HIP(hipMalloc(&c->virial, (size_t)cap * 9 * ssz));
    HTF_CHECK_HIP(hipMemset(c->virial, 0, (size_t)cap * 9 * ssz));
    c->capacity = cap;
    c->cfg.max_n = max_n;
    return HTF_OK;
}

extern "C" int htf_create(const htf_config *cfg, htf_ctx **out) {
    using namespace htf;
    HTF_REQUIRE(cfg && out, "htf_create: null pointer");
    HTF_REQUIRE(cfg->period >= 1, "htf_create: period must be >= 1");
    HTF_REQUIRE(cfg->scalar_dtype == HTF_F32 || cfg->scalar_dtype == HTF_F64, "htf_create: bad scalar dtype %d", cfg->scalar_dtype);
    HTF_REQUIRE(cfg->force_mode == HTF_TF2HOOMD || cfg->force_mode == HTF_HOOMD2TF, "htf_create: bad force mode %d", cfg->force_mode);
    HTF_REQUIRE(cfg->nneighs == 0 || cfg->r_cut > 0, "htf_create: r_cut must be > 0 when nneighs > 0");
    htf_ctx *c = new (std::nothrow) htf_ctx();
    if (!c) {
        set_error("htf_create: out of host memory");
        return HTF_ERR_NOMEM;
    }
    c->cfg = *cfg;
    hipError_t e = hipMalloc((void **)&c->flag, sizeof(unsigned));
    if (e != hipSuccess) {
        set_error("htf_create: hipMalloc failed: %s", hipGetErrorString(e));
        delete c;
        return HTF_ERR_DEVICE;
    }
    int rc = ctx_alloc(c, cfg->max_n);
    if (rc != HTF_OK) {
        htf_destroy(c);
        return rc;
    }
    *out = c;
    return HTF_OK;
}

extern "C" void htf_destroy(htf_ctx *ctx) {
    if (!ctx) return;
    ctx_free(ctx);
    if (ctx->flag) (void)hipFree(ctx->flag);
    for (void *d : ctx->d_epilogue)
        if (d) (void)hipFree(d);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    delete ctx;
}

extern "C" int htf_set_potential(htf_ctx *ctx, const htf_potential *pot) {
    using namespace htf;
    HTF_REQUIRE(ctx, "htf_set_potential: null context");
    ctx->pot = pot;
    // (a registered step epilogue was judged against the potential it was registered under: register it again)
    ctx->epilogue_slot = -1;
    for (bool &ok : ctx->epilogue_ok) ok = false;
    return HTF_OK;
}

extern "C" int htf_resize(htf_ctx *ctx, unsigned max_n) {
    using namespace htf;
    HTF_REQUIRE(ctx, "htf_resize: null context");
    return ctx_alloc(ctx, max_n);
}

extern "C" void *htf_get_nlist_buffer(htf_ctx *ctx) { return ctx ? ctx->nlist : nullptr; }

extern "C" int htf_reset_nlist_buffer(htf_ctx *ctx, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(ctx, "htf_reset_nlist_buffer: null context");
    if (!ctx->counts || ctx->capacity == 0) return HTF_OK;
    // every row is declared fully live: the next build rewrites the whole zero tail of every row
    HTF_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)ctx->counts, (int)ctx->cfg.nneighs, ctx->capacity, (hipStream_t)stream));
    return HTF_OK;
}
extern "C" void *htf_get_positions_buffer(htf_ctx *ctx) { return ctx ? ctx->positions : nullptr; }
extern "C" void *htf_get_virial_buffer(htf_ctx *ctx) { return ctx ? ctx->virial : nullptr; }
extern "C" unsigned htf_get_batch_capacity(htf_ctx *ctx) { return ctx ? ctx->capacity : 0; }

// Rows [row_begin, row_begin + row_count) of one computeForces call.  batch_size == 0: the scratch
// holds all N rows, row i in slot i, so a step may be computed in several row ranges (interior
// rows while the ghost halo is in flight, boundary rows after it) and the buffers still end
// up holding the whole step.  batch_size > 0: slot = row - batch offset, as upstream.
static int compute_rows(htf_ctx *ctx, unsigned timestep, const htf_hoomd_arrays *a, unsigned row_begin,
                        unsigned row_count, htf_stream stream) {
    using namespace htf;
    HTF_REQUIRE(ctx && a, "htf_compute_forces: null pointer");
    const htf_config &cfg = ctx->cfg;
    if (timestep % cfg.period != 0) return HTF_OK; // .cc:133 -- previous forces stay in m_force
    HTF_REQUIRE(a->pos, "htf_compute_forces: null positions");
    // FORCE_MODE::hoomd2tf (TensorflowCompute.cc:177-187): forces flow FROM HOOMD to the model as training
    // labels; this call then only stages the model inputs (pair vectors, positions) and writes no force --
    // the training sweep (htf_train_pair_grad on htf_get_nlist_buffer) is the caller's next call
    const bool to_hoomd = cfg.force_mode == HTF_TF2HOOMD;
    HTF_REQUIRE(a->force || !to_hoomd, "htf_compute_forces: null force array");
    // SimModel.compute_inputs: tf.Assert(reduce_sum(box[2]) < 0.0001)  simmodel.py:195
    if (!(a->box.tilt[0] + a->box.tilt[1] + a->box.tilt[2] < 0.0001)) {
        set_error("box is skewed");
        return HTF_ERR_SKEWED_BOX;
    }
    const unsigned N = a->N;
    if (N == 0 || row_count == 0) return HTF_OK;
    HTF_REQUIRE(row_begin < N && row_count <= N - row_begin, "htf_compute_forces_rows: rows [%u, +%u) outside [0, %u)", row_begin, row_count, N);
    if (cfg.batch_size == 0 && N > ctx->capacity) {
        int rc = ctx_alloc(ctx, N); // MaxParticleNumberChange -> reallocate (.cc:88)
        if (rc != HTF_OK) return rc;
    }
    const size_t ssz = cfg.scalar_dtype == HTF_F64 ? 8 : 4;
    hipStream_t s = (hipStream_t)stream;
    RoctxScope scope_all("TensorflowCompute");
    const unsigned bs = cfg.batch_size == 0 ? row_count : cfg.batch_size;
    const unsigned row_end = row_begin + row_count;
    for (unsigned offset = row_begin; offset < row_end; offset += bs) { // .cc:143
        const unsigned n = std::min(row_end - offset, bs);
        const size_t slot0 = cfg.batch_size == 0 ? offset : 0;
        float4 *c_nlist = ctx->nlist ? ctx->nlist + slot0 * cfg.nneighs : nullptr;
        float4 *c_positions = ctx->positions + slot0;
        unsigned *c_counts = ctx->counts ? ctx->counts + slot0 : nullptr;
        void *c_virial = (char *)ctx->virial + slot0 * 9 * ssz;
        int rc;
        const bool prof = ctx->profiling && cfg.nneighs > 0 && ctx->pot != nullptr && (ctx->prof_tick++ % ctx->profiling) == 0;
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
        if (prof) {
            e0 = next_event(ctx);
            e1 = next_event(ctx);
            e2 = next_event(ctx);
            HTF_REQUIRE(e0 && e1 && e2, "htf_compute_forces: hipEventCreate failed");
            ctx->ev_one_scope.push_back(0);
            ctx->ev_complete.push_back(0);
        }
        const bool fused = to_hoomd && cfg.fused && cfg.nneighs > 0 && ctx->pot != nullptr && !own_evaluator(ctx->pot);
        // the one-kernel step is timed by the launch itself (kernel begin / end, as rocprofv3 reports it); the two-kernel
        // path by events recorded around its kernels
        if (prof && !fused) HTF_CHECK_HIP(hipEventRecord(e0, s));
        if (fused) {
            HTF_REQUIRE(a->n_neigh && a->nlist && a->head_list, "htf_compute_forces: null neighbor list");
            if (prof) {
                ctx->ev_one_scope.back() = 1; // no separate build scope in fused mode: e1 stays unrecorded
                launch_events() = LaunchEvents{e0, e2, false};
            }
            if (cfg.check_nlist) HTF_CHECK_HIP(hipMemsetAsync(ctx->flag, 0, sizeof(unsigned), s));
            void *fo = (char *)a->force + (size_t)offset * 4 * ssz;
            // (the epilogue rides on this launch only: the request is this thread's, cleared behind the call)
            step_epilogue_request() = ctx->epilogue_slot >= 0 ? ctx->d_epilogue[ctx->epilogue_slot] : nullptr;
            step_epilogue_level() = ctx->epilogue_slot >= 0 ? ctx->epilogue_level[ctx->epilogue_slot] : 0;
            rc = fused_forces_impl(ctx->pot->pp, a->pos, cfg.scalar_dtype, N, cfg.nneighs, offset, n, &a->box, a->n_neigh,
                                   a->nlist, a->head_list, cfg.r_cut, fo, cfg.scalar_dtype,
                                   cfg.virial ? c_virial : nullptr, cfg.check_nlist ? ctx->flag : nullptr,
                                   c_positions, cfg.fused == 2 ? c_nlist : nullptr, cfg.fused == 2 ? c_counts : nullptr, s);
            step_epilogue_request() = nullptr;
            const bool stamped = launch_events().used;
            launch_events() = LaunchEvents{};
            if (rc != HTF_OK) return rc;
            if (prof) ctx->ev_complete.back() = stamped ? 1 : 0; // (a launch form that took no events: the scope is skipped)
            if (cfg.check_nlist) {
                unsigned h = 0;
                HTF_CHECK_HIP(hipMemcpyAsync(&h, ctx->flag, sizeof(unsigned), hipMemcpyDeviceToHost, s));
                HTF_CHECK_HIP(hipStreamSynchronize(s));
                if (!(h < cfg.nneighs)) {
                    set_error("Neighbor list is full!");
                    return HTF_ERR_NLIST_OVERFLOW;
                }
            }
            if (cfg.virial && a->virial) {
                rc = htf_add_virial((char *)a->virial + (size_t)offset * ssz, c_virial, cfg.scalar_dtype, n,
                                    a->virial_pitch, stream);
                if (rc != HTF_OK) return rc;
            }
            continue;
        }
        if (cfg.nneighs > 0) {
            HTF_REQUIRE(a->n_neigh && a->nlist && a->head_list, "htf_compute_forces: null neighbor list");
            // positions side buffer (m_positions_comm.receiveArray, .cc:172) is staged by the same kernel
            RoctxScope scope_build("TensorflowCompute::reshapeNeighbors");
            rc = build_pair_vectors_impl(c_nlist, HTF_F32, a->pos, cfg.scalar_dtype, N, cfg.nneighs, offset, n,
                                         &a->box, a->n_neigh, a->nlist, a->head_list, cfg.r_cut, nullptr,
                                         c_positions, c_counts, s);
            if (rc != HTF_OK) return rc;
        }
        if (prof) HTF_CHECK_HIP(hipEventRecord(e1, s));
        if (cfg.nneighs == 0) { // positions-only models (nneighbor_cutoff = 0) live above the ABI
            rc = htf_copy_positions(c_positions, HTF_F32, a->pos, cfg.scalar_dtype, offset, n, 1, stream);
            if (rc != HTF_OK) return rc;
            continue;
        }
        if (ctx->pot == nullptr || !to_hoomd) continue;
        if (cfg.check_nlist) {
            unsigned h = 0;
            HTF_CHECK_HIP(hipMemsetAsync(ctx->flag, 0, sizeof(unsigned), s));
            rc = htf_check_nlist(c_nlist, HTF_F32, n, cfg.nneighs, ctx->flag, stream);
            if (rc != HTF_OK) return rc;
            HTF_CHECK_HIP(hipMemcpyAsync(&h, ctx->flag, sizeof(unsigned), hipMemcpyDeviceToHost, s));
            HTF_CHECK_HIP(hipStreamSynchronize(s));
            if (!(h < cfg.nneighs)) { // tf.debugging.assert_less(NN, nneighbor_cutoff)  simmodel.py:220-224
                set_error("Neighbor list is full!");
                return HTF_ERR_NLIST_OVERFLOW;
            }
        }
        RoctxScope scope_eval("TensorflowCompute::Force Update");
        void *force_out = (char *)a->force + (size_t)offset * 4 * ssz; // m_forces_comm.setOffset(offset) .cc:192
        if (own_evaluator(ctx->pot))
            rc = htf_eval_forces(ctx->pot, c_nlist, HTF_F32, n, cfg.nneighs, force_out, cfg.scalar_dtype,
                                 cfg.virial ? c_virial : nullptr, stream);
        else {
            PotParams pp = ctx->pot->pp;
            pp.own = c_positions; // (the positions side buffer, staged with the pair vectors: what a typed generated body reads)
            pp.own_f64 = 0;
            rc = eval_pair_dispatch(pp, c_nlist, HTF_F32, n, cfg.nneighs, force_out, cfg.scalar_dtype,
                                    cfg.virial ? c_virial : nullptr, c_counts, s);
        }
        if (rc != HTF_OK) return rc;
        if (prof) {
            HTF_CHECK_HIP(hipEventRecord(e2, s));
            ctx->ev_complete.back() = 1;
        }
        if (cfg.virial && a->virial) { // receiveVirial(offset, N) .cc:200-204
            rc = htf_add_virial((char *)a->virial + (size_t)offset * ssz, c_virial, cfg.scalar_dtype, n,
                                a->virial_pitch, stream);
            if (rc != HTF_OK) return rc;
        }
    }
    return HTF_OK;
}

namespace htf {
// htfs_step_epilogue (host struct, include/htf_standin.h) -> the kernel argument.  The integrator's box is standin_gate.h
// make_sbox's three expressions on the caller's htf_box, the halo's wrap box brick.hip make_args': the same bits as the separate
// kernels.  
template <typename T>
static StepEpilogue<T> make_step_epilogue(const htfs_step_epilogue *r) {
    const htf_box *hb = &r->box;
    StepEpilogue<T> e;
    for (int d = 0; d < 3; ++d) {
        e.lo[d] = (T)hb->lo[d];
        e.L[d] = (T)hb->hi[d] - (T)hb->lo[d];
        e.Linv[d] = (T)1 / e.L[d];
        e.periodic[d] = hb->periodic[d];
    }
    for (int m = 0; m < 8; ++m) {
        e.ghost_off[m] = e.ghost_off_opp[m] = 0;
        for (int c = 0; c < 3; ++c) e.shift[m][c] = (T)0;
    }
    e.vel = r->d_vel;
    e.pos_next = r->d_pos_next;
    e.dt = (T)r->dt;
    if (r->brick != nullptr && r->d_row_slots != nullptr && (r->d_halo_send != nullptr || r->d_ghost_direct != nullptr)) {
        const htfs_brick *g = r->brick;
        e.row_slots = r->d_row_slots;
        e.send = r->d_halo_send;
        e.direct = r->d_ghost_direct;
        e.cap_int = g->cap_int;
        e.n_msg = g->n_msg;
        e.halo_wrap = g->halo_wrap;
        for (int m = 0; m < HTFS_BRICK_MAX_MSG && m < 8; ++m) {
            e.ghost_off[m] = g->ghost_off[m];
            e.ghost_off_opp[m] = m < g->n_msg ? g->ghost_off[g->n_msg - 1 - m] : 0u;
            for (int c = 0; c < 3; ++c) e.shift[m][c] = (T)g->shift[m][c];
        }
    }
    return e;
}
} // namespace htf

extern "C" int htfs_set_step_epilogue(htf_ctx *ctx, int slot, const htfs_step_epilogue *ep, int *applies) {
    using namespace htf;
    HTF_REQUIRE(ctx && ep, "htfs_set_step_epilogue: null pointer");
    HTF_REQUIRE(slot >= 0 && slot < HTFS_EPILOGUE_SLOTS, "htfs_set_step_epilogue: slot %d outside [0, %d)", slot, HTFS_EPILOGUE_SLOTS);
    if (applies) *applies = 0;
    HTF_REQUIRE(ep->d_vel && ep->d_pos_next && (ep->dtype == HTF_F32 || ep->dtype == HTF_F64), "htfs_set_step_epilogue: null pointer or bad dtype");
    HTF_REQUIRE(ep->brick == nullptr || (ep->d_row_slots && (ep->d_halo_send || ep->d_ghost_direct)),
                "htfs_set_step_epilogue: a brick needs its row slots and a destination for the messages");
    const htf_config &cfg = ctx->cfg;
    const htf_potential *pot = ctx->pot;
    // honoured by: the one-kernel route of LJModel / WCARepulsion on fp32 positions (the forms compiled with an epilogue), no virial
    // (the one-row kernel), unbatched, forces written into HOOMD's array, everything in the context's Scalar
    const bool ok = cfg.force_mode == HTF_TF2HOOMD && cfg.fused != 0 && cfg.nneighs > 0 && !cfg.virial && cfg.batch_size == 0 &&
                    cfg.period == 1 && !cfg.check_nlist && ep->dtype == cfg.scalar_dtype && (ep->dtype == HTF_F32 || HTF_EPILOGUE_F64) && pot != nullptr && !own_evaluator(pot) &&
                    (pot->pp.kind == HTF_POT_LJ || pot->pp.kind == HTF_POT_WCA) && // (the forms compiled with an epilogue: fused_eval.hip)
                    (ep->brick == nullptr || !ep->brick->halo_wrap);                // (a replica brick on the GLOBAL cell grid wraps its messages)
    ctx->epilogue_ok[slot] = false;
    if (ok) {
        if (ctx->d_epilogue[slot] == nullptr) HTF_CHECK_HIP(hipMalloc(&ctx->d_epilogue[slot], sizeof(StepEpilogue<double>)));
        if (ep->dtype == HTF_F64) {
            const StepEpilogue<double> e = make_step_epilogue<double>(ep);
            HTF_CHECK_HIP(hipMemcpy(ctx->d_epilogue[slot], &e, sizeof e, hipMemcpyHostToDevice));
        } else {
            const StepEpilogue<float> e = make_step_epilogue<float>(ep);
            HTF_CHECK_HIP(hipMemcpy(ctx->d_epilogue[slot], &e, sizeof e, hipMemcpyHostToDevice));
        }
        ctx->epilogue_ok[slot] = true;
        ctx->epilogue_level[slot] = ep->brick != nullptr ? 2 : 1;
    }
    if (applies) *applies = ok ? 1 : 0;
    return HTF_OK;
}

extern "C" int htfs_use_step_epilogue(htf_ctx *ctx, int slot) {
    using namespace htf;
    HTF_REQUIRE(ctx, "htfs_use_step_epilogue: null context");
    HTF_REQUIRE(slot >= -1 && slot < HTFS_EPILOGUE_SLOTS, "htfs_use_step_epilogue: slot %d", slot);
    HTF_REQUIRE(slot < 0 || ctx->epilogue_ok[slot], "htfs_use_step_epilogue: slot %d holds no descriptor this context honours", slot);
    ctx->epilogue_slot = slot;
    return HTF_OK;
}

extern "C" int htf_compute_forces(htf_ctx *ctx, unsigned timestep, const htf_hoomd_arrays *a, htf_stream stream) {
    return compute_rows(ctx, timestep, a, 0, a ? a->N : 0, stream);
}

extern "C" int htf_compute_forces_rows(htf_ctx *ctx, unsigned timestep, const htf_hoomd_arrays *a, unsigned row_begin,
                                       unsigned row_count, htf_stream stream) {
    return compute_rows(ctx, timestep, a, row_begin, row_count, stream);
}

extern "C" int htf_profile_enable(htf_ctx *ctx, int on) {
    using namespace htf;
    HTF_REQUIRE(ctx, "htf_profile_enable: null context");
    ctx->profiling = on > 0 ? (unsigned)on : 0u;
    ctx->prof_tick = 0;
    ctx->ev_used = 0;
    ctx->ev_one_scope.clear();
    ctx->ev_complete.clear();
    // events for the first 64 bracketed calls exist before the caller's timed region starts (creating one costs microseconds)
    while (on > 0 && ctx->ev_pool.size() < 192) {
        hipEvent_t e;
        HTF_CHECK_HIP(hipEventCreate(&e));
        ctx->ev_pool.push_back(e);
    }
    return HTF_OK;
}

extern "C" int htf_profile_read(htf_ctx *ctx, double *build_ms, double *eval_ms, unsigned *n_calls) {
    using namespace htf;
    HTF_REQUIRE(ctx, "htf_profile_read: null context");
    double b = 0, e = 0;
    const size_t triples = ctx->ev_used / 3;
    unsigned counted = 0;
    int rc = HTF_OK;
    // a batch that returned early (nlist overflow, launch error) left its closing event unrecorded: skipped.
    // Whatever happens below, the accumulators are reset, so that one bad read cannot poison the next.
    for (size_t t = 0; t < triples && rc == HTF_OK; ++t) {
        if (t >= ctx->ev_complete.size() || !ctx->ev_complete[t]) continue;
        float ms = 0;
        hipError_t he = hipEventSynchronize(ctx->ev_pool[3 * t + 2]);
        if (he == hipSuccess && ctx->ev_one_scope[t]) {
            he = hipEventElapsedTime(&ms, ctx->ev_pool[3 * t], ctx->ev_pool[3 * t + 2]);
            e += ms;
        } else if (he == hipSuccess) {
            he = hipEventElapsedTime(&ms, ctx->ev_pool[3 * t], ctx->ev_pool[3 * t + 1]);
            b += ms;
            if (he == hipSuccess) he = hipEventElapsedTime(&ms, ctx->ev_pool[3 * t + 1], ctx->ev_pool[3 * t + 2]);
            e += ms;
        }
        if (he != hipSuccess) {
            set_error("htf_profile_read: %s", hipGetErrorString(he));
            rc = HTF_ERR_DEVICE;
        } else {
            ++counted;
        }
    }
    if (build_ms) *build_ms = b;
    if (eval_ms) *eval_ms = e;
    if (n_calls) *n_calls = counted;
    ctx->ev_used = 0;
    ctx->ev_one_scope.clear();
    ctx->ev_complete.clear();
    return rc;
}
