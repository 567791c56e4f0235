// Per-slot arithmetic of the closed-form pair potentials, shared by the streaming
// evaluator (eval_pair.hip) and the fused gather-evaluate kernel (fused_eval.hip).
#pragma once
#include "htf_common.h"
#include "htf_internal.h"

namespace htf {

// v_rcp_f32: 1 ulp, no IEEE div expansion (v_div_scale/fmas/fixup: ~10 instructions each).
// Worst amplification is s^13 in the LJ force: 13 ulp ~ 1.5e-6 relative, inside the 2e-5
// parity tolerance (measured: tests/test_gpu_parity.py ratios in gpurun_out/parity_stats.json).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// v_sqrt_f32 (1 ulp).  sqrtf() is correctly rounded under hipcc's defaults: 16 instructions (scale,
// two fma fix-ups, class checks) where the evaluators need one; PMC showed the fused kernel
// VALU-issue-bound (48 M wave-instructions = 78 us of its 95 us), so this matters.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Trainable potentials keep theta on the device; every kernel resolves it once at entry
// (wave-uniform scalar loads) so an optimizer step is visible to the next launch.
// The WCA mask `tf.norm(x) < cut` (layers.py:97) without the square root: sqrtf is monotone, so
// sqrtf(r2) < c  <=>  r2 < t with t the smallest float whose correctly rounded root reaches c.
// Found by stepping a few ulps around c*c; bit-for-bit the same mask, 16 instructions fewer per slot.
__host__ __device__ inline float sqrt_threshold(float c) {
    float t = c * c;
    for (int i = 0; i < 8; ++i) {
        const float tp = nextafterf(t, 0.0f);
        if (sqrtf(tp) >= c) t = tp; else break;
    }
    for (int i = 0; i < 8; ++i) {
        if (sqrtf(t) < c) t = nextafterf(t, __builtin_huge_valf()); else break;
    }
    return t;
}

// (KIND is a template argument and every index is static: a run-time switch / loop over
// coef[] would move the whole struct to scratch memory -- measured +2x on the LJ evaluator.)
template <int KIND>
__device__ __forceinline__ PotParams resolve_theta(PotParams p) {
    if constexpr (KIND == HTF_POT_LJ_PARAM) {
        if (p.theta != nullptr) {
            p.lj_w0 = p.theta[0];
            p.lj_w1 = p.theta[1];
        }
    } else if constexpr (KIND == HTF_POT_WCA) {
        if (p.theta != nullptr) {
            p.sigma = p.theta[0];
            p.wca_cut = p.sigma * 1.2599210498948732f;
            p.wca_cut_r2 = sqrt_threshold(p.wca_cut);
        }
    } else if constexpr (KIND == HTF_POT_RINV_POLY) {
        if (p.theta != nullptr) {
#pragma unroll
            for (int k = 0; k < HTF_MAX_POLY_TERMS; ++k)
                if (k < p.n_terms) p.coef[k] = p.theta[k];
        }
    }
    return p;
}

// shared forward of every rinv-based energy: t = x + 1e-7, r' = |t|, s = nlist_rinv
struct RinvFwd {
    float tx, ty, tz, rp, irp, s; // irp = 1 / r'
    bool cond;
};

// s = 1 / (r' + 3e-6) and 1 / r' from ONE transcendental instruction where there were three (v_sqrt_f32, v_rcp_f32 of the sum,
// v_rcp_f32 of r'): they issue at a quarter of the VALU rate and the evaluators are VALU-bound.  With s0 = rsq(r'^2) = 1 / r'
// and u = 3e-6 s0:  1 / (r' + 3e-6) = s0 / (1 + u) = s0 (1 - u + u^2) to within u^3, which for a slot with r' > 0.015
// (u < 2e-4, u^3 < 1e-11) is far below an ulp -- and ~2 ulp in all, against ~2.5 for sqrt + add + rcp.  A closer (unphysical)
// live slot takes the three-instruction form.  The CHOICE IS PER LANE (a slot's e and f depend on that slot alone, whatever
// else shares its wave and whichever kernel evaluates it -- ADVICE r2); only the cost is per wave: the three transcendental
// instructions are issued behind a wave-uniform branch that is taken when some lane needs them.
__device__ __forceinline__ RinvFwd rinv_fwd(float x, float y, float z) {
    RinvFwd f;
    f.tx = x + kNormDelta;
    f.ty = y + kNormDelta;
    f.tz = z + kNormDelta;
    const float t2 = f.tx * f.tx + f.ty * f.ty + f.tz * f.tz;
#ifdef HTF_VALU_PAD // experiment: HTF_VALU_PAD independent dummy VALU instructions per slot (does the instruction count bind?)
    {
        float pad = f.tx;
#pragma unroll
        for (int i = 0; i < HTF_VALU_PAD; ++i) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(pad));
    }
#endif
    const float s0 = __builtin_amdgcn_rsqf(t2);
    const float u = kRinvDelta * s0;
    f.rp = t2 * s0;
    f.cond = f.rp > kRinvDelta;
    f.irp = s0;
    const float sr = fmaf(u * u, s0, fmaf(-u, s0, s0));
    f.s = f.cond ? sr : 0.0f;
    const bool near = f.cond && u >= 2e-4f;
    if (__builtin_amdgcn_ballot_w64(near) != 0ull) { // wave-uniform, rare
        asm volatile("" ::: "memory"); // keeps this a BRANCH: if-converted, its three transcendentals and selects ran for every slot
        const float rp = fast_sqrt(t2);
        const float irp = fast_rcp(rp);
        const float s3 = fast_rcp(rp + kRinvDelta);
        // (rp > 3e-6 holds for every `near` lane: u >= 2e-4 with cond set means 3e-6 < r' <= 0.015)
        f.rp = near ? rp : f.rp;
        f.irp = near ? irp : f.irp;
        f.s = near ? s3 : f.s;
    }
    return f;
}

// A ROW FUNCTION (round 6, generated units only): energy_i = F(rho_i), rho_i = sum_j e_ij of the unit's pair body -- an embedding
// term, a coordination-number restraint; what the reference gets from tf.gradients of any compute() that feeds a reduce_sum into a
// nonlinearity (simmodel.py:87-121, 526-555).  The gradient with respect to row i's own pair vectors is F'(rho_i) x the pair
// body's, so the kernels accumulate a row exactly as for a pair energy and finish it here: forces x dF, energy = Fv.
template <int KIND> struct HasRowFn { static constexpr bool value = false; };
#ifdef HTF_JIT_ROW_FN
template <> struct HasRowFn<HTF_POT_JIT> { static constexpr bool value = true; };
#endif
template <int KIND>
__device__ __forceinline__ void row_function(const PotParams &p, float rho, float &Fv, float &dF) {
    Fv = rho;
    dF = 1.0f;
#ifdef HTF_JIT_ROW_FN
    if constexpr (KIND == HTF_POT_JIT) {
        HTF_JIT_ROW_FN
    }
#endif
    (void)p;
}
// ... on a finished row as the one-kernel step holds it: lanes 0 / 16 / 32 / 48 of `tot` carry fx, fz, fy, e (wave_sum4)
template <int KIND>
__device__ __forceinline__ float finish_row_sums(const PotParams &p, float tot, unsigned lane, float *abs_dF = nullptr) {
    if constexpr (HasRowFn<KIND>::value) {
        float Fv, dF;
        row_function<KIND>(p, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tot), 48)), Fv, dF);
        if (abs_dF) *abs_dF = fabsf(dF);
        return lane >= 48u ? Fv : tot * dF;
    } else {
        (void)p;
        (void)lane;
        if (abs_dF) *abs_dF = 1.0f;
        return tot;
    }
}

// Each potential returns the pair energy e (its share of E_i) and nlist_forces_ij =
// 2 * dE/dx_ij (the reference's "nlist_forces", simmodel.py:548) in (fx, fy, fz).
// (pair_eval_f takes the shared forward f = rinv_fwd(x, y, z) from its caller: a sweep that evaluates two potentials per slot --
//  config C4 -- computes it once; pair_eval below is the one-potential form)
template <int KIND>
__device__ __forceinline__ void pair_eval_f(const RinvFwd &f, float x, float y, float z, const PotParams &p,
                                            float &e, float &fx, float &fy, float &fz, float tj = 0.0f, float ti = 0.0f) {
    // (tj, ti: the neighbor's and the row particle's own type as floats -- nlist[i, j, 3] and positions[i, 3] -- read by generated
    //  bodies only; every built-in closed form ignores them and the compiler drops the conversions at its call sites)
    if constexpr (KIND == HTF_POT_SIMPLE) {
        // build_examples.py:9-22: -1 * ((1/|x|) * x), non-finite -> 0 (forward only)
        float rs = sqrtf(x * x + y * y + z * z);
        float inv = 1.0f / rs;
        float ax = -1.0f * (inv * x), ay = -1.0f * (inv * y), az = -1.0f * (inv * z);
        fx = isfinite(ax) ? ax : 0.0f;
        fy = isfinite(ay) ? ay : 0.0f;
        fz = isfinite(az) ? az : 0.0f;
        e = 0.0f;
        return;
    } else if constexpr (KIND == HTF_POT_LJ_PARAM) {
        // example 06 LJLayer on r = safe_norm(x): q = w1^6 / r^6 (divide_no_nan; TF's kernels flush
        // the padded slots' r^6 ~ 2.7e-41 to zero, i.e. q = 0 there: same as the r > 3e-6 mask),
        // e = w0 * 4 (q^2 - q) / 2,  de/dr = 2 w0 (2q - 1) (-6 q / r),  nlist_forces = 2 de/dr t / r
        const float tx = x + kNormDelta, ty = y + kNormDelta, tz = z + kNormDelta;
        const float r = fast_sqrt(tx * tx + ty * ty + tz * tz);
        const bool m = r > kRinvDelta;
        const float rr = fast_rcp(r);
        const float ri = m ? rr : 0.0f;
        const float a = p.lj_w1 * ri, a2 = a * a;
        const float q = a2 * a2 * a2;
        e = 2.0f * p.lj_w0 * (q * q - q);
        const float c = 2.0f * (2.0f * p.lj_w0 * (2.0f * q - 1.0f) * (-6.0f * q * ri)) * ri;
        fx = c * tx;
        fy = c * ty;
        fz = c * tz;
        return;
    } else if constexpr (KIND == HTF_POT_JIT) {
        // A traced elementwise energy (hoomd_tf_amd/codegen.py): HTF_JIT_BODY is generated code that reads
        //   s  = nlist_rinv of the slot (0 where masked), with d s / d r' = -s^2 where f.cond,
        //   r  = safe_norm of the slot (r' = |x + 1e-7|),
        //   x, y, z (for masks on the plain norm), tj, ti (types: parameter tables by species)
        // and assigns `e` (the slot's energy) and `dedr` (its total derivative with respect to r', forward mode).
        // nlist_forces = 2 de/dr' t / r' (simmodel.py:548), as every rinv-based closed form above.
#ifdef HTF_JIT_BODY
        const float s = f.s, r = f.rp;
        const float ds = f.cond ? -(s * s) : 0.0f;
        float dedr = 0.0f;
        e = 0.0f;
        { HTF_JIT_BODY }
        const float c = f.rp > 0.0f ? 2.0f * dedr * f.irp : 0.0f;
        fx = c * f.tx;
        fy = c * f.ty;
        fz = c * f.tz;
#else
        e = fx = fy = fz = 0.0f; // (never instantiated outside a generated unit)
#endif
        return;
    } else if constexpr (KIND == HTF_POT_GAUSS) {
        // one RBFExpansion channel as a pair energy: r = safe_norm(x) (simmodel.py:581-594),
        // phi = exp(-(r - r0)^2 / gap) (layers.py:46-49), masked with the nlist_rinv criterion.
        // nlist_forces = 2 * c * dphi/dr * t / r,  dphi/dr = -2 (r - r0) / gap * phi
        // (the sweeps that evaluate this are VALU-issue bound: exp(-d^2 / gap) as exp2 of ONE product with -log2(e) / gap, the
        //  force factor 2 coef (-2 d / gap) phi / r' as three multiplies on the precomputed -4 coef / gap; a masked slot gets
        //  phi = 0, which zeroes energy and force alike -- one select instead of two)
        const float d = f.rp - p.gauss_r0;
        const float ex = __builtin_amdgcn_exp2f((d * d) * p.gauss_k_exp);
        const float phi = f.cond ? ex : 0.0f;
        e = p.gauss_coef * phi;
        // (v_mul_legacy_f32: 0 times ANYTHING, inf and NaN included, is +0.  The one unphysical slot with r' = 0 exactly --
        //  x = y = z = -1e-7 -- has irp = inf and, on rinv_fwd's common branch, rp = 0 * inf = NaN, hence d = NaN: both
        //  products that meet its phi = 0 are legacy multiplies, so its force is an exact zero without a select)
        float phi_irp, c;
        asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(phi_irp) : "v"(phi), "v"(f.irp));
        asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(c) : "v"(p.gauss_k_force * d), "v"(phi_irp));
        fx = c * f.tx;
        fy = c * f.ty;
        fz = c * f.tz;
        return;
    } else {
        const float s = f.s, s2 = s * s;
        float dEds;
        if constexpr (KIND == HTF_POT_LJ) {
            // build_examples.py:70-74: inv_r6 = rinv**6; 4/2 * (inv_r6*inv_r6 - inv_r6).  The chain
            // 2 dE/ds (ds/dr') / r' = 2 [12 (2 s^6 - 1) s^5] (-s^2) / r' is folded to
            // -24 (2 s^6 - 1) s^7 / r': five multiplies instead of eight (the evaluators are
            // VALU-issue bound); s = 0 on masked slots makes e and c vanish by themselves.
            const float s6 = s2 * s2 * s2;
            e = 2.0f * fmaf(s6, s6, -s6);
            const float c = f.cond ? (fmaf(2.0f, s6, -1.0f) * (s6 * s)) * f.irp * -24.0f : 0.0f;
            fx = c * f.tx;
            fy = c * f.ty;
            fz = c * f.tz;
            return;
        } else if constexpr (KIND == HTF_POT_WCA) {
            // layers.py:91-98
            float q = p.sigma * s, q2 = q * q;
            float q6 = q2 * q2 * q2;
            bool in = (x * x + y * y + z * z) < p.wca_cut_r2; // == sqrtf(...) < wca_cut, see sqrt_threshold
            float e_raw = in ? q6 : 0.0f;
            e = fminf(fmaxf(e_raw, 0.0f), 10.0f);
            bool pass = in && (e_raw >= 0.0f) && (e_raw <= 10.0f); // clip_by_value gradient
            dEds = pass ? 6.0f * (q2 * q2 * q) * p.sigma : 0.0f;
        } else { // HTF_POT_RINV_POLY
            e = 0.0f;
            dEds = 0.0f;
#pragma unroll
            for (int k = 0; k < HTF_MAX_POLY_TERMS; ++k) { // static indices: p stays in registers
                if (k < p.n_terms) {
                    int pw = p.power[k] - 1; // powers validated >= 1 on the host
                    float b = s, acc = 1.0f;
                    while (pw > 0) {
                        if (pw & 1) acc *= b;
                        b *= b;
                        pw >>= 1;
                    }
                    dEds += p.coef[k] * (float)p.power[k] * acc;
                    e += p.coef[k] * (acc * s);
                }
            }
            if (p.poly_cut_r2 > 0.0f) { // wave-uniform: example 01's `cast(norm < cut) * energy`, a mask without a gradient
                asm volatile("" ::: "memory");
                const bool in = plain_sq3(x, y, z) < p.poly_cut_r2;
                e = in ? e : 0.0f;
                dEds = in ? dEds : 0.0f;
            }
        }
        // d s / d r' = -s^2 (where cond), d r' / d t = t / r'; times 2 (simmodel.py:548)
        float c = f.cond ? 2.0f * (dEds * (-s2)) * f.irp : 0.0f;
        fx = c * f.tx;
        fy = c * f.ty;
        fz = c * f.tz;
        if (!f.cond) e = 0.0f;
    }
}


template <int KIND>
__device__ __forceinline__ void pair_eval(float x, float y, float z, const PotParams &p,
                                          float &e, float &fx, float &fy, float &fz, float tj = 0.0f, float ti = 0.0f) {
    if constexpr (KIND == HTF_POT_SIMPLE || KIND == HTF_POT_LJ_PARAM) {
        pair_eval_f<KIND>(RinvFwd(), x, y, z, p, e, fx, fy, fz); // these two do not use the shared forward
    } else {
        pair_eval_f<KIND>(rinv_fwd(x, y, z), x, y, z, p, e, fx, fy, fz, tj, ti);
    }
}

// pair_eval for a slot that may have been dropped, WITHOUT a branch (the fused kernels are VALU-bound and a divergent
// branch saves nothing unless a whole wave is dropped): potentials that vanish identically far out (s^6 underflows, the WCA
// mask is false, exp(-r^2 / gap) is 0) are evaluated at x = 1e18 (1e12 for the Gaussian), where energy and force come out as exact zeros -- one
// select; the others are evaluated where they are and their four results selected.
template <int KIND>
__device__ __forceinline__ void pair_eval_if(bool keep, float x, float y, float z, const PotParams &p, float &e, float &fx,
                                             float &fy, float &fz, float tj = 0.0f, float ti = 0.0f) {
    if constexpr (KIND == HTF_POT_LJ || KIND == HTF_POT_WCA || KIND == HTF_POT_LJ_PARAM || KIND == HTF_POT_GAUSS) {
        // far enough that w1^6 / r^6 underflows for any sane w1 (1e18), near enough that (r - r0)^2 / gap stays finite (1e12)
        constexpr float kFar = KIND == HTF_POT_GAUSS ? 1e12f : 1e18f;
        pair_eval<KIND>(keep ? x : kFar, y, z, p, e, fx, fy, fz);
    } else {
        pair_eval<KIND>(x, y, z, p, e, fx, fy, fz, tj, ti);
        e = keep ? e : 0.0f;
        fx = keep ? fx : 0.0f;
        fy = keep ? fy : 0.0f;
        fz = keep ? fz : 0.0f;
    }
}

// Per-slot derivatives through the force for training: d(e, nlist_forces)/d(theta_k), k < P.
// dd[k] = (d fx, d fy, d fz, d e)/d theta_k.  Returns e, (fx, fy, fz) as pair_eval does.
template <int KIND> struct NumParams;
template <> struct NumParams<HTF_POT_LJ_PARAM> { static constexpr int value = 2; };
template <> struct NumParams<HTF_POT_WCA> { static constexpr int value = 1; };
template <> struct NumParams<HTF_POT_RINV_POLY> { static constexpr int value = HTF_MAX_POLY_TERMS; };
#ifdef HTF_JIT_NPARAMS // a generated unit of a traced energy with weights (hoomd_tf_amd/codegen.py generate_train_body)
template <> struct NumParams<HTF_POT_JIT> { static constexpr int value = HTF_JIT_NPARAMS; };
#endif

template <int KIND>
__device__ __forceinline__ void pair_eval_grad(float x, float y, float z, const PotParams &p, float &e, float &fx,
                                               float &fy, float &fz, float4 (&dd)[NumParams<KIND>::value], float tj = 0.0f) {
    constexpr int P = NumParams<KIND>::value;
#pragma unroll
    for (int k = 0; k < P; ++k) dd[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (KIND == HTF_POT_JIT) {
        // A traced energy with WEIGHTS (round 6): HTF_JIT_TRAIN_BODY is generated code that reads s, ds, r, x, y, z, tj and the
        // weights w_k = p.theta[k] and assigns e, dedr (as HTF_JIT_BODY does) AND, per weight k, dedw[k] = d e / d w_k and
        // d2edrdw[k] = d (de/dr') / d w_k (forward-mode jets over (r', w_k): codegen._JetEmitter).  nlist_forces = 2 de/dr' t / r'
        // (simmodel.py:548), so d nlist_forces / d w_k = 2 d2edrdw[k] t / r'.
#ifdef HTF_JIT_TRAIN_BODY
        const RinvFwd f = rinv_fwd(x, y, z);
        const float s = f.s, r = f.rp;
        const float ds = f.cond ? -(s * s) : 0.0f;
        const float ti = 0.0f; // (a trainable traced energy does not read the row particle's own type: simmodel declines)
        float dedr = 0.0f, dedw[P], d2edrdw[P];
#pragma unroll
        for (int k = 0; k < P; ++k) dedw[k] = d2edrdw[k] = 0.0f;
        e = 0.0f;
        { HTF_JIT_TRAIN_BODY }
        (void)ti;
        const float g = f.rp > 0.0f ? 2.0f * f.irp : 0.0f;
        const float c = g * dedr;
        fx = c * f.tx;
        fy = c * f.ty;
        fz = c * f.tz;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float ck = g * d2edrdw[k];
            dd[k] = make_float4(ck * f.tx, ck * f.ty, ck * f.tz, dedw[k]);
        }
#else
        e = fx = fy = fz = 0.0f;
#endif
    } else if constexpr (KIND == HTF_POT_LJ_PARAM) {
        const float tx = x + kNormDelta, ty = y + kNormDelta, tz = z + kNormDelta;
        const float r = fast_sqrt(tx * tx + ty * ty + tz * tz);
        const bool m = r > kRinvDelta;
        const float rr = fast_rcp(r);
        const float ri = m ? rr : 0.0f;
        const float a = p.lj_w1 * ri, a2 = a * a;
        const float q = a2 * a2 * a2;
        const float w0 = p.lj_w0, w1 = p.lj_w1;
        e = 2.0f * w0 * (q * q - q);
        const float dedr = 2.0f * w0 * (2.0f * q - 1.0f) * (-6.0f * q * ri);
        const float c = 2.0f * dedr * ri;
        fx = c * tx; fy = c * ty; fz = c * tz;
        // d/dw0: everything is linear in w0
        const float c0 = 2.0f * (2.0f * (2.0f * q - 1.0f) * (-6.0f * q * ri)) * ri;
        dd[0] = make_float4(c0 * tx, c0 * ty, c0 * tz, 2.0f * (q * q - q));
        // d/dw1: dq/dw1 = 6 q / w1
        const float dq = w1 != 0.0f ? 6.0f * q / w1 : 0.0f;
        const float c1 = 2.0f * (2.0f * w0 * (-6.0f * ri) * (4.0f * q - 1.0f) * dq) * ri;
        dd[1] = make_float4(c1 * tx, c1 * ty, c1 * tz, 2.0f * w0 * (2.0f * q - 1.0f) * dq);
    } else {
        RinvFwd f = rinv_fwd(x, y, z);
        const float s = f.s, s2 = s * s;
        const float geo = f.cond ? 2.0f * (-s2) * f.irp : 0.0f; // nlist_forces = geo * dE/ds * t
        float dEds;
        if constexpr (KIND == HTF_POT_WCA) {
            const float sig = p.sigma;
            const float q = sig * s, q2 = q * q, q5 = q2 * q2 * q, q6 = q5 * q;
            const bool in = (x * x + y * y + z * z) < p.wca_cut_r2;
            const float e_raw = in ? q6 : 0.0f;
            e = fminf(fmaxf(e_raw, 0.0f), 10.0f);
            const bool pass = in && (e_raw >= 0.0f) && (e_raw <= 10.0f);
            dEds = pass ? 6.0f * q5 * sig : 0.0f;
            // d e / d sigma = 6 sigma^5 s^6 ; d(dE/ds)/d sigma = 36 sigma^5 s^5 (mask and clip are piecewise constant)
            const float s5 = s2 * s2 * s, sg5 = sig * sig * sig * sig * sig;
            const float de = pass ? 6.0f * sg5 * s5 * s : 0.0f;
            const float dd_ds = pass ? 36.0f * sg5 * s5 : 0.0f;
            dd[0] = make_float4(geo * dd_ds * f.tx, geo * dd_ds * f.ty, geo * dd_ds * f.tz, de);
        } else { // RINV_POLY: linear in the coefficients
            e = 0.0f;
            dEds = 0.0f;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                if (k < p.n_terms) {
                    int pw = p.power[k] - 1;
                    float b = s, acc = 1.0f;
                    while (pw > 0) {
                        if (pw & 1) acc *= b;
                        b *= b;
                        pw >>= 1;
                    }
                    const float dk = (float)p.power[k] * acc; // d(dE/ds)/dc_k
                    const float ek = acc * s;                 // d e / d c_k
                    dEds += p.coef[k] * dk;
                    e += p.coef[k] * ek;
                    dd[k] = make_float4(geo * dk * f.tx, geo * dk * f.ty, geo * dk * f.tz, f.cond ? ek : 0.0f);
                }
            }
            if (p.poly_cut_r2 > 0.0f && !(plain_sq3(x, y, z) < p.poly_cut_r2)) {
                e = 0.0f;
                dEds = 0.0f;
#pragma unroll
                for (int k = 0; k < P; ++k) dd[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float c = geo * dEds;
        fx = c * f.tx; fy = c * f.ty; fz = c * f.tz;
        if (!f.cond) e = 0.0f;
    }
}

// simmodel.py:509-523 per-slot virial term: -(|nf| / (2 |x|)) x (x) x with divide_no_nan
struct Virial6 {
    float xx = 0.f, xy = 0.f, xz = 0.f, yy = 0.f, yz = 0.f, zz = 0.f;
    __device__ __forceinline__ void add(float x, float y, float z, float ax, float ay, float az) {
        float fmag = sqrtf(ax * ax + ay * ay + az * az);
        float den = 2.0f * sqrtf(x * x + y * y + z * z);
        float frs = (den == 0.0f) ? 0.0f : fmag / den;
        xx -= frs * x * x;
        xy -= frs * x * y;
        xz -= frs * x * z;
        yy -= frs * y * y;
        yz -= frs * y * z;
        zz -= frs * z * z;
    }
};

} // namespace htf
