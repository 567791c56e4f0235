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

// shared forward of every rinv-based energy: t = x + 1e-7, r' = |t|, s = nlist_rinv
struct RinvFwd {
    float tx, ty, tz, rp, s;
    bool cond;
};

__device__ __forceinline__ RinvFwd rinv_fwd(float x, float y, float z) {
    RinvFwd f;
    f.tx = x + kNormDelta;
    f.ty = y + kNormDelta;
    f.tz = z + kNormDelta;
    f.rp = sqrtf(f.tx * f.tx + f.ty * f.ty + f.tz * f.tz);
    f.cond = f.rp > kRinvDelta;
    f.s = f.cond ? fast_rcp(f.rp + kRinvDelta) : 0.0f;
    return f;
}

// Each potential returns the pair energy e (its share of E_i) and nlist_forces_ij =
// 2 * dE/dx_ij (the reference's "nlist_forces", simmodel.py:548) in (fx, fy, fz).
template <int KIND>
__device__ __forceinline__ void pair_eval(float x, float y, float z, const PotParams &p,
                                          float &e, float &fx, float &fy, float &fz) {
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
    } else if constexpr (KIND == HTF_POT_GAUSS) {
        // one RBFExpansion channel as a pair energy: r = safe_norm(x) (simmodel.py:581-594),
        // phi = exp(-(r - r0)^2 / gap) (layers.py:46-49), masked with the nlist_rinv criterion.
        // nlist_forces = 2 * c * dphi/dr * t / r,  dphi/dr = -2 (r - r0) / gap * phi
        const float tx = x + kNormDelta, ty = y + kNormDelta, tz = z + kNormDelta;
        const float r = sqrtf(tx * tx + ty * ty + tz * tz);
        const bool m = r > kRinvDelta;
        const float d = r - p.gauss_r0;
        const float phi = m ? __expf(-(d * d) * p.gauss_ginv) : 0.0f;
        e = p.gauss_coef * phi;
        const float c = m ? 2.0f * p.gauss_coef * (-2.0f * d * p.gauss_ginv) * phi * fast_rcp(r) : 0.0f;
        fx = c * tx;
        fy = c * ty;
        fz = c * tz;
        return;
    } else {
        RinvFwd f = rinv_fwd(x, y, z);
        const float s = f.s, s2 = s * s;
        float dEds;
        if constexpr (KIND == HTF_POT_LJ) {
            // build_examples.py:70-74: inv_r6 = rinv**6; 4/2 * (inv_r6*inv_r6 - inv_r6)
            float s6 = s2 * s2 * s2;
            e = 2.0f * (s6 * s6 - s6);
            dEds = 2.0f * (2.0f * s6 - 1.0f) * (6.0f * (s2 * s2 * s));
        } else if constexpr (KIND == HTF_POT_WCA) {
            // layers.py:91-98
            float q = p.sigma * s, q2 = q * q;
            float q6 = q2 * q2 * q2;
            float r = sqrtf(x * x + y * y + z * z);
            bool in = r < p.wca_cut;
            float e_raw = in ? q6 : 0.0f;
            e = fminf(fmaxf(e_raw, 0.0f), 10.0f);
            bool pass = in && (e_raw >= 0.0f) && (e_raw <= 10.0f); // clip_by_value gradient
            dEds = pass ? 6.0f * (q2 * q2 * q) * p.sigma : 0.0f;
        } else { // HTF_POT_RINV_POLY
            e = 0.0f;
            dEds = 0.0f;
            for (int k = 0; k < p.n_terms; ++k) {
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
        // d s / d r' = -s^2 (where cond), d r' / d t = t / r'; times 2 (simmodel.py:548)
        float c = f.cond ? 2.0f * (dEds * (-s2)) * fast_rcp(f.rp) : 0.0f;
        fx = c * f.tx;
        fy = c * f.ty;
        fz = c * f.tz;
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
