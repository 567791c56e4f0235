/* C restatement of the hoomd-tf force path for CPU-baseline timing (kind "port").
 *
 * TEST INFRASTRUCTURE ONLY: built into oracle/_build/libhtf_oracle.so, loaded only by
 * tests/ and by bench.py's cpu_baseline leg.  The product never links it.
 *
 * Two flavours of the same arithmetic, both validated against oracle/htf_oracle.py
 * (tests/test_oracle_c.py):
 *   htfo_compute_forces_lj     fused closed form, OpenMP over particles (the fastest
 *                              honest CPU version of the path);
 *   htfo_prepare_neighbors     TensorflowCompute.cc:303-374 prepareNeighbors;
 *   htfo_lj_from_nlist         LJModel on a dense [N,NN,4] fp32 tensor
 *                              (build_examples.py:67-77 + simmodel.py:526-578,618-635).
 * The reference's real CPU path is a TF2 graph (one pass over [N,NN] per op); its
 * cost structure is timed separately by oracle/graph_torch.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RINV_DELTA 3e-6f
#define NORM_DELTA 1e-7f

void htfo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int htfo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* HOOMD BoxDim::minImage (rint form), orthorhombic + tilt, double precision */
static inline void min_image_d(double *x, double *y, double *z, const double *L, const double *Linv,
                               const double *tilt, const int *periodic) {
    if (periodic[2]) {
        double img = rint(*z * Linv[2]);
        *z -= L[2] * img;
        *y -= L[2] * tilt[2] * img;
        *x -= L[2] * tilt[1] * img;
    }
    if (periodic[1]) {
        double img = rint(*y * Linv[1]);
        *y -= L[1] * img;
        *x -= L[1] * tilt[0] * img;
    }
    if (periodic[0]) {
        double img = rint(*x * Linv[0]);
        *x -= L[0] * img;
    }
}

static inline void min_image_f(float *x, float *y, float *z, const float *L, const float *Linv,
                               const float *tilt, const int *periodic) {
    if (periodic[2]) {
        float img = rintf(*z * Linv[2]);
        *z -= L[2] * img;
        *y -= L[2] * tilt[2] * img;
        *x -= L[2] * tilt[1] * img;
    }
    if (periodic[1]) {
        float img = rintf(*y * Linv[1]);
        *y -= L[1] * img;
        *x -= L[1] * tilt[0] * img;
    }
    if (periodic[0]) {
        float img = rintf(*x * Linv[0]);
        *x -= L[0] * img;
    }
}

/* TensorflowCompute.cc:303-374, Scalar = float.  pos4: [Ntot,4] (w = int type bits).
 * dest: [batch, NN, 4] float. */
void htfo_prepare_neighbors_f32(float *dest, const float *pos4, const uint32_t *n_neigh,
                                const uint32_t *head_list, const uint32_t *nlist, const double *lo,
                                const double *hi, const double *tilt_d, const int *periodic, double r_cut,
                                unsigned NN, unsigned offset, unsigned batch) {
    float L[3], Linv[3], tilt[3];
    for (int d = 0; d < 3; ++d) {
        L[d] = (float)hi[d] - (float)lo[d];
        Linv[d] = 1.0f / L[d];
        tilt[d] = (float)tilt_d[d];
    }
    const float rc = (float)r_cut, rc2 = rc * rc;
    memset(dest, 0, (size_t)batch * NN * 4 * sizeof(float));
#pragma omp parallel for schedule(static)
    for (long bi = 0; bi < (long)batch; ++bi) {
        const unsigned i = offset + (unsigned)bi;
        const float *pi = pos4 + 4 * (size_t)i;
        const uint32_t head = head_list[i];
        unsigned nno = 0;
        float *row = dest + (size_t)bi * NN * 4;
        for (uint32_t j = 0; j < n_neigh[i]; ++j) {
            const uint32_t k = nlist[head + j];
            const float *pk = pos4 + 4 * (size_t)k;
            float dx = pk[0] - pi[0], dy = pk[1] - pi[1], dz = pk[2] - pi[2];
            min_image_f(&dx, &dy, &dz, L, Linv, tilt, periodic);
            if (dx * dx + dy * dy + dz * dz > rc2) continue;
            int32_t ty;
            memcpy(&ty, pk + 3, 4);
            row[4 * nno + 0] = dx;
            row[4 * nno + 1] = dy;
            row[4 * nno + 2] = dz;
            row[4 * nno + 3] = (float)ty;
            nno = (nno + 1) % NN;
        }
    }
}

void htfo_prepare_neighbors_f64(double *dest, const double *pos4, const uint32_t *n_neigh,
                                const uint32_t *head_list, const uint32_t *nlist, const double *lo,
                                const double *hi, const double *tilt, const int *periodic, double r_cut,
                                unsigned NN, unsigned offset, unsigned batch) {
    double L[3], Linv[3];
    for (int d = 0; d < 3; ++d) {
        L[d] = hi[d] - lo[d];
        Linv[d] = 1.0 / L[d];
    }
    const double rc2 = r_cut * r_cut;
    memset(dest, 0, (size_t)batch * NN * 4 * sizeof(double));
#pragma omp parallel for schedule(static)
    for (long bi = 0; bi < (long)batch; ++bi) {
        const unsigned i = offset + (unsigned)bi;
        const double *pi = pos4 + 4 * (size_t)i;
        const uint32_t head = head_list[i];
        unsigned nno = 0;
        double *row = dest + (size_t)bi * NN * 4;
        for (uint32_t j = 0; j < n_neigh[i]; ++j) {
            const uint32_t k = nlist[head + j];
            const double *pk = pos4 + 4 * (size_t)k;
            double dx = pk[0] - pi[0], dy = pk[1] - pi[1], dz = pk[2] - pi[2];
            min_image_d(&dx, &dy, &dz, L, Linv, tilt, periodic);
            if (dx * dx + dy * dy + dz * dz > rc2) continue;
            int32_t ty;
            memcpy(&ty, pk + 3, 4); /* low 32 bits (little endian): HOOMD __double_as_int */
            row[4 * nno + 0] = dx;
            row[4 * nno + 1] = dy;
            row[4 * nno + 2] = dz;
            row[4 * nno + 3] = (double)ty;
            nno = (nno + 1) % NN;
        }
    }
}

/* LJModel (build_examples.py:67-77) on a dense fp32 nlist: force[N,4] = (F, E_i). */
void htfo_lj_from_nlist(const float *nl, unsigned N, unsigned NN, float *force) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; ++i) {
        const float *row = nl + (size_t)i * NN * 4;
        float fx = 0, fy = 0, fz = 0, en = 0;
        for (unsigned j = 0; j < NN; ++j) {
            const float tx = row[4 * j] + NORM_DELTA, ty = row[4 * j + 1] + NORM_DELTA, tz = row[4 * j + 2] + NORM_DELTA;
            const float rp = sqrtf(tx * tx + ty * ty + tz * tz);
            if (!(rp > RINV_DELTA)) continue;
            const float s = 1.0f / (rp + RINV_DELTA);
            const float s2 = s * s, s6 = s2 * s2 * s2;
            en += 2.0f * (s6 * s6 - s6);
            const float dEds = 2.0f * (2.0f * s6 - 1.0f) * (6.0f * (s2 * s2 * s));
            const float c = 2.0f * (dEds * (-s2)) / rp;
            fx += c * tx;
            fy += c * ty;
            fz += c * tz;
        }
        force[4 * i] = fx;
        force[4 * i + 1] = fy;
        force[4 * i + 2] = fz;
        force[4 * i + 3] = en;
    }
}

/* One computeForces pass (TensorflowCompute.cc:129-216) for the LJ model, fp32
 * HOOMD build: prepareNeighbors into scratch [N,NN,4] then the model. */
void htfo_compute_forces_lj_f32(const float *pos4, unsigned N, const uint32_t *n_neigh, const uint32_t *head_list,
                                const uint32_t *nlist, const double *lo, const double *hi, const double *tilt,
                                const int *periodic, double r_cut, unsigned NN, float *scratch, float *force) {
    htfo_prepare_neighbors_f32(scratch, pos4, n_neigh, head_list, nlist, lo, hi, tilt, periodic, r_cut, NN, 0, N);
    htfo_lj_from_nlist(scratch, N, NN, force);
}

/* WCA model (build_examples.py:221-228 over layers.py:52-98 WCARepulsion) on a dense fp32 nlist:
 * rinv = nlist_rinv; e = [|x| < sigma 2^(1/3)] (sigma rinv)^6; energy column = sum_j clip(e, 0, 10)
 * (no 1/2); clip_by_value passes the gradient only where 0 <= e <= 10. */
void htfo_wca_from_nlist(const float *nl, unsigned N, unsigned NN, float sigma, float *force) {
    const float cut = sigma * (float)1.2599210498948732;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; ++i) {
        const float *row = nl + (size_t)i * NN * 4;
        float fx = 0, fy = 0, fz = 0, en = 0;
        for (unsigned j = 0; j < NN; ++j) {
            const float x = row[4 * j], y = row[4 * j + 1], z = row[4 * j + 2];
            const float tx = x + NORM_DELTA, ty = y + NORM_DELTA, tz = z + NORM_DELTA;
            const float rp = sqrtf(tx * tx + ty * ty + tz * tz);
            if (!(rp > RINV_DELTA)) continue;
            if (!(sqrtf(x * x + y * y + z * z) < cut)) continue;
            const float s = 1.0f / (rp + RINV_DELTA);
            const float q = sigma * s, q2 = q * q, q6 = q2 * q2 * q2;
            en += fminf(fmaxf(q6, 0.0f), 10.0f);
            if (!(q6 >= 0.0f && q6 <= 10.0f)) continue;
            const float dEds = 6.0f * (q2 * q2 * q) * sigma;
            const float c = 2.0f * (dEds * (-(s * s))) / rp;
            fx += c * tx;
            fy += c * ty;
            fz += c * tz;
        }
        force[4 * i] = fx;
        force[4 * i + 1] = fy;
        force[4 * i + 2] = fz;
        force[4 * i + 3] = en;
    }
}

/* Pair-MLP composite (oracle/htf_oracle.py:pair_mlp_model): r = safe_norm(x);
 * phi_k = exp(-(r - c_k)^2 / gap), c = linspace(low, high, K) in fp32; Dense(H1) - act - Dense(H2) - act -
 * Dense(1); u masked with r > 3e-6; E_i = 1/2 sum_j u; nlist_forces = 2 dE/dx = du/dr t / r.
 * Row-major Keras kernels W1 [K][H1], W2 [H1][H2], W3 [H2].  K, H1, H2 <= 64.  tanh_act: 1 tanh, 0 linear. */
void htfo_mlp_from_nlist(const float *nl, unsigned N, unsigned NN, int K, int H1, int H2, float low, float high,
                         const float *W1, const float *b1, const float *W2, const float *b2, const float *W3,
                         const float *b3, int tanh_act, float *force) {
    float c[64];
    for (int k = 0; k < K; ++k) c[k] = K > 1 ? low + (high - low) * (float)k / (float)(K - 1) : low;
    c[K - 1] = high;
    const float gap = c[1] - c[0];
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; ++i) {
        const float *row = nl + (size_t)i * NN * 4;
        float fx = 0, fy = 0, fz = 0, en = 0;
        float phi[64], d[64], h1[64], h2[64], g2[64], g1[64];
        for (unsigned j = 0; j < NN; ++j) {
            const float tx = row[4 * j] + NORM_DELTA, ty = row[4 * j + 1] + NORM_DELTA, tz = row[4 * j + 2] + NORM_DELTA;
            const float r = sqrtf(tx * tx + ty * ty + tz * tz);
            if (!(r > RINV_DELTA)) continue;
            for (int k = 0; k < K; ++k) {
                d[k] = r - c[k];
                phi[k] = expf(-(d[k] * d[k]) / gap);
            }
            for (int a = 0; a < H1; ++a) {
                float z = b1[a];
                for (int k = 0; k < K; ++k) z += phi[k] * W1[k * H1 + a];
                h1[a] = tanh_act ? tanhf(z) : z;
            }
            float u = b3[0];
            for (int b = 0; b < H2; ++b) {
                float z = b2[b];
                for (int a = 0; a < H1; ++a) z += h1[a] * W2[a * H2 + b];
                h2[b] = tanh_act ? tanhf(z) : z;
                u += h2[b] * W3[b];
                g2[b] = tanh_act ? W3[b] * (1.0f - h2[b] * h2[b]) : W3[b];
            }
            for (int a = 0; a < H1; ++a) {
                float s = 0;
                for (int b = 0; b < H2; ++b) s += g2[b] * W2[a * H2 + b];
                g1[a] = tanh_act ? s * (1.0f - h1[a] * h1[a]) : s;
            }
            float dudr = 0;
            for (int k = 0; k < K; ++k) {
                float s = 0;
                for (int a = 0; a < H1; ++a) s += g1[a] * W1[k * H1 + a];
                dudr += s * (-2.0f * d[k] / gap) * phi[k];
            }
            en += 0.5f * u;
            const float cc = dudr / r; /* 2 * (1/2) du/dr / r */
            fx += cc * tx;
            fy += cc * ty;
            fz += cc * tz;
        }
        force[4 * i] = fx;
        force[4 * i + 1] = fy;
        force[4 * i + 2] = fz;
        force[4 * i + 3] = en;
    }
}

/* One computeForces pass for the WCA / pair-MLP models (prepareNeighbors, then the model). */
void htfo_compute_forces_wca_f32(const float *pos4, unsigned N, const uint32_t *n_neigh, const uint32_t *head_list,
                                 const uint32_t *nlist, const double *lo, const double *hi, const double *tilt,
                                 const int *periodic, double r_cut, unsigned NN, float sigma, float *scratch,
                                 float *force) {
    htfo_prepare_neighbors_f32(scratch, pos4, n_neigh, head_list, nlist, lo, hi, tilt, periodic, r_cut, NN, 0, N);
    htfo_wca_from_nlist(scratch, N, NN, sigma, force);
}

/* Config C4's model on a dense fp32 nlist (oracle/htf_oracle.py:eds_rdf_model + compute_rdf): LJModel + alpha *
 * soft-RDF CV with cv = (1/N) sum_i sum_j exp(-(r - r0)^2 / gap) [r > 3e-6], r = safe_norm(x); energy column
 * E_lj,i + alpha * cv; forces = compute_nlist_forces; plus the compute_rdf histogram over nb_total bins of
 * tf.histogram_fixed_width semantics (hist: nb_total uint64 counters, zeroed here).  Two passes, as the alpha * cv
 * term of every energy needs the global cv.  Returns cv. */
double htfo_eds_from_nlist(const float *nl, unsigned N, unsigned NN, float alpha, float r0, float gap, float rdf_r0,
                           float rdf_r1, unsigned nb_total, unsigned long long *hist, float *force) {
    double cv_sum = 0.0;
    for (unsigned b = 0; b < nb_total; ++b) hist[b] = 0;
#pragma omp parallel
    {
        unsigned long long *h = (unsigned long long *)calloc(nb_total, sizeof(unsigned long long));
        double cv_local = 0.0;
#pragma omp for schedule(static)
        for (long i = 0; i < (long)N; ++i) {
            const float *row = nl + (size_t)i * NN * 4;
            float fx = 0, fy = 0, fz = 0, en = 0, phis = 0;
            for (unsigned j = 0; j < NN; ++j) {
                const float x = row[4 * j], y = row[4 * j + 1], z = row[4 * j + 2];
                /* compute_rdf: plain norm, every slot (padding included) lands in a bin */
                const float rr = sqrtf((x * x + y * y) + z * z);
                float fi = floorf((float)nb_total * ((rr - rdf_r0) / (rdf_r1 - rdf_r0)));
                const long bin = fi < 0.f ? 0 : (fi > (float)(nb_total - 1) ? (long)(nb_total - 1) : (long)fi);
                h[bin] += 1;
                const float tx = x + NORM_DELTA, ty = y + NORM_DELTA, tz = z + NORM_DELTA;
                const float rp = sqrtf(tx * tx + ty * ty + tz * tz);
                if (!(rp > RINV_DELTA)) continue;
                const float s = 1.0f / (rp + RINV_DELTA);
                const float s2 = s * s, s6 = s2 * s2 * s2;
                en += 2.0f * (s6 * s6 - s6);
                const float dEds = 2.0f * (2.0f * s6 - 1.0f) * (6.0f * (s2 * s2 * s));
                float c = 2.0f * (dEds * (-s2)) / rp;
                const float d = rp - r0;
                const float phi = expf(-(d * d) / gap);
                phis += phi;
                c += 2.0f * alpha * (-2.0f * d / gap * phi) / rp;
                fx += c * tx;
                fy += c * ty;
                fz += c * tz;
            }
            force[4 * i] = fx;
            force[4 * i + 1] = fy;
            force[4 * i + 2] = fz;
            force[4 * i + 3] = en;
            cv_local += (double)phis;
        }
#pragma omp critical
        {
            cv_sum += cv_local;
            for (unsigned b = 0; b < nb_total; ++b) hist[b] += h[b];
        }
        free(h);
    }
    const float cv = (float)(cv_sum / (double)N);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; ++i) force[4 * i + 3] += alpha * cv;
    return (double)cv;
}
