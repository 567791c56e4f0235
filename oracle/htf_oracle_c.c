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
