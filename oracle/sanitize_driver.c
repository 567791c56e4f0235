/* Sanitizer harness for the C restatement ("-fsanitize=address,undefined CPU oracle
 * build").  TEST INFRASTRUCTURE ONLY.  Builds a small jittered cubic system with an O(N^2) full
 * neighbor list (some rows overflow NN on purpose: the slot wrap must stay inside the row) and
 * runs prepareNeighbors (f32 + f64) and the LJ model under ASan/UBSan.
 *   gcc -O1 -g -fsanitize=address,undefined -fopenmp oracle/sanitize_driver.c oracle/htf_oracle_c.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void htfo_set_threads(int n);
void htfo_prepare_neighbors_f32(float *dest, const float *pos4, const uint32_t *n_neigh, const uint32_t *head_list,
                                const uint32_t *nlist, const double *lo, const double *hi, const double *tilt_d,
                                const int *periodic, double r_cut, unsigned NN, unsigned offset, unsigned batch);
void htfo_prepare_neighbors_f64(double *dest, const double *pos4, const uint32_t *n_neigh, const uint32_t *head_list,
                                const uint32_t *nlist, const double *lo, const double *hi, const double *tilt_d,
                                const int *periodic, double r_cut, unsigned NN, unsigned offset, unsigned batch);
void htfo_compute_forces_lj_f32(const float *pos4, unsigned N, const uint32_t *n_neigh, const uint32_t *head_list,
                                const uint32_t *nlist, const double *lo, const double *hi, const double *tilt,
                                const int *periodic, double r_cut, unsigned NN, float *scratch, float *force);

int main(void) {
    const int n = 6;
    const unsigned N = n * n * n, NN = 12; /* 27 neighbors within r_list: every row overflows NN */
    const double a = 1.2, L = n * a, r_list = 1.45 * a;
    double lo[3] = {-L / 2, -L / 2, -L / 2}, hi[3] = {L / 2, L / 2, L / 2}, tilt[3] = {0, 0, 0};
    int periodic[3] = {1, 1, 1};
    float *pos = malloc(sizeof(float) * 4 * N);
    double *posd = malloc(sizeof(double) * 4 * N);
    unsigned seed = 12345u;
    for (unsigned i = 0; i < N; ++i) {
        int ix = i % n, iy = (i / n) % n, iz = i / (n * n);
        int c[3] = {ix, iy, iz};
        for (int d = 0; d < 3; ++d) {
            seed = seed * 1664525u + 1013904223u;
            double jit = ((seed >> 8) / 16777216.0 - 0.5) * 0.1;
            posd[4 * i + d] = (c[d] + 0.5) * a - L / 2 + jit;
            pos[4 * i + d] = (float)posd[4 * i + d];
        }
        pos[4 * i + 3] = 0.f;
        posd[4 * i + 3] = 0.0;
    }
    uint32_t *nn = calloc(N, sizeof(uint32_t)), *head = malloc(sizeof(uint32_t) * N);
    uint32_t *nl = malloc(sizeof(uint32_t) * N * 64);
    for (unsigned i = 0; i < N; ++i) {
        head[i] = i * 64;
        for (unsigned j = 0; j < N; ++j) {
            if (i == j) continue;
            double r2 = 0;
            for (int d = 0; d < 3; ++d) {
                double dx = posd[4 * j + d] - posd[4 * i + d];
                dx -= L * rint(dx / L);
                r2 += dx * dx;
            }
            if (r2 <= r_list * r_list && nn[i] < 64) nl[head[i] + nn[i]++] = j;
        }
    }
    float *scratch = malloc(sizeof(float) * 4 * N * NN), *force = malloc(sizeof(float) * 4 * N);
    double *scratchd = malloc(sizeof(double) * 4 * N * NN);
    for (int threads = 1; threads <= 4; threads += 3) {
        htfo_set_threads(threads);
        htfo_prepare_neighbors_f32(scratch, pos, nn, head, nl, lo, hi, tilt, periodic, 1.4 * a, NN, 0, N);
        htfo_prepare_neighbors_f32(scratch, pos, nn, head, nl, lo, hi, tilt, periodic, 1.4 * a, NN, 7, 50); /* a batch */
        htfo_prepare_neighbors_f64(scratchd, posd, nn, head, nl, lo, hi, tilt, periodic, 1.4 * a, NN, 0, N);
        htfo_compute_forces_lj_f32(pos, N, nn, head, nl, lo, hi, tilt, periodic, 1.4 * a, NN, scratch, force);
    }
    double s = 0;
    for (unsigned i = 0; i < 4 * N; ++i) s += force[i];
    if (!isfinite(s)) {
        fprintf(stderr, "non-finite forces\n");
        return 2;
    }
    printf("sanitize_driver ok (%u particles, sum %.6g)\n", N, s);
    free(pos); free(posd); free(nn); free(head); free(nl); free(scratch); free(scratchd); free(force);
    return 0;
}
