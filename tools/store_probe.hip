// Micro-benchmark behind DESIGN 3.1: what bounds a kernel that writes one 2 KB row per wave
// (131072 rows = 268 MB) after a short dependent-load prologue?
//   build: hipcc -O3 --offload-arch=gfx950 tools/store_probe.hip -o tools/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// MODE 0: stores only.  MODE 1: one dependent load (n[idx]) before the stores.
// MODE 2: chain of three dependent loads (n -> head -> idx list element) before the stores.
// ROWS: rows per wave (sequential).  PERSIST: grid-stride over rows with a fixed grid.
__device__ __forceinline__ void st_nt(float4 *p, const float4 &v) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z);
    __builtin_nontemporal_store(v.w, &p->w);
}

// NT: streaming (nontemporal) stores instead of ordinary write-back stores
template <int MODE, int ROWS, bool PERSIST, bool NT = false>
__global__ __launch_bounds__(256) void k(float4 *__restrict__ dest, const unsigned *__restrict__ a,
                                         const unsigned *__restrict__ b, const unsigned *__restrict__ c,
                                         unsigned nrows) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned nw = (gridDim.x * blockDim.x) >> 6;
    for (unsigned base = wave0 * ROWS; base < nrows; base += (PERSIST ? nw * ROWS : nrows)) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const unsigned row = base + r;
            if (row >= nrows) break;
            float v = 1.0f;
            if (MODE >= 1) {
                unsigned x = a[row];
                if (MODE >= 2) {
                    unsigned y = b[x];
                    unsigned z = c[y + lane];
                    v = (float)z;
                } else {
                    v = (float)x;
                }
            }
            float4 o = make_float4(v, v, v, v);
            float4 *p = dest + (size_t)row * 128;
            if (NT) {
                st_nt(p + lane, o);
                st_nt(p + 64 + lane, o);
            } else {
                p[lane] = o;
                p[64 + lane] = o;
            }
        }
    }
}

// chains of ROWS rows issued together (loads of all rows in flight), then all the stores
template <int ROWS>
__global__ __launch_bounds__(256) void k_front(float4 *__restrict__ dest, const unsigned *__restrict__ a,
                                               const unsigned *__restrict__ b, const unsigned *__restrict__ c,
                                               unsigned nrows) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wave0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned base = wave0 * ROWS;
    unsigned x[ROWS], y[ROWS], z[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) x[r] = a[min(base + r, nrows - 1)];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) y[r] = b[x[r]];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) z[r] = c[y[r] + lane];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const unsigned row = base + r;
        if (row >= nrows) break;
        const float v = (float)z[r];
        float4 o = make_float4(v, v, v, v);
        float4 *p = dest + (size_t)row * 128;
        p[lane] = o;
        p[64 + lane] = o;
    }
}

template <int ROWS>
static void run_front(const char *name, float4 *d, unsigned *a, unsigned *b, unsigned *c, unsigned nrows) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned waves = (nrows + ROWS - 1) / ROWS;
    unsigned grid = (waves + 3) / 4;
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_front<ROWS>), dim3(grid), dim3(256), 0, 0, d, a, b, c, nrows);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-52s %7.1f us  %.2f TB/s\n", name, best * 1e3, nrows * 2048.0 / best / 1e9);
}

template <int MODE, int ROWS, bool PERSIST, bool NT = false>
static void run(const char *name, float4 *d, unsigned *a, unsigned *b, unsigned *c, unsigned nrows) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned waves = (nrows + ROWS - 1) / ROWS;
    unsigned grid = PERSIST ? 256 * 8 : (waves + 3) / 4;
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, ROWS, PERSIST, NT>), dim3(grid), dim3(256), 0, 0, d, a, b, c, nrows);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-52s %7.1f us  %.2f TB/s\n", name, best * 1e3, nrows * 2048.0 / best / 1e9);
}

int main() {
    const unsigned nrows = 131072;
    float4 *d;
    unsigned *a, *b, *c;
    hipMalloc(&d, (size_t)nrows * 2048);
    hipMalloc(&a, nrows * 4);
    hipMalloc(&b, nrows * 4);
    hipMalloc(&c, (size_t)nrows * 144 * 4);
    std::vector<unsigned> h(nrows);
    for (unsigned i = 0; i < nrows; ++i) h[i] = i;
    hipMemcpy(a, h.data(), nrows * 4, hipMemcpyHostToDevice);
    for (unsigned i = 0; i < nrows; ++i) h[i] = i * 144;
    hipMemcpy(b, h.data(), nrows * 4, hipMemcpyHostToDevice);
    hipMemset(c, 0, (size_t)nrows * 144 * 4);
    run<0, 1, false>("stores only, 1 row/wave, 32768 blocks", d, a, b, c, nrows);
    run<0, 4, false>("stores only, 4 rows/wave", d, a, b, c, nrows);
    run<0, 1, true>("stores only, persistent 2048 blocks", d, a, b, c, nrows);
    run<1, 1, false>("1 dependent load, 1 row/wave", d, a, b, c, nrows);
    run<2, 1, false>("3 dependent loads, 1 row/wave", d, a, b, c, nrows);
    run<2, 4, false>("3 dependent loads, 4 rows/wave (sequential)", d, a, b, c, nrows);
    run<2, 1, true>("3 dependent loads, persistent 2048 blocks", d, a, b, c, nrows);
    run<2, 4, true>("3 dependent loads, persistent, 4 rows/trip", d, a, b, c, nrows);
    run<0, 1, false, true>("NT stores only, 1 row/wave", d, a, b, c, nrows);
    run<1, 1, false, true>("NT, 1 dependent load, 1 row/wave", d, a, b, c, nrows);
    run<2, 1, false, true>("NT, 3 dependent loads, 1 row/wave", d, a, b, c, nrows);
    run<2, 4, false, true>("NT, 3 dependent loads, 4 rows/wave (sequential)", d, a, b, c, nrows);
    run<2, 1, true, true>("NT, 3 dependent loads, persistent 2048 blocks", d, a, b, c, nrows);
    run_front<2>("3 dependent loads, 2 rows/wave, loads up front", d, a, b, c, nrows);
    run_front<4>("3 dependent loads, 4 rows/wave, loads up front", d, a, b, c, nrows);
    run_front<8>("3 dependent loads, 8 rows/wave, loads up front", d, a, b, c, nrows);
    return 0;
}
