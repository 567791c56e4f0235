// Micro-benchmark behind DESIGN 3.1: how fast can a CU issue per-lane 16-B gathers from an
// L2-resident 2 MB table (the positions table of the pair-vector build)?
//   build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o tools/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int W>
struct Vec;
template <> struct Vec<4> { using type = float4; };
template <> struct Vec<1> { using type = float; };

template <int W>
__global__ __launch_bounds__(256) void gather_kernel(const unsigned *__restrict__ idx, const typename Vec<W>::type *__restrict__ table,
                                                     float *__restrict__ out, unsigned n_per_thread, unsigned total_threads) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    // 4 independent gathers in flight per lane per iteration
    for (unsigned i = 0; i < n_per_thread; i += 4) {
        unsigned k0 = idx[(size_t)(i + 0) * total_threads + t];
        unsigned k1 = idx[(size_t)(i + 1) * total_threads + t];
        unsigned k2 = idx[(size_t)(i + 2) * total_threads + t];
        unsigned k3 = idx[(size_t)(i + 3) * total_threads + t];
        if constexpr (W == 4) {
            float4 a = table[k0], b = table[k1], c = table[k2], d = table[k3];
            acc += a.x + b.y + c.z + d.w;
        } else {
            acc += table[k0] + table[k1] + table[k2] + table[k3];
        }
    }
    if (acc == 1234.5f) out[t] = acc;
}

int main() {
    const unsigned table_elems = 131072;  // x 16 B = 2 MB
    const unsigned blocks = 256 * 8, threads = blocks * 256, per = 64;
    std::vector<unsigned> h((size_t)threads * per);
    unsigned *d_idx;
    float4 *d_tab;
    float *d_out;
    hipMalloc(&d_idx, h.size() * 4);
    hipMalloc(&d_tab, table_elems * 16);
    hipMalloc(&d_out, threads * 4);
    hipMemset(d_tab, 0, table_elems * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[] = {"random", "runs of 4 (one 64-B line)", "runs of 16", "sequential"};
    for (int pat = 0; pat < 4; ++pat) {
        srand(1);
        for (size_t i = 0; i < h.size(); ++i) {
            const unsigned lane_global = (unsigned)(i % threads);
            const unsigned run = pat == 0 ? 1 : pat == 1 ? 4 : pat == 2 ? 16 : 64;
            // lanes of a run share a random base
            static unsigned base = 0;
            if (lane_global % run == 0) base = (unsigned)(((unsigned long long)rand() * 2654435761u) % (table_elems - 64));
            h[i] = pat == 3 ? (unsigned)((i * 1u) % table_elems) : base + lane_global % run;
        }
        hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int w = 0; w < 2; ++w) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (w == 0)
                    hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(256), 0, 0, d_idx, d_tab, d_out, per, threads);
                else
                    hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_idx, (const float *)d_tab, d_out, per, threads);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double lanes = (double)threads * per;
            printf("%-28s %2d-B gathers: %7.1f us  %6.2f G lane-loads/s  = %.2f lanes/clk/CU @2.4GHz  (index stream %.0f GB/s)\n", names[pat],
                   w == 0 ? 16 : 4, ms * 1e3, lanes / ms / 1e6, lanes / (ms * 1e-3) / 256 / 2.4e9, lanes * 4 / ms / 1e6);
        }
    }
    return 0;
}
