// What does one wave-level gather instruction cost the CU's texture-address / L1 path (TA / TCP)?
// Written when the one-kernel step's gathers were the suspect (TA busy ~85 % of the kernel); they turned out to be hidden under a
// saturated VALU.  This probe prices a gather by index pattern, access width and table residency, with the index
// stream itself read coalesced -- 100 MB of it, which sets the probe's own floor (~19 us).
//   build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe2.hip -o tools/gather_probe2
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

template <int W> struct Vec;
template <> struct Vec<4> { using type = float4; };
template <> struct Vec<2> { using type = float2; };
template <> struct Vec<1> { using type = float; };

template <int W>
__device__ __forceinline__ float first(const typename Vec<W>::type &v) {
    if constexpr (W == 1) return v; else return v.x;
}

// every wave: `per` gather instructions, 6 independent ones in flight (what the force kernel keeps in flight)
template <int W>
__global__ __launch_bounds__(256) void gather_kernel(const unsigned *__restrict__ idx, const typename Vec<W>::type *__restrict__ table,
                                                     float *__restrict__ out, unsigned per, unsigned total_threads) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (unsigned i = 0; i < per; i += 6) {
        unsigned k[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) k[j] = idx[(size_t)(i + j) * total_threads + t];
        typename Vec<W>::type v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = table[k[j]];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += first<W>(v[j]);
    }
    if (acc == 1234.5f) out[t] = acc;
}

int main() {
    const unsigned blocks = 256 * 8, threads = blocks * 256, per = 48;
    std::vector<unsigned> h((size_t)threads * per);
    unsigned *d_idx;
    float4 *d_tab;
    float *d_out;
    const unsigned max_elems = 1u << 17; // 2 MB of float4
    hipMalloc(&d_idx, h.size() * 4);
    hipMalloc(&d_tab, (size_t)max_elems * 16);
    hipMalloc(&d_out, threads * 4);
    hipMemset(d_tab, 0, (size_t)max_elems * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Pat { const char *name; int run; int align; int gap; };
    // run: consecutive lanes sharing a base; align: base aligned to `align` elements; gap > 0: ascending with random gaps 1..gap
    const Pat pats[] = {{"random (own line per lane)", 1, 1, 0}, {"runs of 4, 64-B aligned", 4, 4, 0}, {"runs of 4, unaligned", 4, 1, 0},
                        {"runs of 8, unaligned", 8, 1, 0}, {"runs of 16, unaligned", 16, 1, 0}, {"64 consecutive", 64, 1, 0},
                        {"ascending, gaps 1-2 (sorted row)", 64, 1, 2}, {"ascending, gaps 1-4", 64, 1, 4}};
    std::mt19937 rng(1);
    for (unsigned table_elems : {1024u, 131072u}) { // 16 KB (L1-resident) and 2 MB (L2-resident)
        for (const Pat &p : pats) {
            unsigned base = 0, cur = 0;
            for (size_t i = 0; i < h.size(); ++i) {
                const unsigned lane = (unsigned)(i % threads) % 64;
                if (lane % p.run == 0) {
                    base = (unsigned)(rng() % (table_elems - 64 * (p.gap ? p.gap : 1) - 64));
                    base -= base % p.align;
                    cur = base;
                }
                if (p.gap) {
                    h[i] = cur;
                    cur += 1 + rng() % p.gap;
                } else {
                    h[i] = base + lane % p.run;
                }
            }
            hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            for (int w : {4, 2, 1}) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (w == 4) hipLaunchKernelGGL(gather_kernel<4>, dim3(blocks), dim3(256), 0, 0, d_idx, d_tab, d_out, per, threads);
                    if (w == 2) hipLaunchKernelGGL(gather_kernel<2>, dim3(blocks), dim3(256), 0, 0, d_idx, (const float2 *)d_tab, d_out, per, threads);
                    if (w == 1) hipLaunchKernelGGL(gather_kernel<1>, dim3(blocks), dim3(256), 0, 0, d_idx, (const float *)d_tab, d_out, per, threads);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = std::min(best, ms);
                }
                const double lanes = (double)threads * per;
                printf("table %4u KB  %-34s %2d-B: %7.1f us  %.2f lanes/clk/CU @2.4GHz  = %5.1f clk per 64-lane gather\n",
                       table_elems * 16 / 1024, p.name, 4 * w, best * 1e3, lanes / (best * 1e-3) / 256 / 2.4e9,
                       64.0 / (lanes / (best * 1e-3) / 256 / 2.4e9));
            }
        }
    }
    return 0;
}
