// Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE (and the raw L2 fabric-side request counters they are derived from)
// on KNOWN byte counts in the access patterns of the one-kernel force step (csrc/fused_eval.hip):
//   calib_stream_f4      268 MB read once, 16 B per lane, coalesced  (the guide's calibrated case: FETCH_SIZE = 1/2)
//   calib_stream_u32      73 MB read once,  4 B per lane, coalesced  (how the kernel reads its index rows)
//   calib_rows_u32        index rows as the kernel walks them: one wave per row of `nn` u32 out of a `pitch`-entry
//                         row, three clamped 64-lane trips (j < nn ? j : nn - 1) -- the clamped lanes re-read one word
//   calib_gather16        16-B gathers from a 2 MB table through a 73 MB coalesced index stream, indices ascending
//                         with small gaps inside a wave (a neighbor row's pattern); the table is L2-resident, so the
//                         fabric sees it once per XCD (8 x 2 MB) and the index stream once
//   calib_write_nt       268 MB written once with nontemporal 16-B stores (WRITE_SIZE's calibrated case)
//   calib_rows_write_nt  the kernel's store pattern: per row `live` of 128 float4 slots written, the tail skipped
// MI355X_MICROARCH.md (HBM): "Other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern before trusting an absolute."  The program prints the known bytes per launch as JSON; run it under
//   rocprofv3 --pmc FETCH_SIZE -- tools/fetch_calib          (and WRITE_SIZE / the raw TCC_EA0_* counters, one pass each)
// and divide (tools/fetch_calib_report.py).
//   build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

__device__ __forceinline__ void store_nt(float4 *p, const float4 &v) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z);
    __builtin_nontemporal_store(v.w, &p->w);
}

__global__ __launch_bounds__(256) void calib_stream_f4(const float4 *__restrict__ src, size_t n, float *__restrict__ out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1234.5f) out[threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void calib_stream_u32(const unsigned *__restrict__ src, size_t n, float *__restrict__ out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 0x12345678u) out[threadIdx.x] = (float)acc;
}

// one wave per row, three clamped trips -- fused_rows_group's index loads
__global__ __launch_bounds__(256) void calib_rows_u32(const unsigned *__restrict__ nlist, unsigned rows, unsigned pitch,
                                                      const unsigned *__restrict__ n_neigh, float *__restrict__ out) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= rows) return;
    const unsigned nn = n_neigh[w];
    const unsigned *nl = nlist + (size_t)w * pitch;
    unsigned acc = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const unsigned j = t * 64 + lane;
        acc += nl[j < nn ? j : nn - 1];
    }
    if (acc == 0x12345678u) out[threadIdx.x] = (float)acc;
}

__global__ __launch_bounds__(256) void calib_gather16(const unsigned *__restrict__ idx, size_t n, const float4 *__restrict__ table,
                                                      float *__restrict__ out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = table[idx[i]];
        acc += v.x + v.w;
    }
    if (acc == 1234.5f) out[threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void calib_write_nt(float4 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        store_nt(&dst[i], make_float4(1.f, 2.f, 3.f, 4.f));
}

__global__ __launch_bounds__(256) void calib_rows_write_nt(float4 *__restrict__ dst, unsigned rows, unsigned NN,
                                                           const unsigned *__restrict__ live) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= rows) return;
    const unsigned n = live[w];
    for (unsigned s = lane; s < n; s += 64) store_nt(&dst[(size_t)w * NN + s], make_float4(1.f, 2.f, 3.f, (float)s));
}

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                    \
            return 1;                                                                                  \
        }                                                                                              \
    } while (0)

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    const unsigned rows = 131072, NN = 128, pitch = 216; // C3: the stand-in's pitch at r_cut 3.0 + 0.4 (ceil(143 * 1.5 / 8) * 8)
    const size_t n_f4 = (size_t)rows * NN;                // 268 MB
    const size_t n_u32 = (size_t)rows * 139;              // 72.9 MB, the index rows' live entries
    const unsigned table = 131072;                        // 2 MB of float4
    float4 *d_f4, *d_tab;
    unsigned *d_u32, *d_rows, *d_nn, *d_live;
    float *d_out;
    CK(hipMalloc(&d_f4, n_f4 * 16));
    CK(hipMalloc(&d_tab, (size_t)table * 16));
    CK(hipMalloc(&d_u32, n_u32 * 4));
    CK(hipMalloc(&d_rows, (size_t)rows * pitch * 4));
    CK(hipMalloc(&d_nn, rows * 4));
    CK(hipMalloc(&d_live, rows * 4));
    CK(hipMalloc(&d_out, 4096));
    CK(hipMemset(d_f4, 0, n_f4 * 16));
    CK(hipMemset(d_tab, 0, (size_t)table * 16));
    CK(hipMemset(d_rows, 0, (size_t)rows * pitch * 4));
    std::mt19937 rng(7);
    std::vector<unsigned> h(n_u32), hn(rows), hl(rows);
    // ascending with gaps 1-2 inside each run of 64 (one wave-level gather), random base: a neighbor row's pattern
    unsigned cur = 0;
    for (size_t i = 0; i < n_u32; ++i) {
        if (i % 64 == 0) cur = rng() % (table - 200);
        h[i] = cur;
        cur += 1 + rng() % 2;
    }
    double rows_bytes = 0, rows_lines128 = 0, rows_lines64 = 0, live_bytes = 0;
    for (unsigned r = 0; r < rows; ++r) {
        hn[r] = 125 + rng() % 29;       // 125..153 candidates (mean 139)
        hl[r] = 80 + rng() % 31;        // 80..110 survivors (mean 95)
        rows_bytes += 4.0 * hn[r];
        const size_t b0 = (size_t)r * pitch * 4, b1 = b0 + 4 * (size_t)hn[r];
        rows_lines128 += (double)((b1 + 127) / 128 - b0 / 128);
        rows_lines64 += (double)((b1 + 63) / 64 - b0 / 64);
        live_bytes += 16.0 * hl[r];
    }
    CK(hipMemcpy(d_u32, h.data(), n_u32 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_nn, hn.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_live, hl.data(), rows * 4, hipMemcpyHostToDevice));
    const dim3 blk(256), grid_p(256 * 16), grid_r(rows / 4);
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(calib_stream_f4, grid_p, blk, 0, 0, d_f4, n_f4, d_out);
        hipLaunchKernelGGL(calib_stream_u32, grid_p, blk, 0, 0, d_u32, n_u32, d_out);
        hipLaunchKernelGGL(calib_rows_u32, grid_r, blk, 0, 0, d_rows, rows, pitch, d_nn, d_out);
        hipLaunchKernelGGL(calib_gather16, grid_p, blk, 0, 0, d_u32, n_u32, d_tab, d_out);
        hipLaunchKernelGGL(calib_write_nt, grid_p, blk, 0, 0, d_f4, n_f4);
        hipLaunchKernelGGL(calib_rows_write_nt, grid_r, blk, 0, 0, d_f4, rows, NN, d_live);
    }
    CK(hipDeviceSynchronize());
    printf("{\"calib_stream_f4\": {\"read_bytes\": %.0f}, \"calib_stream_u32\": {\"read_bytes\": %.0f}, "
           "\"calib_rows_u32\": {\"read_bytes\": %.0f, \"read_bytes_128B_lines\": %.0f, \"read_bytes_64B_lines\": %.0f}, "
           "\"calib_gather16\": {\"read_bytes\": %.0f, \"note\": \"index stream + the 2 MB table once per XCD (8 L2s)\"}, "
           "\"calib_write_nt\": {\"write_bytes\": %.0f}, \"calib_rows_write_nt\": {\"write_bytes\": %.0f}}\n",
           (double)n_f4 * 16, (double)n_u32 * 4, rows_bytes + 4.0 * rows, rows_lines128 * 128 + 4.0 * rows, rows_lines64 * 64 + 4.0 * rows,
           (double)n_u32 * 4 + 8.0 * table * 16, (double)n_f4 * 16, live_bytes + 4.0 * rows);
    return 0;
}
