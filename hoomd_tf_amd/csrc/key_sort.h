// Stable counting sort of N elements by a small key, shared by the stand-in's translation units (standin.hip, brick.hip).
#pragma once
#include "htf_common.h"

namespace htf {

// Stable counting sort over NK (16 or 32) keys with MANY members each (a decomposition plan's (destination, class) keys: tens
// of thousands of "stay, interior" particles) -- htfs_cell_sort orders the members of a cell with a one-thread insertion sort,
// right for cells of a handful of particles and quadratic here.  Tiles of 4096 elements: (1) per-tile histograms, (2) one small
// block turns them into per-(tile, key) output offsets, key-major, (3) every tile ranks its elements among equals in index order
// -- a wave ballot per key, wave totals through LDS, 256 elements per round -- and scatters i to order[offset].  Deterministic.
// `n` may live on the device (n_dev, nullable): the kernels then take min(n, *n_dev) -- a fixed-capacity array whose live
// length only the device knows (hoomd_tf_amd/brick.py) is sorted without the host reading it.
constexpr unsigned kSortTile = 4096;

__device__ __forceinline__ unsigned sort_len(unsigned n, const unsigned *n_dev) {
    return n_dev != nullptr ? min(n, *n_dev) : n;
}

template <unsigned NK, unsigned TILE = kSortTile>
__global__ __launch_bounds__(256) void key_hist_kernel(const unsigned *__restrict__ key, unsigned n_max, const unsigned *__restrict__ n_dev,
                                                       unsigned *__restrict__ tile_hist) {
    __shared__ unsigned h[NK];
    const unsigned n = sort_len(n_max, n_dev);
    if (threadIdx.x < NK) h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned base = blockIdx.x * TILE;
    for (unsigned r = 0; r < TILE / 256; ++r) {
        const unsigned i = base + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[key[i] & (NK - 1u)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < NK) tile_hist[blockIdx.x * NK + threadIdx.x] = h[threadIdx.x];
}

template <unsigned NK>
__global__ __launch_bounds__(64) void key_scan_kernel(unsigned *__restrict__ tile_hist, unsigned ntiles, unsigned *__restrict__ start) {
    // lane k < NK walks key k's tile counts; the key totals are exchanged through LDS for the key-major base
    __shared__ unsigned tot[NK];
    const unsigned k = threadIdx.x;
    unsigned sum = 0;
    if (k < NK)
        for (unsigned t = 0; t < ntiles; ++t) sum += tile_hist[t * NK + k];
    if (k < NK) tot[k] = sum;
    __syncthreads();
    if (k < NK) {
        unsigned base = 0;
        for (unsigned j = 0; j < k; ++j) base += tot[j];
        start[k] = base;
        if (k == NK - 1u) start[NK] = base + sum;
        unsigned run = base;
        for (unsigned t = 0; t < ntiles; ++t) { // counts -> offsets, in place
            const unsigned c = tile_hist[t * NK + k];
            tile_hist[t * NK + k] = run;
            run += c;
        }
    }
}

template <unsigned NK, unsigned TILE = kSortTile>
__global__ __launch_bounds__(256) void key_scatter_kernel(const unsigned *__restrict__ key, unsigned n_max, const unsigned *__restrict__ n_dev,
                                                          const unsigned *__restrict__ tile_off, unsigned *__restrict__ order) {
    __shared__ unsigned run[NK];       // next free output slot of each key in this tile
    __shared__ unsigned wcnt[4][NK];   // this round's per-wave counts
    const unsigned n = sort_len(n_max, n_dev);
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < NK) run[threadIdx.x] = tile_off[blockIdx.x * NK + threadIdx.x];
    __syncthreads();
    const unsigned base = blockIdx.x * TILE;
    for (unsigned r = 0; r < TILE / 256; ++r) {
        const unsigned i = base + r * 256 + threadIdx.x;
        const bool live = i < n;
        const unsigned my = live ? (key[i] & (NK - 1u)) : NK;
        unsigned rank = 0;
#pragma unroll
        for (unsigned k = 0; k < NK; ++k) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(my == k);
            if (my == k) rank = ballot_rank(m);
            if (lane == 0) wcnt[wave][k] = (unsigned)__popcll(m);
        }
        __syncthreads();
        if (live) {
            unsigned off = run[my] + rank;
            for (unsigned w = 0; w < wave; ++w) off += wcnt[w][my];
            order[off] = i;
        }
        __syncthreads();
        if (threadIdx.x < NK) run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        __syncthreads();
    }
}

// d_scratch: NK * ceil(n_max / TILE) words; d_start: NK + 1 words; three launches
template <unsigned NK, unsigned TILE = kSortTile>
inline int key_sort(const unsigned *d_key, unsigned n_max, const unsigned *d_n, unsigned *d_scratch, unsigned *d_start, unsigned *d_order,
                    hipStream_t s) {
    const unsigned ntiles = (n_max + TILE - 1) / TILE;
    if (ntiles) hipLaunchKernelGGL((key_hist_kernel<NK, TILE>), dim3(ntiles), dim3(256), 0, s, d_key, n_max, d_n, d_scratch);
    hipLaunchKernelGGL((key_scan_kernel<NK>), dim3(1), dim3(64), 0, s, d_scratch, ntiles, d_start);
    if (ntiles) hipLaunchKernelGGL((key_scatter_kernel<NK, TILE>), dim3(ntiles), dim3(256), 0, s, d_key, n_max, d_n, d_scratch, d_order);
    return check_launch("key_sort");
}

// Scatter with the scan folded in (round 5: inside a hipGraph a dependent launch costs >= 4.5 us whatever it computes, and the
// one-block scan computed a few hundred additions): every tile sums, for each key, the key's total over all tiles and its counts
// in the tiles before its own -- ntiles x NK words from L2 -- and derives its own offsets; block 0 also publishes start[].
// tile_hist holds COUNTS (left untouched).
template <unsigned NK, unsigned TILE>
__global__ __launch_bounds__(256) void key_scan_scatter_kernel(const unsigned *__restrict__ key, unsigned n, const unsigned *__restrict__ tile_hist,
                                                               unsigned ntiles, unsigned *__restrict__ start, unsigned *__restrict__ order) {
    __shared__ unsigned run[NK];
    __shared__ unsigned tot[NK];
    __shared__ unsigned wcnt[4][NK];
    __shared__ unsigned part[2][256];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    {
        // all 256 threads walk the table: thread t takes key t % NK of tiles t / NK, t / NK + 256 / NK, ... (coalesced rows)
        constexpr unsigned kChunks = 256u / NK;
        const unsigned k = threadIdx.x & (NK - 1u);
        unsigned total = 0, before = 0;
        for (unsigned t = threadIdx.x / NK; t < ntiles; t += kChunks) {
            const unsigned c = tile_hist[t * NK + k];
            total += c;
            before += t < blockIdx.x ? c : 0u;
        }
        part[0][threadIdx.x] = total;
        part[1][threadIdx.x] = before;
        __syncthreads();
        if (threadIdx.x < NK) {
            total = before = 0;
#pragma unroll
            for (unsigned q = 0; q < kChunks; ++q) {
                total += part[0][q * NK + threadIdx.x];
                before += part[1][q * NK + threadIdx.x];
            }
            tot[threadIdx.x] = total;
            run[threadIdx.x] = before;
        }
    }
    __syncthreads();
    if (threadIdx.x < NK) {
        unsigned base = 0;
        for (unsigned j = 0; j < threadIdx.x; ++j) base += tot[j];
        run[threadIdx.x] += base;
        if (blockIdx.x == 0) {
            start[threadIdx.x] = base;
            if (threadIdx.x == NK - 1u) start[NK] = base + tot[NK - 1u];
        }
    }
    __syncthreads();
    const unsigned base = blockIdx.x * TILE;
    for (unsigned r = 0; r < TILE / 256; ++r) {
        const unsigned i = base + r * 256 + threadIdx.x;
        const bool live = i < n;
        const unsigned my = live ? (key[i] & (NK - 1u)) : NK;
        unsigned rank = 0;
#pragma unroll
        for (unsigned k = 0; k < NK; ++k) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(my == k);
            if (my == k) rank = ballot_rank(m);
            if (lane == 0) wcnt[wave][k] = (unsigned)__popcll(m);
        }
        __syncthreads();
        if (live) {
            unsigned off = run[my] + rank;
            for (unsigned w = 0; w < wave; ++w) off += wcnt[w][my];
            order[off] = i;
        }
        __syncthreads();
        if (threadIdx.x < NK) run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
        __syncthreads();
    }
}

// the sort when the kernel that produced the keys has already left the per-tile histograms in d_scratch: ONE launch
template <unsigned NK, unsigned TILE>
inline int key_sort_from_hist(const unsigned *d_key, unsigned n_max, unsigned *d_scratch, unsigned *d_start, unsigned *d_order, hipStream_t s) {
    const unsigned ntiles = (n_max + TILE - 1) / TILE;
    hipLaunchKernelGGL((key_scan_scatter_kernel<NK, TILE>), dim3(ntiles > 0 ? ntiles : 1), dim3(256), 0, s, d_key, n_max, (const unsigned *)d_scratch,
                       ntiles, d_start, d_order);
    return check_launch("key_sort");
}

// per-tile histogram of keys a kernel has just produced: called by EVERY thread of a 256-thread block that owns tile blockIdx.x
// (TILE / 256 keys per thread, key = NK for "no element")
template <unsigned NK>
__device__ __forceinline__ void tile_hist_begin(unsigned *h) {
    if (threadIdx.x < NK) h[threadIdx.x] = 0;
    __syncthreads();
}
template <unsigned NK>
__device__ __forceinline__ void tile_hist_end(unsigned *h, unsigned *__restrict__ tile_hist) {
    __syncthreads();
    if (threadIdx.x < NK) tile_hist[blockIdx.x * NK + threadIdx.x] = h[threadIdx.x];
}

} // namespace htf
