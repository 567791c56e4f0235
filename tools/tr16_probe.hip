// Probe: the lane <-> element map of ds_read_b64_tr_b16 (gfx950), with exact integer data.
//   hipcc --offload-arch=gfx950 -O2 tools/tr16_probe.hip -o tools/tr16_probe && ./tools/tr16_probe
// LDS holds halfs with value = index.  Lane 16g + 4q + p of a 16-lane group supplies the address of 4 consecutive halfs
// (row q of the group's 4 x 16 block, columns 4p .. 4p+3); the claim checked is
//   lane 16g + i receives, in element q, half (i & 3) of the 8 bytes addressed by lane 16g + 4q + (i >> 2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __fp16 v4h __attribute__((__vector_size__(4 * sizeof(__fp16))));

__global__ void probe(const int *addr_halfs, float *out) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (_Float16)(float)i;
    __syncthreads();
    auto p = (__attribute__((address_space(3))) v4h *)(lds + addr_halfs[threadIdx.x]);
    v4h r = __builtin_amdgcn_ds_read_tr16_b64_v4f16(p);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)r[j];
}

int main() {
    std::vector<int> addr(64);
    // group g reads rows 4g .. 4g+3 of a [row][64 halfs] image at columns 16 (g & 1) + 4p (a different block per group)
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, q = (l >> 2) & 3, p = l & 3;
        addr[l] = (4 * g + q) * 64 + 16 * (g & 1) + 4 * p;
    }
    int *d_addr;
    float *d_out;
    hipMalloc(&d_addr, 64 * sizeof(int));
    hipMalloc(&d_out, 256 * sizeof(float));
    hipMemcpy(d_addr, addr.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    std::vector<float> out(256);
    hipMemcpy(out.data(), d_out, 256 * sizeof(float), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        for (int q = 0; q < 4; ++q) {
            const int want = addr[16 * g + 4 * q + (i >> 2)] + (i & 3);
            if ((int)out[l * 4 + q] != want) ++bad;
        }
    }
    printf("ds_read_b64_tr_b16 map: %s (%d mismatches)\n", bad ? "DIFFERENT FROM THE CLAIM" : "as claimed", bad);
    if (bad)
        for (int l = 0; l < 64; ++l)
            printf("lane %2d addr %4d -> %4.0f %4.0f %4.0f %4.0f\n", l, addr[l], out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
    return bad ? 1 : 0;
}
