// What does one DEPENDENT trivial kernel cost on this box -- launched eagerly on a stream by a host loop that does nothing else,
// and as a node of a hipGraph?  (The decomposed step is ~40 such kernels around one 12 us force kernel per ten steps.)
//   build: hipcc -O3 --offload-arch=gfx950 tools/launch_floor_probe.hip -o tools/launch_floor_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void bump(float *x, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += 1.0f;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int K = 2000;
    for (int n : {64, 16384, 1 << 20}) {
        float *x;
        hipMalloc(&x, sizeof(float) * n);
        hipMemset(x, 0, sizeof(float) * n);
        hipStream_t s;
        hipStreamCreate(&s);
        const int grid = (n + 255) / 256;
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(bump, dim3(grid), dim3(256), 0, s, x, n);
        hipStreamSynchronize(s);
        // eager
        double t0 = now();
        for (int i = 0; i < K; ++i) hipLaunchKernelGGL(bump, dim3(grid), dim3(256), 0, s, x, n);
        double t_issue = now() - t0;
        hipStreamSynchronize(s);
        double t_eager = now() - t0;
        // graph of K nodes
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < K; ++i) hipLaunchKernelGGL(bump, dim3(grid), dim3(256), 0, s, x, n);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        t0 = now();
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        double t_graph = now() - t0;
        // graphs of 40 nodes, launched back to back
        hipGraph_t g2;
        hipGraphExec_t ge2;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(bump, dim3(grid), dim3(256), 0, s, x, n);
        hipStreamEndCapture(s, &g2);
        hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
        hipGraphLaunch(ge2, s);
        hipStreamSynchronize(s);
        t0 = now();
        for (int i = 0; i < K / 40; ++i) hipGraphLaunch(ge2, s);
        hipStreamSynchronize(s);
        double t_graph40 = now() - t0;
        printf("n = %8d: eager %.2f us per kernel (host issue %.2f), one graph of %d nodes %.2f us per node, graphs of 40 nodes %.2f us per node\n", n,
               1e6 * t_eager / K, 1e6 * t_issue / K, K, 1e6 * t_graph / K, 1e6 * t_graph40 / K);
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
        hipGraphExecDestroy(ge2);
        hipGraphDestroy(g2);
        hipFree(x);
    }
    return 0;
}
