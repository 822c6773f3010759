// Diagnostic: what a hipGraphLaunch boundary costs on this stack.  N tiny dependent kernels as (a) eager launches, (b) one captured graph,
// (c) N/4 graphs of 4 nodes, (d) N graphs of one node - wall time per kernel with the stream kept busy (the host runs ahead), and the
// host time per launch call.    hipcc --offload-arch=gfx950 -O2 tests/diag/graph_gap.hip -o /tmp/graph_gap && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(double* p, int k) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[0] * 1.0000001 + k; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    double* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 64, reps = 200;
    auto capture = [&](int nodes, hipGraphExec_t* x) {
        hipGraph_t g;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return 1;
        for (int i = 0; i < nodes; i++) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, i);
        if (hipStreamEndCapture(s, &g) != hipSuccess) return 1;
        return hipGraphInstantiate(x, g, nullptr, nullptr, 0) == hipSuccess ? 0 : 1;
    };
    hipGraphExec_t g64, g16, g4, g1;
    if (capture(64, &g64) || capture(16, &g16) || capture(4, &g4) || capture(1, &g1)) { printf("capture failed\n"); return 1; }
    struct Case { const char* name; int per; hipGraphExec_t x; } cases[] = {{"eager", 0, nullptr}, {"1 graph x 64", 64, g64}, {"4 graphs x 16", 16, g16}, {"16 graphs x 4", 4, g4}, {"64 graphs x 1", 1, g1}};
    for (auto& c : cases) {
        for (int warm = 0; warm < 2; warm++) {
            CK(hipStreamSynchronize(s));
            const double t0 = now();
            double host = 0.;
            for (int r = 0; r < reps; r++) {
                const double h0 = now();
                if (!c.x) for (int i = 0; i < N; i++) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, i);
                else for (int i = 0; i < N / c.per; i++) CK(hipGraphLaunch(c.x, s));
                host += now() - h0;
            }
            CK(hipStreamSynchronize(s));
            const double t1 = now();
            if (warm) printf("%-16s  %.2f us per kernel (wall, stream saturated)   host %.2f us per launch call\n", c.name, (t1 - t0) / (reps * N), host / (reps * (c.x ? N / c.per : N)));
        }
    }
    // latency of a short chain: 1 graph of 4 nodes + a sync, vs 4 eager + sync
    for (int mode = 0; mode < 3; mode++) {
        CK(hipStreamSynchronize(s));
        const double t0 = now();
        for (int r = 0; r < reps; r++) {
            if (mode == 0) for (int i = 0; i < 4; i++) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, s, d, i);
            else if (mode == 1) CK(hipGraphLaunch(g4, s));
            else { CK(hipGraphLaunch(g1, s)); CK(hipGraphLaunch(g1, s)); CK(hipGraphLaunch(g1, s)); CK(hipGraphLaunch(g1, s)); }
            CK(hipStreamSynchronize(s));
        }
        printf("chain of 4 + sync, %s: %.2f us\n", mode == 0 ? "eager" : (mode == 1 ? "one graph" : "four 1-node graphs"), (now() - t0) / reps);
    }
    return 0;
}
