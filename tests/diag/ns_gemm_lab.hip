// Lab for the f64 GEMM of the Newton-Schulz refinement (X' = 2 X - X T, X symmetric): the library's kernel (extracted from
// csrc/pgo_ml_kernels.hip by tests/diag/ns_gemm_lab.sh into ns_gemm_lab_old.inc) against candidate kernels, bit for bit and timed.
//   bash tests/diag/ns_gemm_lab.sh        (builds build/diag/ns_gemm_lab; run it on the GPU box)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
namespace uzl {
constexpr int kXcds = 8;
#include "ns_gemm_lab_old.inc"
#include "ns_gemm_lab_new.inc"
}  // namespace uzl

static double run(int which, int n, const double* X, const double* T, double* O, int reps)
{
    const int g = (n + 63) / 64, gtri = g * (g + 1) / 2;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    const int burst = getenv("LAB_BURST") ? atoi(getenv("LAB_BURST")) : 1;       // launches back to back per measurement (sustained clocks)
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a, 0));
        for (int q = 0; q < burst; q++) {
            if (which == 0) hipLaunchKernelGGL(uzl::ml_ns_gemm_kernel, dim3(gtri), dim3(256), 0, 0, n, X, T, O, (float*)nullptr, 0);
            else uzl::launch_new(which, n, X, T, O);
        }
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= burst; if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best * 1e3;
}
int main(int argc, char** argv)
{
    std::vector<int> sizes = {1878, 3750, 1008, 750, 1890};
    if (argc > 1) { sizes.clear(); for (int i = 1; i < argc; i++) sizes.push_back(atoi(argv[i])); }
    for (int n : sizes) {
        const size_t nn = (size_t)n * n;
        std::vector<double> hX(nn), hT(nn), o0(nn), o1(nn);
        srand(7);
        for (int i = 0; i < n; i++) for (int j = i; j < n; j++) { const double v = (rand() / (double)RAND_MAX - 0.5) * (i == j ? 4 : 0.1); hX[(size_t)i * n + j] = v; hX[(size_t)j * n + i] = v; }
        // T = A X with A symmetric would make X T symmetric; the kernels only need X symmetric for their operand reads, and the mirrored
        // stores take the upper triangle's values - any T shows the same bits in both kernels
        for (size_t i = 0; i < nn; i++) hT[i] = (rand() / (double)RAND_MAX - 0.5) * 0.2;
        if (getenv("LAB_SIMPLE")) {      // low-entropy operands: does the clock the chip holds depend on the data?
            for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { hX[(size_t)i * n + j] = 0.25 * ((i + j) % 3); hT[(size_t)i * n + j] = 0.5 * ((i * 7 + j) % 2); }
        }
        if (getenv("LAB_ZERO")) { std::fill(hX.begin(), hX.end(), 0.); std::fill(hT.begin(), hT.end(), 0.); }
        double *X, *T, *O;
        CK(hipMalloc((void**)&X, nn * 8)); CK(hipMalloc((void**)&T, nn * 8)); CK(hipMalloc((void**)&O, nn * 8));
        CK(hipMemcpy(X, hX.data(), nn * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(T, hT.data(), nn * 8, hipMemcpyHostToDevice));
        const int g = (n + 63) / 64;
        double flop = 0.;
        for (int i = 0; i < g; i++) for (int j = i; j < g; j++) flop += 2.0 * n * std::min(64, n - 64 * i) * std::min(64, n - 64 * j);
        CK(hipMemset(O, 0xff, nn * 8));
        const double us0 = run(0, n, X, T, O, 6);
        CK(hipMemcpy(o0.data(), O, nn * 8, hipMemcpyDeviceToHost));
        printf("n %5d  library kernel      %8.1f us  %5.1f TFLOP/s = %.3f of 78.6\n", n, us0, flop / us0 / 1e6, flop / us0 / 1e6 / 78.6);
        for (int which = 1; which <= uzl::kNewVariants; which++) {
            CK(hipMemset(O, 0xff, nn * 8));
            const double us = run(which, n, X, T, O, 6);
            CK(hipMemcpy(o1.data(), O, nn * 8, hipMemcpyDeviceToHost));
            size_t diff = 0; for (size_t i = 0; i < nn; i++) diff += memcmp(&o0[i], &o1[i], 8) != 0;
            printf("n %5d  candidate %d (%s) %8.1f us  %5.1f TFLOP/s = %.3f of 78.6   entries differing from the library kernel: %zu\n", n, which, uzl::new_name(which), us,
                   flop / us / 1e6, flop / us / 1e6 / 78.6, diff);
        }
        CK(hipFree(X)); CK(hipFree(T)); CK(hipFree(O));
    }
    return 0;
}
