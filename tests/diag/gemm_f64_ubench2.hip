// Inner loop of the f64 MFMA GEMM in isolation: operands re-read from LDS every k-step (a compiler barrier keeps the reads in the loop),
// 1 or 2 waves per SIMD, with and without software pipelining of the LDS reads.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tests/diag/gemm_f64_ubench2.hip -o build/diag/gemm_f64_ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0: reads of step j, wait, MFMAs of step j (what the compiler makes of the library kernel)
// MODE 1: reads of step j + 1 issued before the MFMAs of step j (register double buffer), counted waits
// MODE 2: as 1, two steps ahead
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(int iters, double* out)
{
    __shared__ double sA[32][128];
    __shared__ double sB[32][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wr = ((wv >> 1) & 3) * 32, wc = (wv & 1) * 32, sw = (lk & 1) << 4;
    for (int e = tid; e < 32 * 128; e += THREADS) sA[e >> 7][e & 127] = 1e-3 * (e % 7);
    for (int e = tid; e < 32 * 64; e += THREADS) sB[e >> 6][e & 63] = 1e-3 * (e % 5);
    __syncthreads();
    v4f64 acc[4];
    for (int a = 0; a < 4; a++) acc[a] = v4f64{0., 0., 0., 0.};
    const double* pa0 = &sA[lk][(wr + li) ^ sw];
    const double* pa1 = &sA[lk][(wr + 16 + li) ^ sw];
    const double* pb0 = &sB[lk][(wc + li) ^ sw];
    const double* pb1 = &sB[lk][(wc + 16 + li) ^ sw];
    if (MODE == 0) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k4 = 0; k4 < 32; k4 += 4) {
                const double a0 = pa0[k4 * 128], a1 = pa1[k4 * 128], b0 = pb0[k4 * 64], b1 = pb1[k4 * 64];
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[3], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
    } else {
        constexpr int AH = MODE;                 // steps ahead
        double ra0[AH + 1], ra1[AH + 1], rb0[AH + 1], rb1[AH + 1];
#pragma unroll
        for (int d = 0; d < AH; d++) { ra0[d] = pa0[d * 4 * 128]; ra1[d] = pa1[d * 4 * 128]; rb0[d] = pb0[d * 4 * 64]; rb1[d] = pb1[d * 4 * 64]; }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int nx = (j + AH) & 7;
                ra0[(j + AH) % (AH + 1)] = pa0[nx * 4 * 128]; ra1[(j + AH) % (AH + 1)] = pa1[nx * 4 * 128];
                rb0[(j + AH) % (AH + 1)] = pb0[nx * 4 * 64]; rb1[(j + AH) % (AH + 1)] = pb1[nx * 4 * 64];
                __builtin_amdgcn_sched_barrier(0);
                const double a0 = ra0[j % (AH + 1)], a1 = ra1[j % (AH + 1)], b0 = rb0[j % (AH + 1)], b1 = rb1[j % (AH + 1)];
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("" ::: "memory");
        }
    }
    double s = 0.;
    for (int a = 0; a < 4; a++) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    if (s == 12345.678) out[0] = s;
}
template <int MODE, int THREADS>
int run(const char* name, int wgs, int iters, double* d)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(wgs), dim3(THREADS), 0, 0, 2, d);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k<MODE, THREADS>), dim3(wgs), dim3(THREADS), 0, 0, iters, d);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
    }
    const double flop = (double)wgs * (THREADS / 64) * iters * 8 * 4 * 2048.;
    printf("%-58s %3d lanes x %4d wgs  %8.3f ms  %6.1f TFLOP/s = %.3f of 78.6\n", name, THREADS, wgs, best, flop / best / 1e9, flop / best / 1e9 / 78.6);
    return 0;
}
int main()
{
    double* d; CK(hipMalloc(&d, 64));
    run<0, 256>("reads of step j, then its MFMAs (1 wave per SIMD)", 256, 800, d);
    run<1, 256>("reads one step ahead (1 wave per SIMD)", 256, 800, d);
    run<2, 256>("reads two steps ahead (1 wave per SIMD)", 256, 800, d);
    run<0, 512>("reads of step j, then its MFMAs (2 waves per SIMD)", 256, 800, d);
    run<1, 512>("reads one step ahead (2 waves per SIMD)", 256, 800, d);
    run<2, 512>("reads two steps ahead (2 waves per SIMD)", 256, 800, d);
    run<0, 256>("reads of step j, then its MFMAs (2 wgs per CU)", 512, 800, d);
    run<1, 256>("reads one step ahead (2 wgs per CU)", 512, 800, d);
    return 0;
}
