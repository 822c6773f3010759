// What limits a 64 x 64-tile f64 MFMA GEMM on gfx950?  Chip-wide rates of (1) the matrix pipe alone, (2) with the LDS operand reads of
// ml_ns_gemm_kernel's inner loop, (3) with the slab hand-over (LDS stores + two barriers per 64-deep slab) but no global loads.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tests/diag/gemm_f64_ubench.hip -o build/diag/gemm_f64_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int ACC>
__global__ __launch_bounds__(256) void k(int iters, double* out)
{
    __shared__ double sA[64][64];
    __shared__ double sB[64][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32, sw = (lk & 1) << 4;
    for (int e = tid; e < 64 * 64; e += 256) { sA[e >> 6][e & 63] = 1e-3 * (e % 7); sB[e >> 6][e & 63] = 1e-3 * (e % 5); }
    __syncthreads();
    v4f64 acc[ACC];
    for (int a = 0; a < ACC; a++) acc[a] = v4f64{0., 0., 0., 0.};
    double a0 = 1e-3 * lane, a1 = 2e-3 * lane, b0 = 1e-3, b1 = 3e-3;
    double2 st = make_double2(1e-3 * tid, 2e-3);
    for (int it = 0; it < iters; it++) {          // one slab: 16 k4-steps
        if (MODE >= 3) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = u * 256 + tid, ek = e / 32, ei = (2 * (e % 32)) ^ ((ek & 1) << 4);
                *reinterpret_cast<double2*>(&sA[ek][ei]) = st;
                *reinterpret_cast<double2*>(&sB[ek][ei]) = st;
            }
            __syncthreads();
        }
#pragma unroll
        for (int k4 = 0; k4 < 64; k4 += 4) {
            if (MODE >= 2) {
                a0 = sA[k4 + lk][(wr + li) ^ sw]; a1 = sA[k4 + lk][(wr + 16 + li) ^ sw];
                b0 = sB[k4 + lk][(wc + li) ^ sw]; b1 = sB[k4 + lk][(wc + 16 + li) ^ sw];
            }
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
            acc[1 % ACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[1 % ACC], 0, 0, 0);
            acc[2 % ACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[2 % ACC], 0, 0, 0);
            acc[3 % ACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[3 % ACC], 0, 0, 0);
        }
        if (MODE >= 3) __syncthreads();
    }
    double s = 0.;
    for (int a = 0; a < ACC; a++) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    if (s == 12345.678) out[0] = s;
}

template <int MODE, int ACC>
int run(const char* name, int wgs, int iters, double* d)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE, ACC>), dim3(wgs), dim3(256), 0, 0, 2, d);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k<MODE, ACC>), dim3(wgs), dim3(256), 0, 0, iters, d);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
    }
    const double flop = (double)wgs * 4 /*waves*/ * iters * 16 * 4 * 2048.;
    printf("%-52s wgs %4d  %8.3f ms  %6.1f TFLOP/s = %.3f of 78.6\n", name, wgs, best, flop / best / 1e9, flop / best / 1e9 / 78.6);
    return 0;
}
int main()
{
    double* d; CK(hipMalloc(&d, 64));
    for (int wgs : {256, 512, 1024}) {
        run<1, 4>("MFMA only, 4 accumulators", wgs, 400, d);
        run<1, 2>("MFMA only, 2 accumulators (dependent pairs)", wgs, 400, d);
        run<2, 4>("+ LDS operand reads (4 ds_read_b64 per 4 MFMA)", wgs, 400, d);
        run<3, 4>("+ slab hand-over (16 ds_write_b128 + 2 barriers)", wgs, 400, d);
    }
    return 0;
}
