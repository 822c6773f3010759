// Is v_mfma_f64_16x16x4_f64 bit-identical to a chain of four FMAs per output, and in which k order?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, const double* C, double* D)
{
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];          // A[row l&15][k l>>4]
    const double b = B[(l >> 4) * 16 + (l & 15)];         // B[k l>>4][col l&15]
    v4f64 c;
    for (int r = 0; r < 4; r++) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}
int main()
{
    double hA[64], hB[64], hC[256], hD[256];
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dD, sizeof(hD));
    long same_up = 0, same_down = 0, same_exactsum = 0, total = 0;
    srand(1);
    for (int trial = 0; trial < 2000; trial++) {
        for (int i = 0; i < 64; i++) { hA[i] = (rand() / (double)RAND_MAX - 0.5) * 8; hB[i] = (rand() / (double)RAND_MAX - 0.5) * 8; }
        for (int i = 0; i < 256; i++) hC[i] = (rand() / (double)RAND_MAX - 0.5) * ((trial & 1) ? 1e-6 : 8);
        hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
            double up = hC[i * 16 + j], down = hC[i * 16 + j];
            long double ex = hC[i * 16 + j];
            for (int kk = 0; kk < 4; kk++) { up = fma(hA[i * 4 + kk], hB[kk * 16 + j], up); ex += (long double)hA[i * 4 + kk] * hB[kk * 16 + j]; }
            for (int kk = 3; kk >= 0; kk--) down = fma(hA[i * 4 + kk], hB[kk * 16 + j], down);
            const double d = hD[i * 16 + j];
            same_up += d == up; same_down += d == down; same_exactsum += d == (double)ex; total++;
        }
    }
    printf("outputs %ld: equal to fma chain k=0..3 %ld, k=3..0 %ld, single rounding of the long-double sum %ld\n", total, same_up, same_down, same_exactsum);
    return 0;
}
