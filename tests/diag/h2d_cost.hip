// Diagnostic: what a host-to-device copy costs the calling thread, from pageable and from pinned memory (hipMemcpyAsync on a stream,
// 200 calls per size, then one synchronize).  The solver's structure is uploaded as ~25 arrays per re-optimisation.
//   hipcc --offload-arch=gfx950 -O2 tests/diag/h2d_cost.hip -o /tmp/h2d_cost && /tmp/h2d_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main()
{
    hipStream_t s; hipStreamCreate(&s);
    const size_t sizes[] = {4096, 81920, 700000, 2000000};
    char* d = nullptr; hipMalloc((void**)&d, 4 << 20);
    char* pin = nullptr; hipHostMalloc((void**)&pin, 4 << 20, hipHostMallocDefault);
    char* pag = (char*)malloc(4 << 20); memset(pag, 1, 4 << 20); memset(pin, 1, 4 << 20);
    for (int rep = 0; rep < 2; rep++)
        for (size_t b : sizes) {
            for (int kind = 0; kind < 3; kind++) {
                hipStreamSynchronize(s);
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 200; i++) {
                    if (kind == 0) hipMemcpyAsync(d, pag, b, hipMemcpyHostToDevice, s);
                    else if (kind == 1) hipMemcpyAsync(d, pin, b, hipMemcpyHostToDevice, s);
                    else { memcpy(pin, pag, b); hipMemcpyAsync(d, pin, b, hipMemcpyHostToDevice, s); }
                }
                const auto t1 = std::chrono::steady_clock::now();
                hipStreamSynchronize(s);
                const auto t2 = std::chrono::steady_clock::now();
                if (rep == 1) printf("%8zu B  %-22s %7.1f us per call (host), %7.1f us per copy incl. the final synchronize\n", b,
                                     kind == 0 ? "pageable" : kind == 1 ? "pinned" : "memcpy to pinned + copy",
                                     std::chrono::duration<double, std::micro>(t1 - t0).count() / 200, std::chrono::duration<double, std::micro>(t2 - t0).count() / 200);
            }
        }
    return 0;
}
