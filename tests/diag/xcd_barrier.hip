// xcd_barrier.hip — diagnostic: what does a barrier among the workgroups of ONE XCD cost, and is a plain-store hand-off through that
// XCD's L2 visible after it?  (hipcc --offload-arch=gfx950 -O2 -o /tmp/xcd_barrier tests/diag/xcd_barrier.hip)
//
// A grid of 8 W workgroups is launched; the dispatcher deals them round-robin over the 8 XCDs.  Every workgroup reads HW_REG_XCC_ID;
// those on XCD `want` take a ticket, the others leave.  The ticket holders run `rounds` rounds of: write a word, barrier, read the
// neighbour's word (checked), barrier.  Every spin is bounded (give-up word) so that the grid always drains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15); }   // HW_REG_XCC_ID[3:0]

struct Ctl { unsigned ticket, arrive, giveup, pad; long long t0, t1; int xcc_hist[8]; unsigned flag[64]; int blk_xcc[4096]; };

// monotonic counter barrier among `n` workgroups that share an L2: lane 0 drains its stores (the vector L1 writes through), arrives
// with an L2 atomic and polls with L1-bypassing loads; everybody then drops its own L1 lines.
// variant 1: the last arriver (it sees target - 1) publishes the generation in a word of another cache line; the others poll that word
__device__ __forceinline__ bool xcd_barrier_flag(unsigned* ctr, unsigned* flag, unsigned target, unsigned gen, unsigned* giveup, int nap, int fmode)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == target) __hip_atomic_store(flag, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
                if (wall_clock64() - t0 > 2000000 || __hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break;
                }
                if (nap) __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    __syncthreads();
    if (fmode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else if (fmode == 3) asm volatile("buffer_inv sc0" ::: "memory");
    return ok;
}
__device__ __forceinline__ bool xcd_barrier(unsigned* ctr, unsigned target, unsigned* giveup)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (wall_clock64() - t0 > 2000000 || __hip_atomic_load(giveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {   // 20 ms
                __hip_atomic_store(giveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // buffer_inv sc1: this CU's L1 lines are dropped
    return ok;
}

__global__ __launch_bounds__(256) void census(Ctl* c)
{
    if (threadIdx.x == 0) { const int x = xcc_id(); atomicAdd(&c->xcc_hist[x & 7], 1); if (blockIdx.x < 4096) c->blk_xcc[blockIdx.x] = x; }
}

__global__ __launch_bounds__(256) void rounds_kernel(Ctl* c, int want, int W, int rounds, double* buf, int words, int* errs, int variant, int fmode)
{
    __shared__ int s_me;
    if (xcc_id() != want) return;
    if (threadIdx.x == 0) s_me = (int)atomicAdd(&c->ticket, 1u);
    __syncthreads();
    const int me = s_me;
    if (me >= W) return;
    unsigned target = 0, gen = 0;
    int bad = 0;
    bool ok = true;
    if (me == 0 && threadIdx.x == 0) c->t0 = wall_clock64();
    for (int it = 0; it < rounds && ok; it++) {
        for (int w = threadIdx.x; w < words; w += blockDim.x) buf[(size_t)me * words + w] = (double)(it * 4096 + me) + 1e-3 * w;
        target += W; gen++; ok = variant ? xcd_barrier_flag(&c->arrive, &c->flag[32], target, gen, &c->giveup, variant == 2, fmode) : xcd_barrier(&c->arrive, target, &c->giveup);
        if (!ok) break;
        const int nb = (me + 1 + it % (W > 1 ? W - 1 : 1)) % W;
        for (int w = threadIdx.x; w < words; w += blockDim.x)
        {
            const double got = (fmode == 1) ? __hip_atomic_load(&buf[(size_t)nb * words + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : buf[(size_t)nb * words + w];
            if (got != (double)(it * 4096 + nb) + 1e-3 * w) bad++;
        }
        target += W; gen++; ok = variant ? xcd_barrier_flag(&c->arrive, &c->flag[32], target, gen, &c->giveup, variant == 2, fmode) : xcd_barrier(&c->arrive, target, &c->giveup);
    }
    if (me == 0 && threadIdx.x == 0) c->t1 = wall_clock64();
    if (bad) atomicAdd(errs, bad);
}

int main()
{
    Ctl* c; double* buf; int* errs;
    CK(hipMalloc(&c, sizeof(Ctl))); CK(hipMalloc(&buf, 256 * 4096 * sizeof(double))); CK(hipMalloc(&errs, 4));
    Ctl h;
    for (int grid : {256, 1000, 1024}) {
        CK(hipMemset(c, 0, sizeof(Ctl)));
        hipLaunchKernelGGL(census, dim3(grid), dim3(256), 0, 0, c);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
        int rr = 0; for (int b = 0; b < grid && b < 4096; b++) rr += ((h.blk_xcc[b] - h.blk_xcc[0] + 8) % 8 == b % 8);
        printf("census grid %d: per XCC", grid); for (int x = 0; x < 8; x++) printf(" %d", h.xcc_hist[x]);
        printf("; blocks with xcc == (xcc(block 0) + b) mod 8: %d of %d\n", rr, grid);
    }
    for (int fmode : {0, 1, 2, 3}) for (int variant : {1}) for (int W : {8, 32, 64}) for (int words : {6, 1024}) {
        const int rounds = 2000;
        CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipMemset(errs, 0, 4));
        hipLaunchKernelGGL(rounds_kernel, dim3(8 * W), dim3(256), 0, 0, c, 0, W, rounds, buf, words, errs, variant, fmode);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
        int e; CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost));
        printf("fence mode %d variant %d: W %2d workgroups on XCC 0 (tickets taken %u), %4d doubles per workgroup per round: %.2f us per (write, barrier, read, barrier) round, wrong words %d, give-up %u\n",
               fmode, variant, W, h.ticket, words, (h.t1 - h.t0) * 1e-2 / rounds, e, h.giveup);
    }
    return 0;
}
