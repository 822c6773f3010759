// match_kernels.hip — gfx950 kernels of the edge-estimation half.
//
//   knn2_kernel      M1  brute-force 2-NN Hamming match (feature_transformation_estimator.cpp:38,58):
//                        one lane per query descriptor (registers), train descriptors streamed through
//                        the scalar cache (wave-uniform address -> s_load), v_xor + v_bcnt accumulate,
//                        top-2 kept as packed (distance<<20 | trainIdx) keys with v_med3/v_min.
//   estimate_kernel  M2..M9  one workgroup per node pair: sensor-pair selection (:73-86), ratio test
//                        (:65-71), 3-D filter (:101-112), sort by (distance, queryIdx) (:114), gather into
//                        an LDS tile (:118-124), PROSAC with one hypothesis per lane and LDS-broadcast
//                        correspondences (:186-243), refit + recount + mse (:245-296), information (:133-137).
//
// This translation unit is compiled with -ffp-contract=off: the float pose recipe and the double
// consensus test round after every operation, exactly like the CPU oracle, so that inlier sets are
// bit-identical.  Only + - * / sqrt fabs and comparisons appear on that path.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>
#include "match_types.hpp"
#include "uzl_common.hpp"

namespace uzl {

// Diagnostic build only (-DUZL_STAMPS, tests/diag/stamps_match.sh): phase times of estimate_kernel (thread 0 of every eighth workgroup,
// 100 MHz clock; [31] counts the workgroups that stamped)
#ifdef UZL_STAMPS
__device__ unsigned long long g_mstamps[32];
#define MSTAMP_DECL unsigned long long st_prev_ = __builtin_amdgcn_s_memrealtime(); int st_i_ = 0; \
                    if ((blockIdx.x & 7) == 0 && threadIdx.x == 0) atomicAdd(&g_mstamps[31], 1ull);
#define MSTAMP() do { if ((blockIdx.x & 7) == 0 && threadIdx.x == 0) { unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); \
        atomicAdd(&g_mstamps[st_i_], n_ - st_prev_); st_prev_ = n_; } st_i_++; } while (0)
#define MSTAMP_AT(k) do { if ((blockIdx.x & 7) == 0 && threadIdx.x == 0) { unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); \
        atomicAdd(&g_mstamps[k], n_ - st_prev_); st_prev_ = n_; } } while (0)
#else
#define MSTAMP_DECL
#define MSTAMP() do { } while (0)
#define MSTAMP_AT(k) do { } while (0)
#endif

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------
// M1  2-NN Hamming
// ------------------------------------------------------------------------------------------------
// invariant best1 <= best2: new second-best = min(best2, max(best1, key))
__device__ __forceinline__ uint32_t second_of(uint32_t best1, uint32_t best2, uint32_t key)
{
    return min(best2, max(best1, key));
}

template <int W>
__global__ __launch_bounds__(kBlock) void knn2_kernel(const uint32_t* __restrict__ arena,
                                                      const Combo* __restrict__ combos,
                                                      uint2* __restrict__ knn)
{
    const Combo c = combos[blockIdx.y];
    if (c.words != W) return;                                  // other instantiation's combo
    const int q0 = blockIdx.x * kBlock;
    if (q0 >= c.nq) return;                                    // wave-uniform
    const int q = q0 + (int)threadIdx.x;
    const int qc = q < c.nq ? q : c.nq - 1;
    const uint32_t* __restrict__ qd = arena + c.desc_to_off + (size_t)qc * W;
    uint32_t qw[W];
#pragma unroll
    for (int k = 0; k < W; k += 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(qd + k);
        qw[k] = v.x; qw[k + 1] = v.y; qw[k + 2] = v.z; qw[k + 3] = v.w;
    }
    uint32_t best1 = 0xffffffffu, best2 = 0xffffffffu;
    const uint32_t* __restrict__ td = arena + c.desc_from_off;  // wave-uniform base
    const int nt = c.nt;
#pragma unroll 4
    for (int t = 0; t < nt; ++t) {
        const uint32_t* __restrict__ tr = td + (size_t)t * W;   // wave-uniform -> scalar loads
        uint32_t d = 0;
#pragma unroll
        for (int k = 0; k < W; ++k) d += __popc(qw[k] ^ tr[k]);
        const uint32_t key = (d << kIdxBits) | (uint32_t)t;
        best2 = second_of(best1, best2, key);
        best1 = min(best1, key);
    }
    if (q < c.nq) knn[c.knn_off + q] = make_uint2(best1, best2);
}

// LDS variant: the train set is staged tile by tile in LDS (coalesced 16-B loads) and every lane reads each train
// descriptor with broadcast ds_read_b128 (all lanes, same address: conflict-free).  Unlike scalar loads, LDS reads
// return in order, so the compiler pipelines them; QPL = 2 queries per lane keeps the LDS pipe (8 cycles per
// descriptor per wave) well below the VALU time (2 x 20 instructions x 2 cycles).
template <int W, int QPL>
__global__ __launch_bounds__(kBlock) void knn2_lds_kernel(const uint32_t* __restrict__ arena,
                                                          const Combo* __restrict__ combos,
                                                          uint2* __restrict__ knn)
{
    constexpr int TILE = 8192 / W;                             // descriptors per 32 KB tile
    __shared__ uint4 st[TILE * W / 4];
    const Combo c = combos[blockIdx.y];
    if (c.words != W) return;
    const int q0 = blockIdx.x * (kBlock * QPL);
    if (q0 >= c.nq) return;
    uint32_t qw[QPL][W];
    int qi[QPL];
#pragma unroll
    for (int u = 0; u < QPL; u++) {
        qi[u] = q0 + u * kBlock + (int)threadIdx.x;
        const int qc = qi[u] < c.nq ? qi[u] : c.nq - 1;
        const uint4* __restrict__ qd = reinterpret_cast<const uint4*>(arena + c.desc_to_off + (size_t)qc * W);
#pragma unroll
        for (int k = 0; k < W / 4; k++) { const uint4 v = qd[k]; qw[u][4 * k] = v.x; qw[u][4 * k + 1] = v.y; qw[u][4 * k + 2] = v.z; qw[u][4 * k + 3] = v.w; }
    }
    uint32_t best1[QPL], best2[QPL];
#pragma unroll
    for (int u = 0; u < QPL; u++) { best1[u] = 0xffffffffu; best2[u] = 0xffffffffu; }
    const uint4* __restrict__ td4 = reinterpret_cast<const uint4*>(arena + c.desc_from_off);
    for (int t0 = 0; t0 < c.nt; t0 += TILE) {
        const int tn = (c.nt - t0 < TILE) ? c.nt - t0 : TILE;
        __syncthreads();
        for (int i = threadIdx.x; i < tn * (W / 4); i += kBlock) st[i] = td4[(size_t)t0 * (W / 4) + i];
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < tn; ++t) {
            uint32_t tw[W];
#pragma unroll
            for (int k = 0; k < W / 4; k++) { const uint4 v = st[t * (W / 4) + k]; tw[4 * k] = v.x; tw[4 * k + 1] = v.y; tw[4 * k + 2] = v.z; tw[4 * k + 3] = v.w; }
#pragma unroll
            for (int u = 0; u < QPL; u++) {
                uint32_t d0 = 0, d1 = 0;
#pragma unroll
                for (int k = 0; k < W; k += 2) { d0 += __popc(qw[u][k] ^ tw[k]); d1 += __popc(qw[u][k + 1] ^ tw[k + 1]); }
                const uint32_t key = ((d0 + d1) << kIdxBits) | (uint32_t)(t0 + t);
                best2[u] = second_of(best1[u], best2[u], key);
                best1[u] = min(best1[u], key);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < QPL; u++)
        if (qi[u] < c.nq) knn[c.knn_off + qi[u]] = make_uint2(best1[u], best2[u]);
}

// ------------------------------------------------------------------------------------------------
// Matrix-core variant.  The Hamming distance matrix of two binary descriptor sets IS a GEMM:
//     d(t, q) = |t| + |q| - 2 <t, q>,   <t, q> = sum over the D bit positions of t_k q_k,
// so the D-bit descriptors are expanded to D int8 values (0 / 1) and <t, q> for a 32-train x 32-query tile is
// v_mfma_i32_32x32x32_i8 over D / 32 k-steps (exact integers; both operands take the SAME bit range [32 s + 16 h, +16)
// for k-step s and lane half h, so the hardware's k order inside a step does not matter).  The accumulator layout puts the
// query on the lane (col = lane & 31) and 16 train rows in the registers (row = (reg&3) + 8 (reg>>2) + 4 (lane>>5)): every
// lane keeps a running top-2 for ITS query over the rows it sees, three VALU instructions per distance
// (v_mad_i32_i24 builds the key (|t| - 2<t,q>) << 20 | t from the per-row word R[t] = |t| << 20 | t, v_med3_i32 + v_min_i32
// update the pair; |q| is constant per lane and is added at the very end; signed compares keep the order while the
// partial distance is negative), against 2 x 8 + 3 = 19 for the xor / popcount form; the two lane halves are merged once at the end.
// One wave owns UT x 32 queries (B fragments expanded once, loop-invariant), a 256-lane workgroup 4 UT x 32; train rows are
// expanded chunk by chunk into LDS (row stride D + 16 bytes: ds_read_b128 of 16 lanes covers all banks).
// Ties, keys and the missing-neighbour sentinel are those of knn2_lds_kernel: results are bit-identical.
// ------------------------------------------------------------------------------------------------
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

// 4 bits -> 4 bytes of 0 / V (bit i -> byte i), V = 1 << S: the nibble times 0x00204081 puts bit i at positions i, i + 7, i + 14,
// i + 21 - no two bits on one position, so no carries - and the mask keeps bit 0 of every byte; shifting the multiplier by S moves
// the kept bit to position S of its byte (15 * (0x00204081 << 7) < 2^32).
template <int S>
__device__ __forceinline__ uint32_t spread4(uint32_t nib) { return (nib * (0x00204081u << S)) & (0x01010101u << S); }
template <int S>
__device__ __forceinline__ v4i32 spread16(uint32_t bits16)
{
    v4i32 v;
    v.x = (int)spread4<S>(bits16 & 0xfu); v.y = (int)spread4<S>((bits16 >> 4) & 0xfu);
    v.z = (int)spread4<S>((bits16 >> 8) & 0xfu); v.w = (int)spread4<S>((bits16 >> 12) & 0xfu);
    return v;
}
// (written with min / max: the compiler folds it to one v_med3_i32 AND pads the MFMA -> VALU read hazard in front of it, which it does
//  not do for inline asm - an inline-asm reader of a matrix-core result returned stale values)
__device__ __forceinline__ int med3_i32(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

// The matrix cores emit the SORT KEY itself.  Train bits are expanded to int8 {0, -128}, query bits to {0, 64}: every common bit adds
// -2^13 to the accumulator, and the accumulator of train row t starts at |t| << 12 | t, so after the k-steps it holds
//     (|t| - 2 <t, q>) << 12 | t            (two's complement: ordered by (partial distance, t) under signed compares)
// - the fold is v_med3_i32 + v_min_i32, two vector instructions per distance where building the key from <t, q> took a multiply-add
// more (round 3: 96 instructions per 16 MFMAs; the fold, not the matrix pipe, bounds the kernel).  12 index bits cover 4096 train rows
// per sweep; a longer train set is swept in pieces whose winners are merged as full keys.  |q| is lane-constant and added at the end.
constexpr int kKeyBits = 12, kSweep = 1 << kKeyBits;
constexpr int kMmInvalid = 1023 << kKeyBits;      // key of a padded train row before |q| is added: above every real key

// RB: 32-row train blocks a wave works on side by side (the same query fragments against two row fragments): with UT = 1 (W = 16: the
// query fragments of one 32-query tile already take 64 VGPRs) a wave would otherwise run ONE chain of sixteen dependent MFMAs per block.
template <int W, int UT, int RB = 1>
__global__ __launch_bounds__(kBlock) void knn2_mfma_kernel(const uint32_t* __restrict__ arena,
                                                           const Combo* __restrict__ combos,
                                                           uint2* __restrict__ knn)
{
    constexpr int ROWB = W * 32 + 16;               // bytes per expanded train row in LDS (padded)
    constexpr int TR = 1024 / W;                    // train rows per chunk: 1024 descriptor words = 4 per lane
    constexpr int QPB = 4 * UT * 32;                // queries per workgroup
    __shared__ __attribute__((aligned(16))) uint8_t sA[TR * ROWB];
    __shared__ __attribute__((aligned(16))) int sR[TR];
    const Combo c = combos[blockIdx.y];
    if (c.words != W) return;
    const int q0 = blockIdx.x * QPB;
    if (q0 >= c.nq) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, h = lane >> 5;
    // ---- this lane's queries: B fragments (0 / 64 bytes) and |q|
    v4i32 bf[UT][W];
    int pa[UT], qi[UT];
#pragma unroll
    for (int u = 0; u < UT; u++) {
        qi[u] = q0 + (wv * UT + u) * 32 + col;
        const int qc = qi[u] < c.nq ? qi[u] : c.nq - 1;
        const uint4* __restrict__ qd = reinterpret_cast<const uint4*>(arena + c.desc_to_off + (size_t)qc * W);
        int pc = 0;
#pragma unroll
        for (int k = 0; k < W / 4; k++) {
            const uint4 v = qd[k];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                pc += __popc(w4[j]);
                bf[u][4 * k + j] = spread16<6>((w4[j] >> (16 * h)) & 0xffffu);
            }
        }
        pa[u] = pc;
    }
    uint32_t g1[UT], g2[UT];                        // winners so far as full keys (distance << kIdxBits | train index)
#pragma unroll
    for (int u = 0; u < UT; u++) { g1[u] = 0xffffffffu; g2[u] = 0xffffffffu; }
    const uint32_t* __restrict__ td = arena + c.desc_from_off;
    const int nt = c.nt;
    for (int base = 0; base < nt; base += kSweep) {
        const int nt_s = min(nt, base + kSweep);     // this sweep: train rows [base, nt_s)
        int b1[UT], b2[UT];
#pragma unroll
        for (int u = 0; u < UT; u++) { b1[u] = 0x7fffffff; b2[u] = 0x7fffffff; }
        // descriptor words of the next chunk are fetched while the matrix cores work on the current one
        uint32_t nxt[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int widx = i * kBlock + tid, t = base + widx / W;
            nxt[i] = (t < nt_s) ? td[(size_t)t * W + widx % W] : 0u;
        }
        for (int t0 = base; t0 < nt_s; t0 += TR) {
            __syncthreads();
            // ---- expand TR train rows: lane -> 4 descriptor words (coalesced), 32 bytes of 0 / -128 each; R[row] = |t| << 12 | (t - base)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int widx = i * kBlock + tid, row = widx / W, word = widx % W, t = t0 + row;
                const uint32_t wd = nxt[i];
                uint4* dst = reinterpret_cast<uint4*>(sA + row * ROWB + word * 32);
                const v4i32 lo = spread16<7>(wd & 0xffffu), hi = spread16<7>(wd >> 16);
                dst[0] = make_uint4((uint32_t)lo.x, (uint32_t)lo.y, (uint32_t)lo.z, (uint32_t)lo.w);
                dst[1] = make_uint4((uint32_t)hi.x, (uint32_t)hi.y, (uint32_t)hi.z, (uint32_t)hi.w);
                int pc = __popc(wd);
#pragma unroll
                for (int m = 1; m < W; m <<= 1) pc += __shfl_xor(pc, m);          // the W lanes of a row are consecutive
                if (word == 0) sR[row] = (t < nt_s) ? ((pc << kKeyBits) | (t - base)) : (kMmInvalid | ((t - base) & (kSweep - 1)));
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int widx = i * kBlock + tid, t = t0 + TR + widx / W;
                nxt[i] = (t < nt_s) ? td[(size_t)t * W + widx % W] : 0u;
            }
            __syncthreads();
            const int rows_here = min(TR, nt_s - t0);
            static_assert(TR % (32 * RB) == 0, "a chunk holds whole groups of row blocks (rows past the train set carry the invalid key)");
            for (int r0 = 0; r0 < rows_here; r0 += 32 * RB) {
                // rows of this lane's registers: 8 (reg>>2) + 4 h + (reg&3): their accumulators start at the rows' key words
                v16i32 acc[RB][UT];
#pragma unroll
                for (int rb = 0; rb < RB; rb++)
#pragma unroll
                    for (int g4 = 0; g4 < 4; g4++) {
                        const v4i32 rr = *reinterpret_cast<const v4i32*>(sR + r0 + 32 * rb + 8 * g4 + 4 * h);
#pragma unroll
                        for (int j = 0; j < 4; j++)
#pragma unroll
                            for (int u = 0; u < UT; u++) acc[rb][u][4 * g4 + j] = rr[j];
                    }
                const uint8_t* arow = sA + (r0 + col) * ROWB + 16 * h;
#pragma unroll
                for (int s_ = 0; s_ < W; s_++) {
#pragma unroll
                    for (int rb = 0; rb < RB; rb++) {
                        const v4i32 a = *reinterpret_cast<const v4i32*>(arow + 32 * rb * ROWB + 32 * s_);
#pragma unroll
                        for (int u = 0; u < UT; u++) acc[rb][u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bf[u][s_], acc[rb][u], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int rb = 0; rb < RB; rb++)
#pragma unroll
                    for (int k = 0; k < 16; k++) {
#pragma unroll
                        for (int u = 0; u < UT; u++) {
                            const int kk = acc[rb][u][k];                           // (|t| - 2 <t, q>) << 12 | t: the matrix cores built the key
                            b2[u] = med3_i32(b1[u], b2[u], kk);
                            b1[u] = min(b1[u], kk);
                        }
                    }
            }
        }
        // ---- this sweep's winners as full keys (padded rows stay the missing-neighbour sentinel), merged into the winners so far
#pragma unroll
        for (int u = 0; u < UT; u++) {
            // (a partial distance may be negative; |q| >= 2 <t, q> - |t| makes the total non-negative, and unsigned arithmetic mod 2^32
            //  carries it through: keys are merged with |q| added - monotone - and a padded row stays the missing-neighbour sentinel)
            const uint32_t qa = (uint32_t)pa[u] << kIdxBits;
            const uint32_t f1 = (b1[u] >= kMmInvalid) ? 0xffffffffu : (((uint32_t)(b1[u] >> kKeyBits) << kIdxBits) + (uint32_t)((b1[u] & (kSweep - 1)) + base) + qa);
            const uint32_t f2 = (b2[u] >= kMmInvalid) ? 0xffffffffu : (((uint32_t)(b2[u] >> kKeyBits) << kIdxBits) + (uint32_t)((b2[u] & (kSweep - 1)) + base) + qa);
            const uint32_t m1 = min(g1[u], f1);
            const uint32_t m2 = min(max(g1[u], f1), min(g2[u], f2));
            g1[u] = m1; g2[u] = m2;
        }
    }
    // ---- merge the two lane halves (same query, disjoint train rows)
#pragma unroll
    for (int u = 0; u < UT; u++) {
        const uint32_t c1 = (uint32_t)__shfl_xor((int)g1[u], 32), c2 = (uint32_t)__shfl_xor((int)g2[u], 32);
        const uint32_t m1 = min(g1[u], c1);
        const uint32_t m2 = min(max(g1[u], c1), min(g2[u], c2));
        if (h == 0 && qi[u] < c.nq) knn[c.knn_off + qi[u]] = make_uint2(m1, m2);
    }
}

// generic descriptor width (words not 8 / 16): query words re-read from L1 each step
__global__ __launch_bounds__(kBlock) void knn2_generic_kernel(const uint32_t* __restrict__ arena,
                                                              const Combo* __restrict__ combos,
                                                              uint2* __restrict__ knn)
{
    const Combo c = combos[blockIdx.y];
    if (c.words == 8 || c.words == 16) return;
    const int q0 = blockIdx.x * kBlock;
    if (q0 >= c.nq) return;
    const int q = q0 + (int)threadIdx.x;
    const int qc = q < c.nq ? q : c.nq - 1;
    const int W = c.words;
    const uint32_t* __restrict__ qd = arena + c.desc_to_off + (size_t)qc * W;
    const uint32_t* __restrict__ td = arena + c.desc_from_off;
    uint32_t best1 = 0xffffffffu, best2 = 0xffffffffu;
    for (int t = 0; t < c.nt; ++t) {
        const uint32_t* __restrict__ tr = td + (size_t)t * W;
        uint32_t d = 0;
        for (int k = 0; k < W; ++k) d += __popc(qd[k] ^ tr[k]);
        const uint32_t key = (d << kIdxBits) | (uint32_t)t;
        best2 = second_of(best1, best2, key);
        best1 = min(best1, key);
    }
    if (q < c.nq) knn[c.knn_off + q] = make_uint2(best1, best2);
}

// ------------------------------------------------------------------------------------------------
// M6a  counter-based sampling (replaces std::random_shuffle on a persistent permutation, :217-225;
//      the equivalence argument is in DESIGN.md, 'Sampling')
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31; return x;
}
__device__ __forceinline__ uint64_t stream_key(uint64_t seed, uint64_t job)
{
    return mix64(seed ^ mix64(job + 0x9e3779b97f4a7c15ULL));
}
__device__ __forceinline__ uint32_t draw_below(uint64_t key, uint32_t iter, uint32_t k, uint32_t n)
{
    const uint64_t h = mix64(key + 0x9e3779b97f4a7c15ULL * (uint64_t)(iter * 4u + k + 1u));
    return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);
}
__device__ __forceinline__ int prosac_prefix(int iter, int iterations, int m)
{
    const int n = (int)ceil(((iter + 3.) / iterations) * m);     // :217
    return n < m ? n : m;
}
// First three steps of a forward Fisher-Yates shuffle over the virtual identity array of length n
// (step s swaps position s with position j_s drawn from [s, n); a step with n - s < 2 is a no-op).
// Only positions 0, j0 and j1 can differ from the identity when step 2 runs, which gives the closed
// form below (the parity tests check it against an explicit swap-bookkeeping implementation).
__device__ __forceinline__ void sample3(uint64_t key, int iter, int n, int& s0, int& s1, int& s2)
{
    const int j0 = (n >= 2) ? (int)draw_below(key, (uint32_t)iter, 0u, (uint32_t)n) : 0;
    const int j1 = (n >= 3) ? 1 + (int)draw_below(key, (uint32_t)iter, 1u, (uint32_t)(n - 1)) : 1;
    const int j2 = (n >= 4) ? 2 + (int)draw_below(key, (uint32_t)iter, 2u, (uint32_t)(n - 2)) : 2;
    s0 = j0;
    s1 = (j1 == j0) ? 0 : j1;
    int v = j2;
    if (j2 == j0) v = 0;
    if (j2 == j1) v = (j0 == 1) ? 0 : 1;
    s2 = v;
}

// ------------------------------------------------------------------------------------------------
// M7  estimatePoseSVD (:299-314) -> pcl::TransformationFromCorrespondences [EXT], float.
//     The operation order below is this build's documented recipe ('Float pose recipe');
//     the parity tests require it to reproduce the CPU restatement bit for bit.
// ------------------------------------------------------------------------------------------------
struct PoseAcc {
    float m1x, m1y, m1z, m2x, m2y, m2z;
    float c00, c01, c02, c10, c11, c12, c20, c21, c22;
    float accw;
};
__device__ __forceinline__ void pose_init(PoseAcc& a)
{
    a.m1x = a.m1y = a.m1z = a.m2x = a.m2y = a.m2z = 0.f;
    a.c00 = a.c01 = a.c02 = a.c10 = a.c11 = a.c12 = a.c20 = a.c21 = a.c22 = 0.f;
    a.accw = 0.f;
}
__device__ __forceinline__ void pose_add(PoseAcc& a, double px, double py, double pz, double qx, double qy, double qz)
{
    const float p0 = (float)px, p1 = (float)py, p2 = (float)pz;
    const float q0 = (float)qx, q1 = (float)qy, q2 = (float)qz;
    a.accw += 1.f;
    const float alpha = 1.f / a.accw;
    const float om = 1.f - alpha;
    const float d10 = p0 - a.m1x, d11 = p1 - a.m1y, d12 = p2 - a.m1z;
    const float d20 = q0 - a.m2x, d21 = q1 - a.m2y, d22 = q2 - a.m2z;
    a.c00 = om * (a.c00 + alpha * (d20 * d10)); a.c01 = om * (a.c01 + alpha * (d20 * d11)); a.c02 = om * (a.c02 + alpha * (d20 * d12));
    a.c10 = om * (a.c10 + alpha * (d21 * d10)); a.c11 = om * (a.c11 + alpha * (d21 * d11)); a.c12 = om * (a.c12 + alpha * (d21 * d12));
    a.c20 = om * (a.c20 + alpha * (d22 * d10)); a.c21 = om * (a.c21 + alpha * (d22 * d11)); a.c22 = om * (a.c22 + alpha * (d22 * d12));
    a.m1x += alpha * d10; a.m1y += alpha * d11; a.m1z += alpha * d12;
    a.m2x += alpha * d20; a.m2y += alpha * d21; a.m2z += alpha * d22;
}

struct M3 { float m[9]; };   // row-major, only ever indexed with compile-time constants

__device__ __forceinline__ float det3f(const M3& a)
{
    const float t0 = a.m[0] * (a.m[4] * a.m[8] - a.m[5] * a.m[7]);
    const float t1 = a.m[1] * (a.m[3] * a.m[8] - a.m[5] * a.m[6]);
    const float t2 = a.m[2] * (a.m[3] * a.m[7] - a.m[4] * a.m[6]);
    return (t0 - t1) + t2;
}

template <int P, int Q>
__device__ __forceinline__ void jacobi_step(M3& W, M3& U, M3& V, float& maxdiag, bool& rotated)
{
    const float precision = 2.f * FLT_EPSILON;
    const float tiny = FLT_MIN;
    float thr = precision * maxdiag;
    if (thr < tiny) thr = tiny;
    if (!(fabsf(W.m[P * 3 + Q]) > thr || fabsf(W.m[Q * 3 + P]) > thr)) return;
    rotated = true;
    const float a = W.m[P * 3 + P], b = W.m[P * 3 + Q], c = W.m[Q * 3 + P], d = W.m[Q * 3 + Q];
    float c1, s1;
    const float t = a + d, dd = c - b;
    if (fabsf(dd) < tiny) { c1 = 1.f; s1 = 0.f; }
    else { const float u = t / dd; const float tmp = sqrtf(1.f + u * u); s1 = 1.f / tmp; c1 = u / tmp; }
    const float x = c1 * a + s1 * c;
    const float y = c1 * b + s1 * d;
    const float z = -s1 * b + c1 * d;
    float cj, sj;
    if (fabsf(y) < tiny) { cj = 1.f; sj = 0.f; }
    else {
        const float tau = (z - x) / (2.f * y);
        const float w = sqrtf(tau * tau + 1.f);
        const float tt = (tau >= 0.f) ? 1.f / (tau + w) : -1.f / (w - tau);
        cj = 1.f / sqrtf(tt * tt + 1.f);
        sj = tt * cj;
    }
    const float cl = cj * c1 + sj * s1;
    const float sl = cj * s1 - sj * c1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float wp = W.m[P * 3 + k], wq = W.m[Q * 3 + k];
        W.m[P * 3 + k] = cl * wp + sl * wq;
        W.m[Q * 3 + k] = cl * wq - sl * wp;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float up = U.m[k * 3 + P], uq = U.m[k * 3 + Q];
        U.m[k * 3 + P] = cl * up + sl * uq;
        U.m[k * 3 + Q] = cl * uq - sl * up;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float wp = W.m[k * 3 + P], wq = W.m[k * 3 + Q];
        W.m[k * 3 + P] = cj * wp - sj * wq;
        W.m[k * 3 + Q] = sj * wp + cj * wq;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float vp = V.m[k * 3 + P], vq = V.m[k * 3 + Q];
        V.m[k * 3 + P] = cj * vp - sj * vq;
        V.m[k * 3 + Q] = sj * vp + cj * vq;
    }
    const float mp = fabsf(W.m[P * 3 + P]), mq = fabsf(W.m[Q * 3 + Q]);
    if (mp > maxdiag) maxdiag = mp;
    if (mq > maxdiag) maxdiag = mq;
}

template <int A, int B>
__device__ __forceinline__ void swap_cols(M3& U, M3& V, float* S)
{
    const float ts = S[A]; S[A] = S[B]; S[B] = ts;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float tu = U.m[k * 3 + A]; U.m[k * 3 + A] = U.m[k * 3 + B]; U.m[k * 3 + B] = tu;
        const float tv = V.m[k * 3 + A]; V.m[k * 3 + A] = V.m[k * 3 + B]; V.m[k * 3 + B] = tv;
    }
}

__device__ __forceinline__ void svd3f(const M3& A, M3& U, M3& V)
{
    M3 W;
    float scale = 0.f;
#pragma unroll
    for (int i = 0; i < 9; i++) { const float a = fabsf(A.m[i]); if (a > scale) scale = a; }
    if (scale == 0.f) scale = 1.f;
#pragma unroll
    for (int i = 0; i < 9; i++) W.m[i] = A.m[i] / scale;
#pragma unroll
    for (int i = 0; i < 9; i++) { U.m[i] = (i % 4 == 0) ? 1.f : 0.f; V.m[i] = U.m[i]; }
    float maxdiag = fabsf(W.m[0]);
    if (fabsf(W.m[4]) > maxdiag) maxdiag = fabsf(W.m[4]);
    if (fabsf(W.m[8]) > maxdiag) maxdiag = fabsf(W.m[8]);
    for (int sweep = 0; sweep < 64; sweep++) {
        bool rotated = false;
        jacobi_step<1, 0>(W, U, V, maxdiag, rotated);
        jacobi_step<2, 0>(W, U, V, maxdiag, rotated);
        jacobi_step<2, 1>(W, U, V, maxdiag, rotated);
        if (!rotated) break;
    }
    float S[3];
    {
        float s = W.m[0]; if (s < 0.f) { s = -s; U.m[0] = -U.m[0]; U.m[3] = -U.m[3]; U.m[6] = -U.m[6]; } S[0] = s * scale;
        s = W.m[4]; if (s < 0.f) { s = -s; U.m[1] = -U.m[1]; U.m[4] = -U.m[4]; U.m[7] = -U.m[7]; } S[1] = s * scale;
        s = W.m[8]; if (s < 0.f) { s = -s; U.m[2] = -U.m[2]; U.m[5] = -U.m[5]; U.m[8] = -U.m[8]; } S[2] = s * scale;
    }
    // selection sort, descending, first maximum wins
    {
        int pos = 0;
        if (S[1] > S[0]) pos = 1;
        if (pos == 0) { if (S[2] > S[0]) pos = 2; } else { if (S[2] > S[1]) pos = 2; }
        if (pos == 1) swap_cols<0, 1>(U, V, S);
        else if (pos == 2) swap_cols<0, 2>(U, V, S);
        if (S[2] > S[1]) swap_cols<1, 2>(U, V, S);
    }
}

__device__ __forceinline__ void pose_finish(const PoseAcc& a, double T[12])
{
    M3 C, U, V;
    C.m[0] = a.c00; C.m[1] = a.c01; C.m[2] = a.c02;
    C.m[3] = a.c10; C.m[4] = a.c11; C.m[5] = a.c12;
    C.m[6] = a.c20; C.m[7] = a.c21; C.m[8] = a.c22;
    svd3f(C, U, V);
    const float sg = (det3f(U) * det3f(V) < 0.f) ? -1.f : 1.f;
    float R[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            R[r * 3 + c] = (U.m[r * 3 + 0] * V.m[c * 3 + 0] + U.m[r * 3 + 1] * V.m[c * 3 + 1]) + (U.m[r * 3 + 2] * sg) * V.m[c * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const float rm = (R[r * 3 + 0] * a.m1x + R[r * 3 + 1] * a.m1y) + R[r * 3 + 2] * a.m1z;
        const float m2 = (r == 0) ? a.m2x : (r == 1) ? a.m2y : a.m2z;
        const float t = m2 - rm;
        T[r * 4 + 0] = (double)R[r * 3 + 0];
        T[r * 4 + 1] = (double)R[r * 3 + 1];
        T[r * 4 + 2] = (double)R[r * 3 + 2];
        T[r * 4 + 3] = (double)t;
    }
}

// M8 distance^2 / distance of one correspondence under T (operation order of uzlo point_dist)
__device__ __forceinline__ double point_dist2(const double* __restrict__ pq, const double* T)
{
    // explicit fused multiply-adds, innermost first: the oracle's point_dist does exactly these (15 instructions per
    // point instead of 26: the vote loop is f64-VALU-issue bound)
    const double px = pq[0], py = pq[1], pz = pq[2];
    const double x = fma(T[0], px, fma(T[1], py, fma(T[2], pz, T[3])));
    const double y = fma(T[4], px, fma(T[5], py, fma(T[6], pz, T[7])));
    const double z = fma(T[8], px, fma(T[9], py, fma(T[10], pz, T[11])));
    const double dx = x - pq[3], dy = y - pq[4], dz = z - pq[5];
    return fma(dx, dx, fma(dy, dy, dz * dz));
}

// smallest double s with sqrt_rn(s) >= t: then  sqrt(d2) < t  <=>  d2 < s  (sqrt_rn is monotone),
// which removes the f64 sqrt from the vote loop without changing a single vote.
__device__ __forceinline__ double sqrt_threshold(double t)
{
    if (!(t > 0.)) return 0.;
    double y = t * t;
    for (int k = 0; k < 8 && sqrt(y) >= t; k++) y = __longlong_as_double(__double_as_longlong(y) - 1);
    for (int k = 0; k < 16 && sqrt(y) < t; k++) y = __longlong_as_double(__double_as_longlong(y) + 1);
    return y;
}

// ------------------------------------------------------------------------------------------------
// block helpers (256 threads = 4 waves of 64)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
constexpr int kEstBlock = 256;          // estimate_kernel: 4 waves; 172 VGPRs + 63 KB LDS allow two workgroups per CU
__device__ __forceinline__ int block_sum(int v, int* s_part)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < kEstBlock / 64; w++) t += s_part[w];
    return t;
}

template <bool IN_LDS>
__global__ __launch_bounds__(kEstBlock) void estimate_kernel(EstimateArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_part[kEstBlock / 64];
    __shared__ int s_misc[8];
    __shared__ double s_T[12];
    __shared__ double s_dbl[2];

    MSTAMP_DECL
    const int tid = threadIdx.x;
    const int jb = blockIdx.x;
    const Job job = A.jobs[jb];
    const RansacParams prm = A.prm;
    const int iters = prm.iterations;

    // ---- LDS carve-up (all offsets multiples of 16 B)
    uint32_t* s_keys = reinterpret_cast<uint32_t*>(smem);                       // sort_cap u32
    size_t off = (size_t)A.sort_cap * 4;
    int* s_cnt = reinterpret_cast<int*>(smem + off);                            // iters (padded to 4)
    off += (size_t)((iters + 3) & ~3) * 4;
    double* pq; double* dist; uint8_t* mask;
    if constexpr (IN_LDS) {
        pq = reinterpret_cast<double*>(smem + off); off += (size_t)prm.lds_points * 48;
        dist = reinterpret_cast<double*>(smem + off); off += (size_t)prm.lds_points * 8;
        mask = smem + off;
    } else {
        pq = A.pq_scratch + (size_t)jb * prm.max_corr * 6;
        dist = A.dist_scratch + (size_t)jb * prm.max_corr;
        mask = A.mask_scratch + (size_t)jb * prm.max_corr;
    }

    int r_n_matches = 0, r_frame_from = -1, r_frame_to = -1;

    int M = 0;
    bool have_pair = false;
    if (A.P_in == nullptr) {
        // ---- M3: sensor-pair selection: most ratio-test survivors, first wins ties (:73-86)
        int best_c = -1, best_score = -1;
        for (int ci = 0; ci < job.combo_count; ci++) {
            const Combo c = A.combos[job.combo_begin + ci];
            const uint2* __restrict__ kn = A.knn + c.knn_off;
            int cnt = 0;
            if (c.nt >= 2) {
                for (int q = tid; q < c.nq; q += kEstBlock) {
                    const uint2 k2 = kn[q];
                    const float d0 = (float)(k2.x >> kIdxBits), d1 = (float)(k2.y >> kIdxBits);
                    cnt += ((double)d0 < 0.99 * (double)d1) ? 1 : 0;                 // :67
                }
            }
            cnt = block_sum(cnt, s_part);
            if (cnt > best_score) { best_score = cnt; best_c = ci; }                  // :81
        }
        if (best_c >= 0) {
            have_pair = true;
            const Combo c = A.combos[job.combo_begin + best_c];
            r_n_matches = best_score;
            r_frame_from = c.frame_from; r_frame_to = c.frame_to;
            const uint2* __restrict__ kn = A.knn + c.knn_off;
            const uint8_t* __restrict__ vfrom = A.arena + c.valid_from_off;
            const uint8_t* __restrict__ vto = A.arena + c.valid_to_off;
            // ---- M2 + M4a: ratio test + valid_3d filter, ordered compaction of (distance, queryIdx) keys
            for (int i = tid; i < A.sort_cap; i += kEstBlock) s_keys[i] = 0xffffffffu;
            __syncthreads();
            int base = 0;
            for (int q0 = 0; q0 < c.nq; q0 += kEstBlock) {
                const int q = q0 + tid;
                bool keep = false;
                uint32_t key = 0;
                if (q < c.nq && c.nt >= 2) {
                    const uint2 k2 = kn[q];
                    const uint32_t d0 = k2.x >> kIdxBits, d1 = k2.y >> kIdxBits;
                    const uint32_t t0 = k2.x & kIdxMask;
                    const bool ratio = (double)(float)d0 < 0.99 * (double)(float)d1;
                    keep = ratio && vfrom[t0] != 0 && vto[q] != 0;                   // :104
                    key = (d0 << kIdxBits) | (uint32_t)q;
                }
                const unsigned long long bal = __ballot(keep);
                const int lane = tid & 63, wv = tid >> 6;
                const int wcnt = __popcll(bal);
                __syncthreads();
                if (lane == 0) s_part[wv] = wcnt;
                __syncthreads();
                int woff = 0;
                for (int w = 0; w < wv; w++) woff += s_part[w];
                int total = 0;
#pragma unroll
                for (int w = 0; w < kEstBlock / 64; w++) total += s_part[w];
                if (keep) s_keys[base + woff + __popcll(bal & ((1ull << lane) - 1ull))] = key;
                base += total;
            }
            M = base;
            __syncthreads();
            MSTAMP();   // 0: selection + ratio/valid compaction
            // ---- M4b: std::sort by distance (:114), order fixed to (distance, queryIdx): bitonic in LDS
            int n2 = 1;
            while (n2 < M) n2 <<= 1;
            if (n2 >= kEstBlock) {
                // Each wave owns a contiguous quarter of the array: every compare-exchange with distance j < n2 / 4 stays
                // inside one wave's quarter, where program order + the in-order LDS queue are synchronisation enough.  Only
                // the three stages with j >= n2 / 4 (of 55 at n2 = 1024) cross waves and need workgroup barriers.
                const int seg = n2 >> 2, lane = tid & 63, wbase = (tid >> 6) * seg;
                for (int k = 2; k <= n2; k <<= 1) {
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        if (j >= seg) {
                            __syncthreads();
                            for (int i = tid; i < n2; i += kEstBlock) {
                                const int ixj = i ^ j;
                                if (ixj > i) {
                                    const uint32_t a = s_keys[i], b = s_keys[ixj];
                                    const bool up = (i & k) == 0;
                                    if ((a > b) == up) { s_keys[i] = b; s_keys[ixj] = a; }
                                }
                            }
                            __syncthreads();
                        } else {
                            // pair index p -> lower element i (bit log2(j) cleared); two pairs per pass, all four reads in
                            // flight before the first compare (the stage is LDS-latency bound, not throughput bound)
                            for (int p0 = lane; p0 < (seg >> 1); p0 += 128) {
                                const int p1 = p0 + 64;
                                const bool two = p1 < (seg >> 1);
                                const int i0 = wbase + (((p0 & ~(j - 1)) << 1) | (p0 & (j - 1)));
                                const int i1 = two ? wbase + (((p1 & ~(j - 1)) << 1) | (p1 & (j - 1))) : i0;
                                const uint32_t a0 = s_keys[i0], b0 = s_keys[i0 + j], a1 = s_keys[i1], b1 = s_keys[i1 + j];
                                if ((a0 > b0) == ((i0 & k) == 0)) { s_keys[i0] = b0; s_keys[i0 + j] = a0; }
                                if (two && (a1 > b1) == ((i1 & k) == 0)) { s_keys[i1] = b1; s_keys[i1 + j] = a1; }
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's exchanges have landed before its next stage
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
                __syncthreads();
            } else {
                for (int k = 2; k <= n2; k <<= 1) {
                    for (int j = k >> 1; j > 0; j >>= 1) {
                        for (int i = tid; i < n2; i += kEstBlock) {
                            const int ixj = i ^ j;
                            if (ixj > i) {
                                const uint32_t a = s_keys[i], b = s_keys[ixj];
                                const bool up = (i & k) == 0;
                                if ((a > b) == up) { s_keys[i] = b; s_keys[ixj] = a; }
                            }
                        }
                        __syncthreads();
                    }
                }
            }
            MSTAMP();   // 1: bitonic sort
            // ---- M5: gather Xd (from/train) and Pd (to/query) (:118-124); P = Pd, Q = Xd (:130)
            const double* __restrict__ pfrom = reinterpret_cast<const double*>(A.arena) + c.pos_from_off;
            const double* __restrict__ pto = reinterpret_cast<const double*>(A.arena) + c.pos_to_off;
            for (int m = tid; m < M; m += kEstBlock) {
                const uint32_t key = s_keys[m];
                const int q = (int)(key & kIdxMask);
                const int t = (int)(kn[q].x & kIdxMask);
                pq[m * 6 + 0] = pto[3 * (size_t)q + 0]; pq[m * 6 + 1] = pto[3 * (size_t)q + 1]; pq[m * 6 + 2] = pto[3 * (size_t)q + 2];
                pq[m * 6 + 3] = pfrom[3 * (size_t)t + 0]; pq[m * 6 + 4] = pfrom[3 * (size_t)t + 1]; pq[m * 6 + 5] = pfrom[3 * (size_t)t + 2];
                if (A.corr_query && m < prm.max_corr) {
                    A.corr_query[(size_t)jb * prm.max_corr + m] = q;
                    A.corr_train[(size_t)jb * prm.max_corr + m] = t;
                    A.corr_dist[(size_t)jb * prm.max_corr + m] = (int32_t)(key >> kIdxBits);
                }
            }
        }
    } else {
        // uzl_ransac_points: correspondences supplied by the caller (estimateSVD entry, :178-184)
        have_pair = true;
        M = job.pq_count;
        r_n_matches = M;
        for (int m = tid; m < M; m += kEstBlock) {
            const size_t col = (size_t)job.pq_off + m;
#pragma unroll
            for (int r = 0; r < 3; r++) { pq[m * 6 + r] = A.P_in[3 * col + r]; pq[m * 6 + 3 + r] = A.Q_in[3 * col + r]; }
        }
    }
    __syncthreads();
    MSTAMP();           // 2: gather

    // ---- M6: PROSAC (:186-243), one hypothesis per lane, correspondences broadcast from the tile
    int max_cons = 0, best_it = -1, it_run = 0;
    if (have_pair && M >= 3) {
        const uint64_t key = stream_key(prm.seed, job.job_id);
        const double thr2 = sqrt_threshold(prm.thresh);
        const bool vote_valu = A.vote_valu != 0;             // A/B switch (UZL_VOTE_VALU=1): the 17-instruction loop on the vector ALU
        bool stop = false;
        for (int r0 = 0; r0 < iters && !stop; r0 += kEstBlock) {
            const int it = r0 + tid;
            double T[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (it < iters) {
                const int n = prm.do_prosac ? prosac_prefix(it, iters, M) : M;
                int s0, s1, s2;
                sample3(key, it, n, s0, s1, s2);
                PoseAcc acc;
                pose_init(acc);
                pose_add(acc, pq[s0 * 6 + 0], pq[s0 * 6 + 1], pq[s0 * 6 + 2], pq[s0 * 6 + 3], pq[s0 * 6 + 4], pq[s0 * 6 + 5]);
                pose_add(acc, pq[s1 * 6 + 0], pq[s1 * 6 + 1], pq[s1 * 6 + 2], pq[s1 * 6 + 3], pq[s1 * 6 + 4], pq[s1 * 6 + 5]);
                pose_add(acc, pq[s2 * 6 + 0], pq[s2 * 6 + 1], pq[s2 * 6 + 2], pq[s2 * 6 + 3], pq[s2 * 6 + 4], pq[s2 * 6 + 5]);
                pose_finish(acc, T);                                                   // :227
            }
            MSTAMP_AT(8);       // sample + float pose of this round's hypotheses (thread 0's own: the lanes run in lock step)
            if (!vote_valu) {
                // ---- votes on the f64 matrix cores (:230).  v_mfma_f64_16x16x4_f64 is, bit for bit, the chain
                // acc = fma(a_k, b_k, acc) for k = 0..3 from acc = C (measured on 512 000 random outputs), so with
                // a = (T3, T2, T1, T0) of one row of the pose and b = (1, pz, py, px) it computes exactly
                // fma(T0, px, fma(T1, py, fma(T2, pz, T3))) - point_dist2's transform - for 16 hypotheses x 16 points per
                // instruction.  Each wave votes for the 64 hypotheses its lanes hold: 4 tiles of 16, three MFMAs (x, y, z) per
                // tile and 16 points; what is left for the vector ALU per (hypothesis, point) is three subtractions, the
                // squared norm (one mul, two fma), the compare and the count: 8 instructions instead of 17.
                const int lane = tid & 63, li = lane & 15, lk = lane >> 4;
                double a[4][3];                      // [tile][row of T]: element 3 - lk of that row, of hypothesis 16 tile + li
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const int src = 16 * q + li;
                        const double v0 = __shfl(T[4 * c + 3], src), v1 = __shfl(T[4 * c + 2], src), v2 = __shfl(T[4 * c + 1], src),
                                     v3 = __shfl(T[4 * c + 0], src);
                        a[q][c] = (lk == 0) ? v0 : (lk == 1) ? v1 : (lk == 2) ? v2 : v3;
                    }
                int cnt4[4][4];
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int r = 0; r < 4; r++) cnt4[q][r] = 0;
                // (The f64 MFMA and the f64 vector instructions share the CU's f64 units on this chip - matrix and vector f64 peaks are the same
                //  78.6 TFLOP/s: a step of 12 MFMAs x 64 cycles + 112 f64 vector instructions x 4 cycles is ~1200 cycles per wave however the two
                //  are interleaved.  Double-buffering the MFMA results so that one half step's transforms run under the other half's folds
                //  changed nothing: 101.8 -> 99.2 us per workgroup, tests/diag/stamps_match.sh.)
                for (int m0 = 0; m0 < M; m0 += 16) {
                    const int m = (m0 + li < M) ? m0 + li : M - 1;
                    const bool pv = m0 + li < M;
                    const double* __restrict__ pm = pq + m * 6;
                    const double b = (lk == 0) ? 1. : pm[3 - lk];        // (1, pz, py, px)
                    const double qx = pm[3], qy = pm[4], qz = pm[5];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const v4f64 z4 = {0., 0., 0., 0.};
                        const v4f64 X = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][0], b, z4, 0, 0, 0);
                        const v4f64 Y = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][1], b, z4, 0, 0, 0);
                        const v4f64 Z = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][2], b, z4, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; r++) {                     // hypothesis 16 q + lk + 4 r, point m0 + li
                            const double dx = X[r] - qx, dy = Y[r] - qy, dz = Z[r] - qz;
                            const double d2 = fma(dx, dx, fma(dy, dy, dz * dz));
                            cnt4[q][r] += (pv && d2 < thr2) ? 1 : 0;
                        }
                    }
                }
                // the 16 lanes of one lk hold the same hypotheses' partial counts over different points
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        int c = cnt4[q][r];
                        c += __shfl_xor(c, 1); c += __shfl_xor(c, 2); c += __shfl_xor(c, 4); c += __shfl_xor(c, 8);
                        const int hit = r0 + (tid & ~63) + 16 * q + lk + 4 * r;
                        if (li == 0 && hit < iters) s_cnt[hit] = c;
                    }
            } else if (it < iters) {
                int cnt = 0;
                int m = 0;
                for (; m + 4 <= M; m += 4) {                // four correspondences per step: their LDS reads overlap
                    double c[24];
                    const double2* __restrict__ p2 = reinterpret_cast<const double2*>(pq + m * 6);   // 48-byte points: 16-B aligned
#pragma unroll
                    for (int k = 0; k < 12; k++) { const double2 v2 = p2[k]; c[2 * k] = v2.x; c[2 * k + 1] = v2.y; }
                    __builtin_amdgcn_sched_barrier(0);      // all twelve reads in flight before the first use
#pragma unroll
                    for (int u = 0; u < 4; u++) cnt += (point_dist2(c + 6 * u, T) < thr2) ? 1 : 0;   // :230
                }
                for (; m < M; m++) cnt += (point_dist2(pq + m * 6, T) < thr2) ? 1 : 0;
                s_cnt[it] = cnt;
            }
            __syncthreads();                                // votes of other lanes' hypotheses are in s_cnt
            MSTAMP_AT(9);       // votes
            // ---- the sequential bookkeeping of :233-242 over this round's votes, without the sequence.  The loop keeps a
            // running strict maximum (first index wins ties) and stops at the first NEW maximum that satisfies
            // stop(c) = c >= 3 && c > break_pct * M.  stop() is monotone in c, so the first iteration whose own count satisfies
            // it is necessarily a new maximum (an earlier count at least as large would have satisfied it first, in this
            // round or a previous one): the stopping iteration is a ballot, the maximum up to it a reduction.
            {
                const int lane = tid & 63, wv = tid >> 6;
                const int c = (it < iters) ? s_cnt[it] : -1;           // own vote (just written by this lane)
                const bool trig = c >= 3 && (double)c > prm.break_pct * (double)M;
                const unsigned long long tb = __ballot(trig);
                if (lane == 0) s_part[wv] = tb ? (wv * 64 + (int)__builtin_ctzll(tb)) : kEstBlock;
                __syncthreads();
                int first = kEstBlock;
#pragma unroll
                for (int w = kEstBlock / 64 - 1; w >= 0; w--) first = (s_part[w] < kEstBlock) ? s_part[w] : first;   // lowest wave that has one
                __syncthreads();
                const int rend = (r0 + kEstBlock < iters) ? r0 + kEstBlock : iters;
                const int last = (first < kEstBlock) ? first : rend - r0 - 1;                                 // local index of the last iteration that counts
                int key = (tid <= last && c >= 0) ? ((c << 8) | (kEstBlock - 1 - tid)) : -1;                    // max count, then lowest index
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) key = max(key, __shfl_xor(key, o));
                if (lane == 0) s_part[wv] = key;
                __syncthreads();
                int best = -1;
#pragma unroll
                for (int w = 0; w < kEstBlock / 64; w++) best = max(best, s_part[w]);
                if (tid == 0) {
                    int mc = (r0 == 0) ? 0 : s_misc[0], bi = (r0 == 0) ? -1 : s_misc[1];
                    if (best >= 0 && (best >> 8) > mc) { mc = best >> 8; bi = r0 + (kEstBlock - 1 - (best & 0xff)); }
                    s_misc[0] = mc; s_misc[1] = bi; s_misc[2] = r0 + last + 1; s_misc[3] = (first < kEstBlock) ? 1 : 0;
                }
            }
            __syncthreads();
            MSTAMP_AT(10);      // bookkeeping
            stop = s_misc[3] != 0;
        }
        max_cons = s_misc[0]; best_it = s_misc[1]; it_run = s_misc[2];
        __syncthreads();
    }

    MSTAMP();           // 3: hypotheses + votes + bookkeeping
    // ---- refit on the best consensus set (:245-258), recount, mse (:285-290)
    double Tfin[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double mse = 0.;
    int cons = 0;
    if (max_cons >= 3) {
        const uint64_t key = stream_key(prm.seed, job.job_id);
        const double thr2 = sqrt_threshold(prm.thresh);
        // hypothesis of the winning iteration, recomputed (bit-identical: same operations)
        if (tid == 0) {
            const int n = prm.do_prosac ? prosac_prefix(best_it, iters, M) : M;
            int s0, s1, s2;
            sample3(key, best_it, n, s0, s1, s2);
            PoseAcc acc;
            pose_init(acc);
            pose_add(acc, pq[s0 * 6 + 0], pq[s0 * 6 + 1], pq[s0 * 6 + 2], pq[s0 * 6 + 3], pq[s0 * 6 + 4], pq[s0 * 6 + 5]);
            pose_add(acc, pq[s1 * 6 + 0], pq[s1 * 6 + 1], pq[s1 * 6 + 2], pq[s1 * 6 + 3], pq[s1 * 6 + 4], pq[s1 * 6 + 5]);
            pose_add(acc, pq[s2 * 6 + 0], pq[s2 * 6 + 1], pq[s2 * 6 + 2], pq[s2 * 6 + 3], pq[s2 * 6 + 4], pq[s2 * 6 + 5]);
            double T[12];
            pose_finish(acc, T);
#pragma unroll
            for (int k = 0; k < 12; k++) s_T[k] = T[k];
        }
        __syncthreads();
        double T[12];
#pragma unroll
        for (int k = 0; k < 12; k++) T[k] = s_T[k];
        for (int m = tid; m < M; m += kEstBlock) mask[m] = (point_dist2(pq + m * 6, T) < thr2) ? 1 : 0;   // maxConsensusSet
        __syncthreads();
        MSTAMP();       // 4: winning hypothesis + its mask
        // pose_function(Pfinal, Qfinal, T) (:257): running mean / covariance in inlier order.  The recurrence is
        // sequential in the inliers, but its 9 covariance entries are independent of each other: lane (r, c) of wave 0
        // carries cov[r][c] together with its own copies of mean1[c] and mean2[r] and performs exactly the scalar
        // operations of pose_add for that entry (bit-identical to the single-lane loop, ~5x shorter dependency chain).
        // First, by the whole workgroup: the inliers in order (ballot-ordered compaction) with alpha = 1 / (their 1-based rank):
        // the sequential wave then walks exactly the inliers, with no per-point branch and no division on its issue slots.
        // {index, alpha} pairs live in `dist`, which is not used before the recount.
        int2* sel = reinterpret_cast<int2*>(dist);
        int n_in = 0;
        for (int m0 = 0; m0 < M; m0 += kEstBlock) {
            const int m = m0 + tid, lane = tid & 63, wv = tid >> 6;
            const bool in = m < M && mask[m] != 0;
            const unsigned long long bal = __ballot(in);
            if (lane == 0) s_part[wv] = __popcll(bal);
            __syncthreads();
            int woff = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < kEstBlock / 64; w++) { const int c = s_part[w]; tot += c; if (w < wv) woff += c; }
            if (in) {
                const int pos = n_in + woff + __popcll(bal & ((1ull << lane) - 1ull));
                // accw counts inliers in float (exact small integers): alpha = 1 / accw after the increment
                sel[pos] = make_int2(m, __float_as_int(1.f / (float)(pos + 1)));
            }
            n_in += tot;
            __syncthreads();
        }
        if (tid < 64) {
            const int rr = (tid < 9) ? tid / 3 : 0, cc = (tid < 9) ? tid % 3 : 0;
            float cov = 0.f, m1 = 0.f, m2 = 0.f;
            const int n_u = __builtin_amdgcn_readfirstlane(n_in);
            for (int k0 = 0; k0 < n_u; k0 += 8) {
                int2 e[8];
                double pd[8], qd[8];
#pragma unroll
                for (int u = 0; u < 8; u++) e[u] = sel[(k0 + u < n_u) ? k0 + u : n_u - 1];
#pragma unroll
                for (int u = 0; u < 8; u++) { pd[u] = pq[e[u].x * 6 + cc]; qd[u] = pq[e[u].x * 6 + 3 + rr]; }
                __builtin_amdgcn_sched_barrier(0);          // all LDS reads in flight before the first use
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    if (k0 + u < n_u) {                     // wave-uniform
                        const float alpha = __int_as_float(e[u].y);
                        const float om = 1.f - alpha;
                        const float d1 = (float)pd[u] - m1, d2 = (float)qd[u] - m2;
                        cov = om * (cov + alpha * (d2 * d1));
                        m1 += alpha * d1;
                        m2 += alpha * d2;
                    }
                }
            }
            const float accw = (float)n_u;
            // gather the 9 + 3 + 3 values into lane 0
            PoseAcc acc;
            acc.c00 = __shfl(cov, 0); acc.c01 = __shfl(cov, 1); acc.c02 = __shfl(cov, 2);
            acc.c10 = __shfl(cov, 3); acc.c11 = __shfl(cov, 4); acc.c12 = __shfl(cov, 5);
            acc.c20 = __shfl(cov, 6); acc.c21 = __shfl(cov, 7); acc.c22 = __shfl(cov, 8);
            acc.m1x = __shfl(m1, 0); acc.m1y = __shfl(m1, 1); acc.m1z = __shfl(m1, 2);       // lanes (0, c)
            acc.m2x = __shfl(m2, 0); acc.m2y = __shfl(m2, 3); acc.m2z = __shfl(m2, 6);       // lanes (r, 0)
            acc.accw = accw;
            if (tid == 0) {
                double Tr[12];
                pose_finish(acc, Tr);
#pragma unroll
                for (int k = 0; k < 12; k++) s_T[k] = Tr[k];
            }
        }
        __syncthreads();
        MSTAMP();       // 5: sequential refit
#pragma unroll
        for (int k = 0; k < 12; k++) Tfin[k] = s_T[k];
        // maxConsensus = consensus_function(P, Q, T, maxConsensusSet) (:258) and the distances for mse
        int c = 0;
        for (int m = tid; m < M; m += kEstBlock) {
            const double d2 = point_dist2(pq + m * 6, Tfin);
            const bool in = d2 < thr2;
            mask[m] = in ? 1 : 0;
            dist[m] = in ? sqrt(d2) : 0.;                                                // outliers add + 0.0: the sum is unchanged
            c += in ? 1 : 0;
        }
        cons = block_sum(c, s_part);
        if (tid == 0) {
            double e = 0.;
            int m = 0;
            for (; m + 16 <= M; m += 16) {                                              // :285-289, index order; loads batched
                double2 dv[8];
                const double2* __restrict__ d2p = reinterpret_cast<const double2*>(dist + m);
#pragma unroll
                for (int u = 0; u < 8; u++) dv[u] = d2p[u];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; u++) { e += dv[u].x; e += dv[u].y; }
            }
            for (; m < M; m++) e += dist[m];
            s_dbl[0] = e / cons;                                                        // :290
        }
        __syncthreads();
        mse = s_dbl[0];
        MSTAMP();       // 6: recount + mse
    } else {
        for (int m = tid; m < M; m += kEstBlock) mask[m] = 0;                              // :291-294
        __syncthreads();
    }
    if (A.inlier_mask) {
        for (int m = tid; m < M && m < prm.max_corr; m += kEstBlock) A.inlier_mask[(size_t)jb * prm.max_corr + m] = mask[m];
    }
    if (tid == 0) {
        uzl_edge_result* res = A.results + jb;
        const int ok = (have_pair && M >= 3) ? 1 : 0;                                  // :118, :156
        res->job_id = job.job_id;
        res->ok = ok;
        res->consensus = ok ? cons : 0;
        res->n_matches = r_n_matches;
        res->n_corr = M;
        res->frame_from = r_frame_from;
        res->frame_to = r_frame_to;
        res->iterations_run = it_run;
        res->best_iteration = best_it;
        res->mse = mse;
#pragma unroll
        for (int k = 0; k < 12; k++) res->T[k] = Tfin[k];
        // M9 information matrix (:133-137)
        double s = 1., sr = 1.;
        if (cons > 0 && mse > 0) { s = 0.1 * cons / mse; sr = s * 100.; }
        for (int k = 0; k < 36; k++) res->information[k] = 0.;
        res->information[0] = s; res->information[7] = s; res->information[14] = s;
        res->information[21] = sr; res->information[28] = sr; res->information[35] = sr;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void launch_knn2(const uint32_t* arena, const Combo* combos, int n_combos, int max_nq, uint2* knn,
                 bool has8, bool has16, bool has_generic, hipStream_t s)
{
    if (n_combos <= 0 || max_nq <= 0) return;
    dim3 grid((max_nq + kBlock - 1) / kBlock, n_combos);
    static const bool scalar_path = diag_flag("UZL_KNN2_SCALAR");   // A/B switch: train rows by scalar loads
    static const bool valu_path = diag_flag("UZL_KNN2_VALU");       // A/B switch: xor / popcount on the vector ALU (LDS-staged)
    if (scalar_path) {
        if (has8) hipLaunchKernelGGL(knn2_kernel<8>, grid, dim3(kBlock), 0, s, arena, combos, knn);
        if (has16) hipLaunchKernelGGL(knn2_kernel<16>, grid, dim3(kBlock), 0, s, arena, combos, knn);
    } else if (valu_path) {
        if (has8) hipLaunchKernelGGL((knn2_lds_kernel<8, 1>), grid, dim3(kBlock), 0, s, arena, combos, knn);
        if (has16) hipLaunchKernelGGL((knn2_lds_kernel<16, 1>), grid, dim3(kBlock), 0, s, arena, combos, knn);
    } else {
        // matrix-core path: 256 (W = 8) / 128 (W = 16) queries per workgroup
        static const int ut8 = diag_int("UZL_KNN2_UT", 2);                     // A/B switch: 32-query tiles per wave (W = 8)
        if (has8 && ut8 == 4) hipLaunchKernelGGL((knn2_mfma_kernel<8, 4>), dim3((max_nq + 511) / 512, n_combos), dim3(kBlock), 0, s, arena, combos, knn);
        else if (has8 && ut8 == 3) hipLaunchKernelGGL((knn2_mfma_kernel<8, 3>), dim3((max_nq + 383) / 384, n_combos), dim3(kBlock), 0, s, arena, combos, knn);
        else if (has8) hipLaunchKernelGGL((knn2_mfma_kernel<8, 2>), dim3((max_nq + 255) / 256, n_combos), dim3(kBlock), 0, s, arena, combos, knn);
        // A/B switch (diagnostic build): two 32-row blocks side by side for W = 16, i.e. two independent MFMA chains per wave instead of one.
        // Measured slower (512 pairs, 64-byte descriptors: 300 keypoints 0.0422 -> 0.0464 ms, 1000 keypoints 0.262 -> 0.309 ms): the single
        // chain was not what the kernel waited for - at 1000 keypoints W = 16 reaches the same 0.40 of the int8 peak as W = 8 with two chains
        // (vector ALU and LDS port beside the matrix pipe: DESIGN_APPENDIX.md, round 5)
        static const int rb16 = diag_int("UZL_KNN2_RB16", 1);
        if (has16 && rb16 == 2) hipLaunchKernelGGL((knn2_mfma_kernel<16, 1, 2>), dim3((max_nq + 127) / 128, n_combos), dim3(kBlock), 0, s, arena, combos, knn);
        else if (has16) hipLaunchKernelGGL((knn2_mfma_kernel<16, 1>), dim3((max_nq + 127) / 128, n_combos), dim3(kBlock), 0, s, arena, combos, knn);
    }
    if (has_generic) hipLaunchKernelGGL(knn2_generic_kernel, grid, dim3(kBlock), 0, s, arena, combos, knn);
}

size_t estimate_lds_bytes(int sort_cap, int iterations, int lds_points, bool in_lds)
{
    size_t b = (size_t)sort_cap * 4 + (size_t)((iterations + 3) & ~3) * 4;
    if (in_lds) b += (size_t)lds_points * (48 + 8 + 1);
    return (b + 15) & ~(size_t)15;
}

hipError_t launch_estimate(const EstimateArgs& a, int n_jobs, bool in_lds, size_t lds_bytes, hipStream_t s)
{
    if (n_jobs <= 0) return hipSuccess;
    if (in_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&estimate_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(estimate_kernel<true>, dim3(n_jobs), dim3(kEstBlock), lds_bytes, s, a);
    } else {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&estimate_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(estimate_kernel<false>, dim3(n_jobs), dim3(kEstBlock), lds_bytes, s, a);
    }
    return hipGetLastError();
}

}  // namespace uzl

#ifdef UZL_STAMPS
extern "C" UZL_DIAG_EXPORT int uzl_debug_read_mstamps(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(uzl::g_mstamps), sizeof(unsigned long long) * 32) != hipSuccess) return -3;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(uzl::g_mstamps), z, sizeof(z)) != hipSuccess) return -3; }
    return 0;
}
#endif
