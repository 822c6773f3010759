// gate_kernels.hip — device side of the edge acceptance gate (graph_slam/src/graph_slam_node.cpp:779-829,
// 1064-1085; graph_slam_common/src/slam_graph.cpp:838-890) for gfx950.
//
// One lane per candidate edge: thresholds on score / transform, then SlamGraph::astar between the two nodes over the
// valid-edge adjacency (CSR), then the plausibility test of checkEdgeHeuristic.  The search is the reference's: the
// heap priority is heuristic_cost(u, target) alone (greedy best-first), stale heap entries are re-expanded, the
// reported distance is the length of the path found.  Irregular, latency-bound integer/pointer work - parallel over
// candidates, nothing to tile; every lane owns its g-score / state / heap arrays in HBM.  Same operation order as the
// CPU checker and -ffp-contract=off, so distances and verdicts are bit-identical.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>
#include <mutex>
#include "gate_types.hpp"

namespace uzl {

namespace {

__device__ __forceinline__ double node_dist(const double* __restrict__ poses, int a, int b)
{
    const double* A = poses + 12 * (size_t)a;
    const double* B = poses + 12 * (size_t)b;
    const double dx = A[3] - B[3], dy = A[7] - B[7], dz = A[11] - B[11];
    return 1. * sqrt((dx * dx + dy * dy) + dz * dz);
}

__device__ __forceinline__ bool hless(const GateHeapEnt& a, const GateHeapEnt& b)
{
    return a.w < b.w || (a.w == b.w && a.v < b.v);
}

// Eigen::Quaterniond(R) then AngleAxisd::angle() = 2 acos(clamp(w)) [EXT, Eigen 3.2]
__device__ __forceinline__ double angle_of(const double* m)
{
    double q0, q1, q2, q3;
    double t = m[0] + m[4] + m[8];
    if (t > 0.) {
        t = sqrt(t + 1.0);
        q0 = 0.5 * t;
        t = 0.5 / t;
        q1 = (m[7] - m[5]) * t; q2 = (m[2] - m[6]) * t; q3 = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
        double qv[3];
        qv[i] = 0.5 * t;
        t = 0.5 / t;
        q0 = (m[k * 3 + j] - m[j * 3 + k]) * t;
        qv[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        qv[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        q1 = qv[0]; q2 = qv[1]; q3 = qv[2];
    }
    const double n2 = (q1 * q1 + q2 * q2) + q3 * q3;
    if (n2 < 1e-12 * 1e-12) return 0.;
    double w = q0;
    if (w < -1.) w = -1.;
    if (w > 1.) w = 1.;
    return 2. * acos(w);
}

// diff_pose = pose(source)^-1 pose(target): its translation norm [m] and rotation angle [deg] (checkEdgeHeuristic, :1069-1072)
__device__ __forceinline__ void pose_gap(const double* __restrict__ poses, int source, int target, double& dn, double& drot)
{
    const double* A = poses + 12 * (size_t)source;
    const double* B = poses + 12 * (size_t)target;
    double Rd[9], ti[3], td[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) Rd[r * 3 + cc] = (A[0 * 4 + r] * B[0 * 4 + cc] + A[1 * 4 + r] * B[1 * 4 + cc]) + A[2 * 4 + r] * B[2 * 4 + cc];
        ti[r] = -((A[0 * 4 + r] * A[3] + A[1 * 4 + r] * A[7]) + A[2 * 4 + r] * A[11]);
    }
#pragma unroll
    for (int r = 0; r < 3; r++) td[r] = ((A[0 * 4 + r] * B[3] + A[1 * 4 + r] * B[7]) + A[2 * 4 + r] * B[11]) + ti[r];
    dn = sqrt((td[0] * td[0] + td[1] * td[1]) + td[2] * td[2]);
    drot = 180. * angle_of(Rd) / M_PI;
}
// Is checkEdgeHeuristic's verdict already known without the search?  The verdict is `true` when the target is not reachable, and
// otherwise when  2 ssf dist + 1 > |t|  and  10 ssf dist + 30 > angle  for the path length `dist` the search finds.  Every path is
// at least as long as the straight line between its end nodes (its pieces are the straight lines between consecutive nodes), both
// tests are monotone in dist (also in floating point: a product with a non-negative constant and a sum), so if they hold for the
// straight line - shortened by 1e-9 of itself, a thousand times the rounding a 20 000-piece sum can collect - they hold for whatever
// the search would return, and so does `true`.  Only a caller that wants the path length itself needs the search then.
__device__ __forceinline__ bool gate_decided(const double* __restrict__ poses, int source, int target, double ssf)
{
    if (!(ssf >= 0.)) return false;
    double dn, drot;
    pose_gap(poses, source, target, dn, drot);
    const double* A = poses + 12 * (size_t)source;
    const double* B = poses + 12 * (size_t)target;
    const double dx = B[3] - A[3], dy = B[7] - A[7], dz = B[11] - A[11];
    const double lower = sqrt((dx * dx + dy * dy) + dz * dz) * (1. - 1e-9);
    return (2 * ssf * lower + 1.0 > dn) && (10 * ssf * lower + 30.0 > drot);
}

}  // namespace

__global__ __launch_bounds__(kGateBlock) void gate_kernel(GateArgs a)
{
    const int k = blockIdx.x * kGateBlock + threadIdx.x;
    if (k >= a.n_query) return;
    if (!a.run[k]) {
        if (!a.keep_unrun) { a.pre_ok[k] = 0; a.heur_ok[k] = 0; a.dist[k] = -1.; }
        return;
    }
    a.pre_ok[k] = 0; a.heur_ok[k] = 0; a.dist[k] = -1.;
    const uzl_gate_edge c = a.cand[k];
    if (!(c.matching_score >= a.min_score)) return;                                  // :798
    const double* T = c.transform;
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const double diff_rot = fabs(angle_of(R)) * 180 / M_PI;                           // :800-801
    const double tn = sqrt((T[3] * T[3] + T[7] * T[7]) + T[11] * T[11]);
    if (!(tn <= a.max_T && diff_rot <= a.max_R)) return;                             // :803
    a.pre_ok[k] = 1;
    if (a.skip_decided && gate_decided(a.poses, c.from, c.to, a.ssf)) { a.heur_ok[k] = 1; a.dist[k] = -2.; return; }

    // ---- SlamGraph::astar(source = from, target = to)
    const int source = c.from, target = c.to, n = a.n;
    double* __restrict__ gs = a.gs + (size_t)k * n;
    uint8_t* __restrict__ st = a.st + (size_t)k * n;
    GateHeapEnt* __restrict__ heap = a.heap + (size_t)k * a.heap_cap;
    int hn = 0, n_open = 1;
    gs[source] = 0.; st[source] = 1;
    heap[hn].w = node_dist(a.poses, source, target); heap[hn].v = source; hn++;
    bool success = false, over = false;
    while (n_open > 0 && hn > 0) {
        const int v = heap[0].v;
        if (v == target) { success = true; break; }
        heap[0] = heap[--hn];
        for (int i = 0;;) {
            const int l = 2 * i + 1, r = l + 1;
            int m = i;
            if (l < hn && hless(heap[l], heap[m])) m = l;
            if (r < hn && hless(heap[r], heap[m])) m = r;
            if (m == i) break;
            const GateHeapEnt t = heap[i]; heap[i] = heap[m]; heap[m] = t; i = m;
        }
        if (st[v] == 1) n_open--;
        st[v] = 2;
        const double gv = gs[v];
        for (int q = a.adj_ptr[v]; q < a.adj_ptr[v + 1]; q++) {
            const int u = a.adj_nbr[q];
            if (st[u] == 2) continue;
            const double tent = gv + node_dist(a.poses, v, u);
            if (st[u] != 1 || tent < gs[u]) {
                gs[u] = tent;
                if (hn >= a.heap_cap) { over = true; break; }
                int i = hn++;
                heap[i].w = node_dist(a.poses, u, target); heap[i].v = u;
                while (i > 0) {
                    const int p = (i - 1) / 2;
                    if (!hless(heap[i], heap[p])) break;
                    const GateHeapEnt t = heap[i]; heap[i] = heap[p]; heap[p] = t; i = p;
                }
                if (st[u] != 1) { st[u] = 1; n_open++; }
            }
        }
        if (over) break;
    }
    if (over) { atomicExch(a.overflow, 1); return; }
    const double dist = success ? gs[target] : DBL_MAX;
    a.dist[k] = dist;
    // ---- checkEdgeHeuristic (:1064-1085)
    bool ok = true;
    if (dist != DBL_MAX) {
        double dn, drot;
        pose_gap(a.poses, source, target, dn, drot);
        ok = (2 * a.ssf * dist + 1.0 > dn) && (10 * a.ssf * dist + 30.0 > drot);     // :1074-1075
    }
    a.heur_ok[k] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------
// The same search, one WAVE per candidate.  A greedy best-first search is a chain of dependent steps (pop -> node ->
// neighbours -> push), ~10^4 of them between two nodes that are far apart on the odometry chain; a lane that keeps its
// heap, g-scores and states in HBM pays 5-6 dependent memory round trips per step.  Here the open list lives in LDS
// (pop = wave-wide arg-min over (h, node id): the pop ORDER of a priority queue depends only on the multiset of its
// entries, so any implementation returns the reference's sequence), a node's position + degree + first neighbours come
// in one 64-byte record, and the neighbours of the popped node are handled by one lane each (state + g-score in one
// 16-byte record): two dependent round trips per step.  Same arithmetic as gate_kernel (node_dist operand order,
// -ffp-contract=off): distances and verdicts are bit-identical (tests/test_gate_gpu.py runs both).
__device__ __forceinline__ double rec_dist(double ax, double ay, double az, double bx, double by, double bz)
{
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    return 1. * sqrt((dx * dx + dy * dy) + dz * dz);
}

__global__ __launch_bounds__(64) void gate_wave_kernel(GateWaveArgs a)
{
    __shared__ double sw[kGateOpenCap];
    __shared__ int32_t sv[kGateOpenCap];
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= a.n_query) return;
    if (!a.run[k] && a.keep_unrun) return;
    if (lane == 0) { a.pre_ok[k] = 0; a.heur_ok[k] = 0; a.dist[k] = -1.; a.redo[k] = 0; }
    if (!a.run[k]) return;
    const uzl_gate_edge c = a.cand[k];
    if (!(c.matching_score >= a.min_score)) return;                                  // :798
    const double* T = c.transform;
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const double diff_rot = fabs(angle_of(R)) * 180 / M_PI;                           // :800-801
    const double tn = sqrt((T[3] * T[3] + T[7] * T[7]) + T[11] * T[11]);
    if (!(tn <= a.max_T && diff_rot <= a.max_R)) return;                             // :803
    if (lane == 0) a.pre_ok[k] = 1;
    if (a.skip_decided && gate_decided(a.poses, c.from, c.to, a.ssf)) { if (lane == 0) { a.heur_ok[k] = 1; a.dist[k] = -2.; } return; }

    // ---- SlamGraph::astar(source = from, target = to)
    const int source = c.from, target = c.to, n = a.n;
    GateState* __restrict__ gst = a.gst + (size_t)k * n;
    const GateNodeRec tr = a.rec[target];
    int n_list = 1, n_open = 1;
    if (lane == 0) {
        const GateNodeRec sr = a.rec[source];
        sw[0] = rec_dist(sr.px, sr.py, sr.pz, tr.px, tr.py, tr.pz); sv[0] = source;
        gst[source].g = 0.; gst[source].st = 1;
    }
    __syncthreads();
    bool success = false, over = false;
    double g_target = 0.;
    while (n_open > 0 && n_list > 0) {
        // pop: arg-min over (w, v)
        double bw = DBL_MAX; int bv = 0x7fffffff, bi = -1;
        for (int i = lane; i < n_list; i += 64) {
            const double w = sw[i]; const int v = sv[i];
            if (w < bw || (w == bw && v < bv)) { bw = w; bv = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const double ow = __shfl_xor(bw, o); const int ov = __shfl_xor(bv, o), oi = __shfl_xor(bi, o);
            if (oi >= 0 && (bi < 0 || ow < bw || (ow == bw && (ov < bv || (ov == bv && oi < bi))))) { bw = ow; bv = ov; bi = oi; }
        }
        const int v = bv;
        if (v == target) { success = true; g_target = gst[target].g; break; }
        // both loads depend only on v: one round trip
        const GateNodeRec vr = a.rec[v];
        const GateState vs = gst[v];
        __syncthreads();                                   // every lane has read the list
        if (lane == 0) { sw[bi] = sw[n_list - 1]; sv[bi] = sv[n_list - 1]; }
        n_list--;
        __syncthreads();                                   // the hole is filled before any push lands on the old last slot
        if (vs.st == 1) n_open--;
        if (lane == 0) gst[v].st = 2;
        const double gv = vs.g;
        const int vdeg = vr.deg & ~kGateRecMulti;
        for (int base = 0; base < vdeg; base += 64) {
            const int q = base + lane;
            const bool has = q < vdeg;
            int u = -1;
            if (has) {
                if (q < kGateRecNbr) {                     // select chain: a run-time index into a register array would go through scratch
                    u = vr.nbr[0];
#pragma unroll
                    for (int j = 1; j < kGateRecNbr; j++) u = (q == j) ? vr.nbr[j] : u;
                } else {
                    u = a.adj_nbr[vr.adj + q];
                }
            }
            // a neighbour listed twice (multi-edge): only its first occurrence acts, as in the sequential loop (the second sees
            // the state and g-score the first one wrote and changes nothing)
            bool first = has;
            for (int j = 0; j < 64 && base + j < vdeg; j++) {
                const int uj = __shfl(u, j);
                if (j < lane && uj == u) first = false;
            }
            bool push = false;
            double hw = 0., tent = 0.;
            bool was_open = false;
            if (first && u != v) {
                const GateState us = gst[u];
                const GateNodeRec ur = a.rec[u];
                if (us.st != 2) {
                    tent = gv + rec_dist(vr.px, vr.py, vr.pz, ur.px, ur.py, ur.pz);
                    if (us.st != 1 || tent < us.g) {
                        push = true; was_open = us.st == 1;
                        hw = rec_dist(ur.px, ur.py, ur.pz, tr.px, tr.py, tr.pz);
                    }
                }
            }
            const unsigned long long m = __ballot(push);
            const int cnt = __popcll(m);
            if (n_list + cnt > kGateOpenCap) { over = true; break; }
            if (push) {
                const int pos = n_list + __popcll(m & ((1ull << lane) - 1ull));
                sw[pos] = hw; sv[pos] = u;
                gst[u].g = tent; gst[u].st = 1;
            }
            n_list += cnt;
            n_open += __popcll(__ballot(push && !was_open));
            __syncthreads();                               // a hub's next 64 neighbours see the states and g-scores this chunk wrote (a neighbour listed in two chunks)
        }
        if (over) break;
        __syncthreads();                                   // pushes and the removal are visible to the next pop
    }
    if (over) { if (lane == 0) a.redo[k] = 1; return; }
    if (lane != 0) return;
    const double dist = success ? g_target : DBL_MAX;
    a.dist[k] = dist;
    // ---- checkEdgeHeuristic (:1064-1085)
    bool ok = true;
    if (dist != DBL_MAX) {
        double dn, drot;
        pose_gap(a.poses, source, target, dn, drot);
        ok = (2 * a.ssf * dist + 1.0 > dn) && (10 * a.ssf * dist + 30.0 > drot);     // :1074-1075
    }
    a.heur_ok[k] = ok ? 1 : 0;
}


// ------------------------------------------------------------------------------------------------------------------
// The same search once more, built around what a step costs on this machine.  The CPU checker expands a node in ~40 ns (cache-resident
// pointer chasing); gate_wave_kernel needs 2 us - two dependent HBM round trips, three workgroup barriers, a six-level shuffle tree
// (ds_bpermute) per pop - and the longest search of a call (10^4 expansions between two nodes far apart on the chain) sets the time of
// the call.  Measured on config 5 the open list never holds more than ~80 entries.  So here:
//   * the open list lives in REGISTERS, kGateRegSlots entries per lane (h, node, g); pop = lane-local minimum, then a wave minimum of
//     the 64-bit key through two DPP reductions on its halves (row rotations and broadcasts, no LDS crossbar), winner by ballot;
//     entries of one node have identical keys (h, node), so the pop takes the one with the smallest g, which IS gs[node] of the
//     sequential code (a node is pushed again only with a smaller g);
//   * closed / open flags are two bitmaps of n bits in LDS; the g of a closed node (needed only when one of its stale entries is
//     popped later - the reference re-expands it) goes to HBM at close and comes back on that rare path;
//   * node records come through a direct-mapped LDS cache of 64 blocks of 8 consecutive records: ids are time-ordered, so the
//     neighbours along the chain arrive with the block;
//   * a step is two LDS round trips (everything addressed by the popped node, then everything addressed by its neighbours), no
//     barrier: one wave's LDS traffic is processed in order.
// A search whose list outgrows the registers is redone by gate_kernel.  Same arithmetic, same pop order (a priority queue's pop order
// depends only on the multiset of its keys): distances and verdicts are bit-identical to gate_kernel / the CPU checker.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
}
// minimum over the wave, the same value in every lane (rocPRIM's gfx9 pattern: quad swaps, row rotations, row broadcasts; lane 63 ends with it)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    unsigned t;
    t = dpp_u32<0xb1>(v); v = t < v ? t : v;                 // quad_perm [1,0,3,2]
    t = dpp_u32<0x4e>(v); v = t < v ? t : v;                 // quad_perm [2,3,0,1]
    t = dpp_u32<0x124>(v); v = t < v ? t : v;                // row_ror:4
    t = dpp_u32<0x128>(v); v = t < v ? t : v;                // row_ror:8
    t = dpp_u32<0x142>(v); v = t < v ? t : v;                // row_bcast:15
    t = dpp_u32<0x143>(v); v = t < v ? t : v;                // row_bcast:31
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// minimum of non-negative doubles (their bit patterns order like unsigned integers); `mine` = this lane holds it
__device__ __forceinline__ double wave_min_pos_f64(double x, bool& mine)
{
    const unsigned long long key = (unsigned long long)__double_as_longlong(x);
    const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
    const unsigned mhi = wave_min_u32(hi);
    const unsigned long long top = __ballot(hi == mhi);
    unsigned mlo;
    if (__popcll(top) == 1) mlo = (unsigned)__builtin_amdgcn_readlane((int)lo, __ffsll((long long)top) - 1);     // the common case: no second reduction
    else mlo = wave_min_u32(hi == mhi ? lo : 0xffffffffu);
    mine = hi == mhi && lo == mlo;
    return __longlong_as_double((long long)(((unsigned long long)mhi << 32) | mlo));
}
__device__ __forceinline__ double readlane_f64(double x, int src)
{
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// kGateRegSlots = open-list entries per lane (2: 128 per search, the cheaper pop; 4: 256)
template <int kGateRegSlots>
__global__ __launch_bounds__(64) void gate_reg_kernel(GateWaveArgs a)
{
    extern __shared__ unsigned char gsm[];
    unsigned long long* __restrict__ cache = reinterpret_cast<unsigned long long*>(gsm);      // [blocks][kGateBlockNodes records][8 qwords]
    int32_t* __restrict__ tag = reinterpret_cast<int32_t*>(cache + kGateCacheBlocks * kGateBlockNodes * 8);
    unsigned* __restrict__ closed = reinterpret_cast<unsigned*>(tag + kGateCacheBlocks);
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= a.n_query) return;
    if (!a.run[k] && a.keep_unrun) return;
    if (lane == 0) { a.pre_ok[k] = 0; a.heur_ok[k] = 0; a.dist[k] = -1.; a.redo[k] = 0; }
    if (!a.run[k]) return;
    const uzl_gate_edge c = a.cand[k];
    if (!(c.matching_score >= a.min_score)) return;                                  // :798
    const double* T = c.transform;
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const double diff_rot = fabs(angle_of(R)) * 180 / M_PI;                           // :800-801
    const double tn = sqrt((T[3] * T[3] + T[7] * T[7]) + T[11] * T[11]);
    if (!(tn <= a.max_T && diff_rot <= a.max_R)) return;                             // :803
    if (lane == 0) a.pre_ok[k] = 1;
    if (a.skip_decided && gate_decided(a.poses, c.from, c.to, a.ssf)) { if (lane == 0) { a.heur_ok[k] = 1; a.dist[k] = -2.; } return; }

    const int source = c.from, target = c.to, n = a.n, nwords = (n + 31) / 32;
    unsigned* __restrict__ opened = closed + nwords;
    double* __restrict__ gclosed = a.gclosed + (size_t)k * n;
    for (int i = lane; i < 2 * nwords; i += 64) closed[i] = 0u;
    if (lane < kGateCacheBlocks) tag[lane] = -1;
    __syncthreads();
    // brings the block of 8 records that holds node `u` into the cache (uniform call)
    // a block = kGateBlockNodes records = 2 KB: every lane moves 2 x 16 bytes (the host pads the record array to whole blocks)
    auto fetch_block = [&](int u) {
        const int blk = u >> kGateBlockShift, slot = blk & (kGateCacheBlocks - 1);
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(a.rec) + (size_t)blk * (kGateBlockNodes * 4);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(cache) + slot * (kGateBlockNodes * 4);
        const uint4 q0 = src[lane], q1 = src[lane + 64];
        dst[lane] = q0; dst[lane + 64] = q1;
        if (lane == 0) tag[slot] = blk;
        __builtin_amdgcn_wave_barrier();
    };
    auto rec_of = [&](int u) { return reinterpret_cast<const GateNodeRec*>(cache + ((u >> kGateBlockShift) & (kGateCacheBlocks - 1)) * (kGateBlockNodes * 8) + (u & (kGateBlockNodes - 1)) * 8); };
    auto tag_ok = [&](int u) { return tag[(u >> kGateBlockShift) & (kGateCacheBlocks - 1)] == (u >> kGateBlockShift); };
    fetch_block(target);
    const double tx = rec_of(target)->px, ty = rec_of(target)->py, tz = rec_of(target)->pz;
    fetch_block(source);
    const double kInf = __longlong_as_double(0x7ff0000000000000ll);
    double lw[kGateRegSlots], lg[kGateRegSlots]; int lv[kGateRegSlots];
#pragma unroll
    for (int j = 0; j < kGateRegSlots; j++) { lw[j] = kInf; lg[j] = 0.; lv[j] = -1; }
    int n_list = 1, n_open = 1;
    {
        const GateNodeRec* sr = rec_of(source);
        const double h0 = rec_dist(sr->px, sr->py, sr->pz, tx, ty, tz);
        if (lane == 0) { lw[0] = h0; lv[0] = source; lg[0] = 0.; opened[source >> 5] |= 1u << (source & 31); }
    }
    bool success = false, over = false;
    double g_target = 0.;
    long long dbg_steps = 0, dbg_max = 0, dbg_sec[4] = {0, 0, 0, 0}, dbg_t = 0;
    const long long dbg_c0 = a.dbg ? clock64() : 0, dbg_w0 = a.dbg ? wall_clock64() : 0;
    while (n_open > 0 && n_list > 0) {
        if (a.dbg) { dbg_steps++; dbg_max = n_list > dbg_max ? n_list : dbg_max; dbg_t = clock64(); }
        // ---- pop: lane-local minimum over (h, node, g), then the wave's
        double bw = lw[0], bg = lg[0]; int bv = lv[0], bk = 0;
#pragma unroll
        for (int j = 1; j < kGateRegSlots; j++)
            if (lw[j] < bw || (lw[j] == bw && (lv[j] < bv || (lv[j] == bv && lg[j] < bg)))) { bw = lw[j]; bv = lv[j]; bg = lg[j]; bk = j; }
        bool mine;
        wave_min_pos_f64(bw, mine);
        unsigned long long tie = __ballot(mine);
        if (__popcll(tie) != 1) {                          // equal h in several lanes (duplicates of a node, or equal distances): node id, then g
            int mv = mine ? bv : 0x7fffffff;
#pragma unroll
            for (int o = 32; o; o >>= 1) { const int t = __shfl_xor(mv, o); mv = t < mv ? t : mv; }
            mine = mine && bv == mv;
            bool gm;
            wave_min_pos_f64(mine ? bg : kInf, gm);
            tie = __ballot(mine && gm);
        }
        const int src = __ffsll((long long)tie) - 1;
        const int v = __builtin_amdgcn_readlane(bv, src);
        const double gpop = readlane_f64(bg, src);
        if (a.dbg) { const long long t = clock64(); dbg_sec[0] += t - dbg_t; dbg_t = t; }
        if (v == target) { success = true; g_target = gpop; break; }
        if (lane == src) {
#pragma unroll
            for (int j = 0; j < kGateRegSlots; j++) if (j == bk) lw[j] = kInf;
        }
        n_list--;
        // ---- everything addressed by v: one LDS round trip
        const GateNodeRec* vrp = rec_of(v);
        bool have = tag_ok(v);
        const unsigned cw = closed[v >> 5];
        if (!have) fetch_block(v);
        const double vx = vrp->px, vy = vrp->py, vz = vrp->pz;
        const int degf = vrp->deg, adj = vrp->adj;
        const int deg = degf & ~kGateRecMulti;
        const bool multi = (degf & kGateRecMulti) != 0;      // some neighbour is listed twice (multi-edge): host-side flag
        int u = -1;
        if (lane < deg) u = lane < kGateRecNbr ? vrp->nbr[lane & (kGateRecNbr - 1)] : a.adj_nbr[adj + lane];
        const bool was_closed = (cw >> (v & 31)) & 1u;
        double gv = gpop;
        if (was_closed) {                                  // a stale entry of a node closed earlier: the reference expands it again with its final g
            gv = readlane_f64(lane == 0 ? gclosed[v] : 0., 0);          // (lane 0 wrote it: a thread reads its own stores)
        } else {
            n_open--;
            if (lane == 0) { closed[v >> 5] = cw | (1u << (v & 31)); opened[v >> 5] &= ~(1u << (v & 31)); gclosed[v] = gv; }
        }
        if (a.dbg) { const long long t = clock64(); dbg_sec[1] += t - dbg_t; dbg_t = t; }
        for (int base = 0; base < deg; base += 64) {
            if (base > 0) { const int q = base + lane; u = q < deg ? a.adj_nbr[adj + q] : -1; }
            const bool has = u >= 0;
            // a neighbour listed twice (multi-edge): only its first occurrence acts, as in the sequential loop
            bool first = has;
            if (multi) {
                const int lim = deg - base < 64 ? deg - base : 64;
                for (int j = 0; j < lim; j++) { const int uj = __shfl(u, j); if (j < lane && uj == u) first = false; }
            }
            // ---- everything addressed by u: one LDS round trip (closed / open words, tag, position)
            const int us = has ? u : 0;
            const unsigned cu = closed[us >> 5], ou = opened[us >> 5];
            bool got = tag_ok(us);
            const GateNodeRec* urp = rec_of(us);
            double ux = urp->px, uy = urp->py, uz = urp->pz;
            const bool act = first && u != v && !((cu >> (us & 31)) & 1u);
            got = got || !act;
            while (true) {                                 // blocks not in the cache: loaded one by one (a lane keeps what it read)
                const unsigned long long miss = __ballot(!got);
                if (!miss) break;
                const int um = __builtin_amdgcn_readlane(us, __ffsll((long long)miss) - 1);
                fetch_block(um);
                if (!got && (us >> kGateBlockShift) == (um >> kGateBlockShift)) { ux = urp->px; uy = urp->py; uz = urp->pz; got = true; }
            }
            const bool is_open = act && ((ou >> (us & 31)) & 1u);
            // an open neighbour is pushed again only with a smaller g: its current g = the smallest g among its entries
            double gu = kInf;
            unsigned long long need = __ballot(is_open);
            while (need) {
                const int s0 = __ffsll((long long)need) - 1;
                need &= need - 1;
                const int u0 = __builtin_amdgcn_readlane(us, s0);
                double m = kInf;
#pragma unroll
                for (int j = 0; j < kGateRegSlots; j++) if (lv[j] == u0 && lw[j] < kInf && lg[j] < m) m = lg[j];
                bool dummy;
                m = wave_min_pos_f64(m, dummy);
                if (lane == s0) gu = m;
            }
            double tent = 0., hw = 0.;
            bool push = false;
            if (act) {
                tent = gv + rec_dist(vx, vy, vz, ux, uy, uz);
                if (!is_open || tent < gu) { push = true; hw = rec_dist(ux, uy, uz, tx, ty, tz); }
            }
            if (a.dbg) { const long long t = clock64(); dbg_sec[2] += t - dbg_t; dbg_t = t; }
            if (push) atomicOr(&opened[us >> 5], 1u << (us & 31));
            n_open += __popcll(__ballot(push && !is_open));
            // ---- pushes: one by one into the first lane with a free slot
            unsigned long long pm = __ballot(push);
            while (pm) {
                const int s0 = __ffsll((long long)pm) - 1;
                pm &= pm - 1;
                const double pw = readlane_f64(hw, s0), pg = readlane_f64(tent, s0);
                const int pu = __builtin_amdgcn_readlane(us, s0);
                bool fr = false;
#pragma unroll
                for (int j = 0; j < kGateRegSlots; j++) fr = fr || !(lw[j] < kInf);
                const unsigned long long fm = __ballot(fr);
                if (!fm) { over = true; break; }
                if (lane == __ffsll((long long)fm) - 1) {
                    bool done = false;
#pragma unroll
                    for (int j = 0; j < kGateRegSlots; j++) if (!done && !(lw[j] < kInf)) { lw[j] = pw; lv[j] = pu; lg[j] = pg; done = true; }
                }
                n_list++;
            }
            if (over) break;
        }
        if (a.dbg) { const long long t = clock64(); dbg_sec[3] += t - dbg_t; dbg_t = t; }
        if (over) break;
    }
    if (a.dbg && lane == 0) { a.dbg[8 * k] = dbg_steps; a.dbg[8 * k + 1] = clock64() - dbg_c0; a.dbg[8 * k + 2] = wall_clock64() - dbg_w0; a.dbg[8 * k + 3] = dbg_max;
                              for (int q = 0; q < 4; q++) a.dbg[8 * k + 4 + q] = dbg_sec[q]; }
    if (over) { if (lane == 0) a.redo[k] = 1; return; }
    if (lane != 0) return;
    const double dist = success ? g_target : DBL_MAX;
    a.dist[k] = dist;
    // ---- checkEdgeHeuristic (:1064-1085)
    bool ok = true;
    if (dist != DBL_MAX) {
        double dn, drot;
        pose_gap(a.poses, source, target, dn, drot);
        ok = (2 * a.ssf * dist + 1.0 > dn) && (10 * a.ssf * dist + 30.0 > drot);     // :1074-1075
    }
    a.heur_ok[k] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------
// A search that only has to DECIDE (uzl_gate_check without astar_dist).  checkEdgeHeuristic's verdict is `true` when the target is not
// reachable and otherwise when two tests that are monotone in the found path length `dist` hold; and dist - the length, summed from the
// source, of SOME path over the valid edges - is at least the shortest-path distance d*.  More precisely, with the same edge lengths
// (rec_dist) and floating-point sums that are monotone in their first operand: while Dijkstra's search from the source has not
// closed the target, the smallest key m in its open list is a lower bound of the length of every path to the target as the
// reference's search would sum it (walk the path from the source to its first node Dijkstra has not closed: that node is in the list
// with a key no larger than the path's sum up to there).  So: Dijkstra from the source, and
//   * the list runs empty                        -> the target is not reachable: verdict true;
//   * the popped key m passes both tests         -> whatever dist the reference's search finds passes them too: verdict true;
//   * the target is popped first (m = d*)        -> undecided (the greedy path may still be long enough): the real search runs.
// The ball this search covers has a radius of metres (config 5: at most 10 m of path, against greedy searches that wander over
// hundreds): a few hundred expansions instead of thousands.  Open list in registers (key = g; duplicates instead of decrease-key,
// stale ones skipped at the pop), closed bitmap and record cache in LDS as in gate_reg_kernel.
__global__ __launch_bounds__(64) void gate_bound_kernel(GateWaveArgs a)
{
    constexpr int kSlots = 4;
    extern __shared__ unsigned char gsm[];
    unsigned long long* __restrict__ cache = reinterpret_cast<unsigned long long*>(gsm);
    int32_t* __restrict__ tag = reinterpret_cast<int32_t*>(cache + kGateCacheBlocks * kGateBlockNodes * 8);
    unsigned* __restrict__ closed = reinterpret_cast<unsigned*>(tag + kGateCacheBlocks);
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= a.n_query) return;
    if (lane == 0) { a.pre_ok[k] = 0; a.heur_ok[k] = 0; a.dist[k] = -1.; a.redo[k] = 0; }
    if (!a.run[k]) return;
    const uzl_gate_edge c = a.cand[k];
    if (!(c.matching_score >= a.min_score)) return;                                  // :798
    const double* T = c.transform;
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const double diff_rot = fabs(angle_of(R)) * 180 / M_PI;                           // :800-801
    const double tn = sqrt((T[3] * T[3] + T[7] * T[7]) + T[11] * T[11]);
    if (!(tn <= a.max_T && diff_rot <= a.max_R)) return;                             // :803
    if (lane == 0) a.pre_ok[k] = 1;
    const int source = c.from, target = c.to, n = a.n, nwords = (n + 31) / 32;
    if (!(a.ssf >= 0.)) { if (lane == 0) a.redo[k] = 2; return; }                     // the tests are not monotone in dist: the real search
    double dn, drot;
    pose_gap(a.poses, source, target, dn, drot);
    auto passes = [&](double d) { return (2 * a.ssf * d + 1.0 > dn) && (10 * a.ssf * d + 30.0 > drot); };
    if (gate_decided(a.poses, source, target, a.ssf)) { if (lane == 0) { a.heur_ok[k] = 1; a.dist[k] = -2.; } return; }

    for (int i = lane; i < nwords; i += 64) closed[i] = 0u;
    if (lane < kGateCacheBlocks) tag[lane] = -1;
    __syncthreads();
    auto fetch_block = [&](int u) {
        const int blk = u >> kGateBlockShift, slot = blk & (kGateCacheBlocks - 1);
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(a.rec) + (size_t)blk * (kGateBlockNodes * 4);
        uint4* __restrict__ dst = reinterpret_cast<uint4*>(cache) + slot * (kGateBlockNodes * 4);
        const uint4 q0 = src[lane], q1 = src[lane + 64];
        dst[lane] = q0; dst[lane + 64] = q1;
        if (lane == 0) tag[slot] = blk;
        __builtin_amdgcn_wave_barrier();
    };
    auto rec_of = [&](int u) { return reinterpret_cast<const GateNodeRec*>(cache + ((u >> kGateBlockShift) & (kGateCacheBlocks - 1)) * (kGateBlockNodes * 8) + (u & (kGateBlockNodes - 1)) * 8); };
    auto tag_ok = [&](int u) { return tag[(u >> kGateBlockShift) & (kGateCacheBlocks - 1)] == (u >> kGateBlockShift); };
    const double kInf = __longlong_as_double(0x7ff0000000000000ll);
    double lw[kSlots]; int lv[kSlots];
#pragma unroll
    for (int j = 0; j < kSlots; j++) { lw[j] = kInf; lv[j] = -1; }
    if (lane == 0) { lw[0] = 0.; lv[0] = source; }
    int n_list = 1;
    int verdict = 1;                                       // 1: decided true, 2: undecided
    while (n_list > 0) {
        double bw = lw[0]; int bv = lv[0], bk = 0;
#pragma unroll
        for (int j = 1; j < kSlots; j++) if (lw[j] < bw) { bw = lw[j]; bv = lv[j]; bk = j; }
        bool mine;
        const double m = wave_min_pos_f64(bw, mine);
        const int src = __ffsll((long long)__ballot(mine)) - 1;
        const int v = __builtin_amdgcn_readlane(bv, src);
        if (passes(m)) break;                              // every path to the target is at least m long
        if (v == target) { verdict = 2; break; }           // d* itself does not pass: only the real search can tell
        if (lane == src) {
#pragma unroll
            for (int j = 0; j < kSlots; j++) if (j == bk) lw[j] = kInf;
        }
        n_list--;
        const unsigned cw = closed[v >> 5];
        if ((cw >> (v & 31)) & 1u) continue;               // a stale duplicate
        if (!tag_ok(v)) fetch_block(v);
        const GateNodeRec* vrp = rec_of(v);
        const double vx = vrp->px, vy = vrp->py, vz = vrp->pz;
        const int deg = vrp->deg & ~kGateRecMulti, adj = vrp->adj;
        if (lane == 0) closed[v >> 5] = cw | (1u << (v & 31));
        bool over = false;
        for (int base = 0; base < deg; base += 64) {
            const int q = base + lane;
            int u = -1;
            if (q < deg) u = q < kGateRecNbr ? vrp->nbr[q & (kGateRecNbr - 1)] : a.adj_nbr[adj + q];
            const int us = u >= 0 ? u : 0;
            const unsigned cu = closed[us >> 5];
            bool got = tag_ok(us);
            const GateNodeRec* urp = rec_of(us);
            double ux = urp->px, uy = urp->py, uz = urp->pz;
            const bool act = u >= 0 && u != v && !((cu >> (us & 31)) & 1u);
            got = got || !act;
            while (true) {
                const unsigned long long miss = __ballot(!got);
                if (!miss) break;
                const int um = __builtin_amdgcn_readlane(us, __ffsll((long long)miss) - 1);
                fetch_block(um);
                if (!got && (us >> kGateBlockShift) == (um >> kGateBlockShift)) { ux = urp->px; uy = urp->py; uz = urp->pz; got = true; }
            }
            const double tent = act ? m + rec_dist(vx, vy, vz, ux, uy, uz) : kInf;
            unsigned long long pm = __ballot(act);
            while (pm) {
                const int s0 = __ffsll((long long)pm) - 1;
                pm &= pm - 1;
                const double pw = readlane_f64(tent, s0);
                const int pu = __builtin_amdgcn_readlane(us, s0);
                bool fr = false;
#pragma unroll
                for (int j = 0; j < kSlots; j++) fr = fr || !(lw[j] < kInf);
                const unsigned long long fm = __ballot(fr);
                if (!fm) { over = true; break; }
                if (lane == __ffsll((long long)fm) - 1) {
                    bool done = false;
#pragma unroll
                    for (int j = 0; j < kSlots; j++) if (!done && !(lw[j] < kInf)) { lw[j] = pw; lv[j] = pu; done = true; }
                }
                n_list++;
            }
            if (over) break;
        }
        if (over) { verdict = 2; break; }                  // the list outgrew the registers: the real search
    }
    if (lane != 0) return;
    if (verdict == 1) { a.heur_ok[k] = 1; a.dist[k] = -2.; }
    else a.redo[k] = 2;
}

bool launch_gate_bound(const GateWaveArgs& a, hipStream_t s)
{
    if (a.n_query <= 0) return true;
    const int bytes = gate_lds_bytes(a.n);
    if (bytes > kGateLdsMax) return false;
    static int configured = 0;
    static std::mutex mu;
    if (bytes > 48 * 1024) {
        std::lock_guard<std::mutex> lock(mu);
        if (!configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gate_bound_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kGateLdsMax) != hipSuccess) return false;
            configured = 1;
        }
    }
    hipLaunchKernelGGL(gate_bound_kernel, dim3(a.n_query), dim3(64), bytes, s, a);
    return true;
}

// false: the graph is too large for the LDS bitmaps (the caller uses gate_wave_kernel)
bool launch_gate_reg(const GateWaveArgs& a, int slots, hipStream_t s)
{
    if (a.n_query <= 0) return true;
    const int bytes = gate_lds_bytes(a.n);
    if (bytes > kGateLdsMax) return false;
    static int configured = 0;                             // process-wide: the kernels are symbols of the code object
    static std::mutex mu;
    if (bytes > 48 * 1024) {
        std::lock_guard<std::mutex> lock(mu);
        if (!configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gate_reg_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kGateLdsMax) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(gate_reg_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, kGateLdsMax) != hipSuccess) return false;
            configured = 1;
        }
    }
    if (slots <= 2) hipLaunchKernelGGL(gate_reg_kernel<2>, dim3(a.n_query), dim3(64), bytes, s, a);
    else hipLaunchKernelGGL(gate_reg_kernel<4>, dim3(a.n_query), dim3(64), bytes, s, a);
    return true;
}

void launch_gate_wave(const GateWaveArgs& a, hipStream_t s)
{
    if (a.n_query <= 0) return;
    hipLaunchKernelGGL(gate_wave_kernel, dim3(a.n_query), dim3(64), 0, s, a);
}

void launch_gate(const GateArgs& a, hipStream_t s)
{
    if (a.n_query <= 0) return;
    hipLaunchKernelGGL(gate_kernel, dim3((a.n_query + kGateBlock - 1) / kGateBlock), dim3(kGateBlock), 0, s, a);
}

}  // namespace uzl
