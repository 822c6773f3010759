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
        const double* A = a.poses + 12 * (size_t)source;
        const double* B = a.poses + 12 * (size_t)target;
        double Rd[9], ti[3], td[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) Rd[r * 3 + cc] = (A[0 * 4 + r] * B[0 * 4 + cc] + A[1 * 4 + r] * B[1 * 4 + cc]) + A[2 * 4 + r] * B[2 * 4 + cc];
            ti[r] = -((A[0 * 4 + r] * A[3] + A[1 * 4 + r] * A[7]) + A[2 * 4 + r] * A[11]);
        }
#pragma unroll
        for (int r = 0; r < 3; r++) td[r] = ((A[0 * 4 + r] * B[3] + A[1 * 4 + r] * B[7]) + A[2 * 4 + r] * B[11]) + ti[r];
        const double dn = sqrt((td[0] * td[0] + td[1] * td[1]) + td[2] * td[2]);
        const double drot = 180. * angle_of(Rd) / M_PI;
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
        for (int base = 0; base < vr.deg; base += 64) {
            const int q = base + lane;
            const bool has = q < vr.deg;
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
            for (int j = 0; j < 64 && base + j < vr.deg; j++) {
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
        const double* A = a.poses + 12 * (size_t)source;
        const double* B = a.poses + 12 * (size_t)target;
        double Rd[9], ti[3], td[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) Rd[r * 3 + cc] = (A[0 * 4 + r] * B[0 * 4 + cc] + A[1 * 4 + r] * B[1 * 4 + cc]) + A[2 * 4 + r] * B[2 * 4 + cc];
            ti[r] = -((A[0 * 4 + r] * A[3] + A[1 * 4 + r] * A[7]) + A[2 * 4 + r] * A[11]);
        }
#pragma unroll
        for (int r = 0; r < 3; r++) td[r] = ((A[0 * 4 + r] * B[3] + A[1 * 4 + r] * B[7]) + A[2 * 4 + r] * B[11]) + ti[r];
        const double dn = sqrt((td[0] * td[0] + td[1] * td[1]) + td[2] * td[2]);
        const double drot = 180. * angle_of(Rd) / M_PI;
        ok = (2 * a.ssf * dist + 1.0 > dn) && (10 * a.ssf * dist + 30.0 > drot);     // :1074-1075
    }
    a.heur_ok[k] = ok ? 1 : 0;
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
