// pgo_schur_kernels.hip — Schur reduction of chain interiors: host-side plan + the three kernels (pgo_schur.hpp has the recurrences).
//
//   schur_eliminate_kernel : two waves per run, block-tridiagonal elimination of (H + lambda I) from both ends of the run towards its middle
//                            vertex.  Lane (r, c) of a wave's first 36 owns element (r, c) of every 6x6 block; operands of a product are
//                            staged in LDS (broadcast reads); the 6x6 inverses by the 36 lanes together.  Writes u | W | T per eliminated
//                            vertex and S_L | g_L | S_R | g_R | F per run.
//   schur_assemble_kernel  : the reduced system: off-diagonal blocks (copies of the separator-separator blocks + the runs' fill
//                            blocks), diagonal blocks and right-hand side (own + contributions of the incident runs, in run order:
//                            a gather, no atomics, bit-reproducible).
//   schur_backsub_kernel   : x of the separators copied to their full-system rows; per run the middle vertex, then two waves outwards.
// HBM-bound in principle (algorithmic bytes per eliminated vertex: 3 blocks of 288 B in, 78 doubles out and in again), latency-bound
// in practice: a run is a chain of <= cap dependent 6x6 steps.
#include <algorithm>
#include <cstring>

#include "uzl_common.hpp"
#include "pgo_device.hpp"
#include "pgo_schur.hpp"

namespace uzl {

// ------------------------------------------------------------------------------------------------------------------ host: the plan
namespace {
struct WEdge { int32_t a, c; double w; };
// sorted by (a, c), parallel edges summed, self loops dropped
void merge_edges(std::vector<WEdge>& E)
{
    std::vector<std::pair<uint64_t, double>> K; K.reserve(E.size());
    for (const WEdge& e : E) if (e.a != e.c) K.push_back({((uint64_t)(uint32_t)e.a << 32) | (uint32_t)e.c, e.w});
    std::sort(K.begin(), K.end(), [](const std::pair<uint64_t, double>& x, const std::pair<uint64_t, double>& y) { return x.first < y.first; });
    E.clear();
    for (size_t k = 0; k < K.size(); k++) {
        if (!E.empty() && k > 0 && K[k].first == K[k - 1].first) E.back().w += K[k].second;
        else E.push_back({(int32_t)(K[k].first >> 32), (int32_t)(K[k].first & 0xffffffffu), K[k].second});
    }
}
// Size-capped agglomeration along STRONG edges: per round a heaviest-edge matching of the current groups, an edge counting only if it is at
// least theta x the heaviest edge at either end and the two groups together stay within `cap`; rounds until nothing merges.  Returns the
// group of every node (groups numbered by their lowest node) and leaves the contracted graph in E.  Deterministic (ties: lower indices).
std::vector<int32_t> strong_groups(int n, std::vector<WEdge>& E, int cap, double theta, int max_rounds, int* n_groups)
{
    std::vector<int32_t> grp((size_t)n), size((size_t)n, 1);
    for (int v = 0; v < n; v++) grp[v] = v;
    int cur = n;
    merge_edges(E);
    std::vector<std::pair<uint64_t, uint32_t>> ord;                       // (weight, heaviest first; index): weights are positive doubles, whose bits order like integers
    for (int round = 0; round < max_rounds && cur > 1; round++) {
        std::vector<int32_t> mate((size_t)cur, -1), nid((size_t)cur, -1);
        std::vector<double> wmax((size_t)cur, 0.);
        ord.clear();
        for (size_t k = 0; k < E.size(); k++) {
            wmax[E[k].a] = std::max(wmax[E[k].a], E[k].w); wmax[E[k].c] = std::max(wmax[E[k].c], E[k].w);
            uint64_t bits; const double w = E[k].w > 0. ? E[k].w : 0.; memcpy(&bits, &w, 8);
            ord.push_back({~bits, (uint32_t)k});
        }
        std::sort(ord.begin(), ord.end());
        bool any = false;
        for (const auto& o : ord) {
            const WEdge& e = E[o.second];
            if (mate[e.a] >= 0 || mate[e.c] >= 0 || size[e.a] + size[e.c] > cap) continue;
            if (e.w < theta * std::max(wmax[e.a], wmax[e.c])) continue;
            mate[e.a] = e.c; mate[e.c] = e.a; any = true;
        }
        if (!any) break;
        int cnt = 0;
        for (int v = 0; v < cur; v++) {
            if (nid[v] >= 0) continue;
            nid[v] = cnt; if (mate[v] >= 0) nid[mate[v]] = cnt;
            cnt++;
        }
        std::vector<int32_t> nsize((size_t)cnt, 0);
        for (int v = 0; v < cur; v++) nsize[nid[v]] += size[v];
        size.swap(nsize);
        for (WEdge& e : E) { const int32_t x = nid[e.a], y = nid[e.c]; e.a = std::min(x, y); e.c = std::max(x, y); }
        merge_edges(E);
        for (int v = 0; v < n; v++) grp[v] = nid[grp[v]];
        cur = cnt;
    }
    *n_groups = cur;
    return grp;
}
}  // namespace

// `slot_w` (optional): a stiffness per slot of the block-CSR (trace of the edge's information matrix).  With it, and at least
// `strong_min` separators, the reduced system is numbered by STRONG AGGREGATES instead of in row order (pgo_schur.hpp): groups of <= 8
// separators that hang together by edges at least `theta` x as stiff as the stiffest at either end, four such groups - again the
// strongly coupled ones - to a block of 32 rows, every group padded to 8 rows and every block to 4 groups with EMPTY rows
// (sep_rows = -1: identity diagonal block, zero right-hand side, no off-diagonal blocks).  A removed run counts as springs in series.
SchurPlan schur_plan(int nb, const std::vector<int32_t>& row_ptr, const std::vector<int32_t>& col, int cap, const double* slot_w, int strong_min,
                     double theta, double max_contiguous, int one_level_max, int min_interiors)
{
    SchurPlan P;
    P.nb = nb;
    if (cap < 1) cap = 1;
    if (cap > 64) cap = 64;                              // (schur_eliminate_kernel keeps a run's rows in the lanes of one wave)
    auto deg = [&](int a) { return row_ptr[a + 1] - row_ptr[a]; };
    // a chain interior: one or two incident edges, to different neighbours (a double edge makes both ends separators)
    std::vector<uint8_t> cand((size_t)std::max(nb, 1), 0);
    for (int a = 0; a < nb; a++) {
        const int d = deg(a);
        if (d == 1) cand[a] = 1;
        else if (d == 2) { const int c0 = col[row_ptr[a]], c1 = col[row_ptr[a] + 1]; cand[a] = !(c0 >= 0 && c0 == c1); }
    }
    std::vector<uint8_t> is_int(cand), seen((size_t)std::max(nb, 1), 0);
    // the slot that leads on from `a` when it was entered through in_slot (-1: a leaf entered from its missing side)
    auto way_on = [&](int a, int in_slot) {
        if (deg(a) == 1) return in_slot < 0 ? row_ptr[a] : -1;
        return in_slot == row_ptr[a] ? row_ptr[a] + 1 : row_ptr[a];
    };
    auto slot_to = [&](int a, int c) { for (int s = row_ptr[a]; s < row_ptr[a + 1]; s++) if (col[s] == c) return s; return -1; };
    // pass 1: every (cap+1)-th vertex of a long chain is promoted to a separator, so that no run is longer than cap
    auto walk_promote = [&](int start, int in_slot) {
        int cur = start, len = 0;
        while (cur >= 0 && cand[cur] && !seen[cur]) {
            seen[cur] = 1;
            if (len == cap) { is_int[cur] = 0; len = 0; } else len++;
            const int out = way_on(cur, in_slot);
            if (out < 0) break;
            const int nx = col[out];
            if (nx < 0) break;
            in_slot = slot_to(nx, cur);
            cur = nx;
        }
    };
    auto outer_slot = [&](int a, const std::vector<uint8_t>& inner) {      // slot of `a` that leads out of the chain (-1: a leaf's missing side), or -2 if none does
        if (deg(a) == 1) return -1;
        for (int s = row_ptr[a]; s < row_ptr[a + 1]; s++) { const int c = col[s]; if (c < 0 || !inner[c]) return s; }
        return -2;
    };
    for (int a = 0; a < nb; a++) {
        if (!cand[a] || seen[a]) continue;
        const int o = outer_slot(a, cand);
        if (o == -2) continue;
        // enter from the outer side: a leaf is entered "from nowhere", i.e. its one slot is the way on
        walk_promote(a, o);
    }
    for (int a = 0; a < nb; a++) {                       // cycles of candidates only (no separator, no fixed vertex on them): break them
        if (!cand[a] || seen[a]) continue;
        seen[a] = 1; is_int[a] = 0;
        const int nx = col[row_ptr[a]];
        if (nx >= 0) walk_promote(nx, slot_to(nx, a));
    }
    // reduced numbering: row order first (the runs are found with it) ...
    P.full2red.assign((size_t)std::max(nb, 1), -1);
    for (int a = 0; a < nb; a++) if (!is_int[a]) { P.full2red[a] = (int32_t)P.sep_rows.size(); P.sep_rows.push_back(a); }
    P.n_sep = (int32_t)P.sep_rows.size();
    P.nbr = P.n_sep;
    P.n_int = nb - P.n_sep;
    if (P.n_int < min_interiors) return P;                  // the caller will not reduce: nothing below is looked at
    // pass 2: the runs
    std::vector<uint8_t> in_run((size_t)std::max(nb, 1), 0);
    P.run_ptr.push_back(0);
    auto red_of = [&](int slot) { return (slot >= 0 && col[slot] >= 0) ? P.full2red[col[slot]] : -1; };
    for (int a = 0; a < nb; a++) {
        if (!is_int[a] || in_run[a]) continue;
        const int o = outer_slot(a, is_int);
        if (o == -2) continue;                           // not an end of its run: reached from the end with the lower row
        int cur = a, in_slot = o;                        // in_slot = -1: a leaf's missing side
        P.endL.push_back(red_of(in_slot));
        int len = 0;
        while (true) {
            in_run[cur] = 1; len++;
            P.run_rows.push_back(cur);
            P.slotP.push_back((in_slot >= 0 && col[in_slot] >= 0) ? in_slot : -1);
            const int out = way_on(cur, in_slot);
            P.slotN.push_back((out >= 0 && col[out] >= 0) ? out : -1);
            const int nx = out >= 0 ? col[out] : -1;
            if (nx < 0 || !is_int[nx] || in_run[nx]) { P.endR.push_back((nx >= 0 && !is_int[nx]) ? P.full2red[nx] : -1); break; }
            in_slot = slot_to(nx, cur);
            cur = nx;
        }
        P.longest_run = std::max(P.longest_run, len);
        P.run_ptr.push_back((int32_t)P.run_rows.size());
    }
    P.n_runs = (int32_t)P.endL.size();
    // ... then, for a system large enough to gain from it, by strong aggregates
    if (slot_w && strong_min > 0 && P.n_sep >= strong_min) {
        std::vector<WEdge> E;
        for (int i = 0; i < P.n_sep; i++) {
            const int a = P.sep_rows[i];
            for (int s = row_ptr[a]; s < row_ptr[a + 1]; s++) {
                const int c = col[s];
                if (c >= 0 && !is_int[c] && P.full2red[c] > i) E.push_back({i, P.full2red[c], slot_w[s]});      // every edge has a slot at both ends: taken at the lower
            }
        }
        for (int r = 0; r < P.n_runs; r++) {
            const int L = P.endL[r], R = P.endR[r];
            if (L < 0 || R < 0 || L == R) continue;
            double inv = 0.;
            const int p0 = P.run_ptr[r], p1 = P.run_ptr[r + 1];
            if (P.slotP[p0] >= 0) inv += 1. / std::max(slot_w[P.slotP[p0]], 1e-300);
            for (int q = p0; q < p1; q++) if (P.slotN[q] >= 0) inv += 1. / std::max(slot_w[P.slotN[q]], 1e-300);
            E.push_back({std::min(L, R), std::max(L, R), inv > 0. ? 1. / inv : 0.});
        }
        int n1 = 0, n2 = 0;
        const std::vector<int32_t> g1 = strong_groups(P.n_sep, E, kMlFanout, theta, 5, &n1);      // E: now the graph of the groups
        // Few groups: every group one aggregate of the level-1 path (one aggregate per workgroup, exact 48 x 48 blocks, the dense operator at
        // level 1: n1 <= one_level_max keeps its n^3 rebuild affordable), blocks of ONE group.  More: blocks of <= 4 strongly tied groups.
        const bool one_level = n1 <= one_level_max;
        std::vector<int32_t> g2;
        if (one_level) { g2.resize((size_t)n1); for (int a1 = 0; a1 < n1; a1++) g2[a1] = a1; n2 = n1; }
        else {
            g2 = strong_groups(n1, E, kMlFanout2, theta, 4, &n2);
            // A/B switch (diagnostic build): blocks that stayed below four groups are packed together in the order of their lowest group,
            // tie or no tie - fewer, fuller blocks (every block is 32 rows of every PCG kernel's work, filled or not).  Measured on config
            // 5 (round 5, tests/diag/online_passes.py): 35 % fewer rows (3328 -> 2176 at the first interval on this layout), 6 % more PCG
            // iterations (57.3 k -> 61.0 k), optimize 1.39 -> 1.36 s: the kernels are latency-bound, the rows were nearly free.  Off.
            static const int pack = diag_int("UZL_SCHUR_BLOCK_PACK", 0);
            if (pack) {
                std::vector<int32_t> bsize((size_t)n2, 0), nid((size_t)n2, -1);
                for (int a1 = 0; a1 < n1; a1++) bsize[g2[a1]]++;
                int cnt = 0, open_id = -1, open_fill = 0;
                for (int b = 0; b < n2; b++) {                                      // blocks are numbered by their lowest group
                    if (bsize[b] >= kMlFanout2) { nid[b] = cnt++; continue; }
                    if (open_id < 0 || open_fill + bsize[b] > kMlFanout2) { open_id = cnt++; open_fill = 0; }
                    nid[b] = open_id; open_fill += bsize[b];
                }
                for (int a1 = 0; a1 < n1; a1++) g2[a1] = nid[g2[a1]];
                n2 = cnt;
            }
            static const bool verbose_plan = diag_flag("UZL_SCHUR_PLAN_DBG");
            if (verbose_plan) fprintf(stderr, "[uzl] schur plan: %d separators, %d groups, %d blocks -> %d rows\n", P.n_sep, n1, n2, n2 * kMlFanout * kMlFanout2);
        }
        // Few groups: every group one aggregate of the level-1 path - never worse than 8 consecutive separators on that path
        // (tests/diag/strong_ab.py: 9.0 -> 7.8 ms at 3000 / 3100, 17.0 -> 14.7 at 12000 / 12700).  More groups need the blocks-of-4 layout,
        // whose path has weaker smoothers and a dearer iteration: do the groups differ from the row order at all?  Where the runs between
        // separators are stiffer than the loop closures the matching follows the chain and most groups are consecutive separators anyway:
        // then the row order with its level-1 path is the better preconditioner (15.6 against 23.1 ms at 8000 / 9000).
        {
            std::vector<int32_t> lo((size_t)n1, P.n_sep), hi((size_t)n1, -1), cnt((size_t)n1, 0);
            for (int i = 0; i < P.n_sep; i++) { lo[g1[i]] = std::min(lo[g1[i]], i); hi[g1[i]] = std::max(hi[g1[i]], i); cnt[g1[i]]++; }
            int64_t in_contig = 0;
            for (int a1 = 0; a1 < n1; a1++) if (hi[a1] - lo[a1] == cnt[a1] - 1) in_contig += cnt[a1];
            P.strong_contiguous = P.n_sep > 0 ? (double)in_contig / P.n_sep : 1.;
        }
        if (n1 <= one_level_max || P.strong_contiguous < max_contiguous) {
        // position of group j of block G = 32 G + 8 j; groups and blocks are numbered by their lowest member: row order survives inside them
        std::vector<int32_t> first1((size_t)n1, -1), slot_in2((size_t)n1, 0), fill2((size_t)n2, 0), fill1((size_t)n1, 0), perm((size_t)P.n_sep);
        for (int a1 = 0; a1 < n1; a1++) slot_in2[a1] = fill2[g2[a1]]++;
        const int rows_per_blk = one_level ? kMlFanout : kMlFanout * kMlFanout2;
        for (int i = 0; i < P.n_sep; i++) { const int a1 = g1[i]; perm[i] = g2[a1] * rows_per_blk + slot_in2[a1] * kMlFanout + fill1[a1]++; }
        P.nbr = n2 * rows_per_blk;
        P.strong = true; P.n_strong1 = n1; P.n_strong2 = one_level ? 0 : n2;
        std::vector<int32_t> sep((size_t)P.nbr, -1);
        for (int i = 0; i < P.n_sep; i++) sep[perm[i]] = P.sep_rows[i];
        P.sep_rows.swap(sep);
        for (int i = 0; i < P.nbr; i++) if (P.sep_rows[i] >= 0) P.full2red[P.sep_rows[i]] = i;
        for (int r = 0; r < P.n_runs; r++) { if (P.endL[r] >= 0) P.endL[r] = perm[P.endL[r]]; if (P.endR[r] >= 0) P.endR[r] = perm[P.endR[r]]; }
        }
    }
    // reduced block-CSR: kept blocks in slot order, then the fill blocks of the incident runs in run order
    std::vector<std::vector<int32_t>> inc((size_t)std::max(P.nbr, 1));
    for (int r = 0; r < P.n_runs; r++) {
        const int L = P.endL[r], R = P.endR[r];
        if (L >= 0 && L == R) inc[L].push_back(4 * r + 2);
        else { if (L >= 0) inc[L].push_back(4 * r + 0); if (R >= 0) inc[R].push_back(4 * r + 1); }
    }
    P.row_ptr.assign((size_t)P.nbr + 1, 0);
    P.inc_ptr.assign((size_t)P.nbr + 1, 0);
    for (int i = 0; i < P.nbr; i++) {
        const int a = P.sep_rows[i];
        if (a >= 0) {
            for (int s = row_ptr[a]; s < row_ptr[a + 1]; s++) {
                const int c = col[s];
                if (c >= 0 && !is_int[c]) { P.col.push_back(P.full2red[c]); P.rsrc.push_back(s); }
            }
            for (int32_t code : inc[i]) {
                const int r = code >> 2, side = code & 3;
                P.inc.push_back(code);
                if (side == 2) continue;                     // both ends here: diagonal only
                const int other = side == 0 ? P.endR[r] : P.endL[r];
                if (other >= 0) { P.col.push_back(other); P.rsrc.push_back(-(2 * r + side) - 1); }
            }
        }
        P.row_ptr[i + 1] = (int32_t)P.col.size();
        P.inc_ptr[i + 1] = (int32_t)P.inc.size();
    }
    P.nslots_r = (int32_t)P.col.size();
    return P;
}

// ------------------------------------------------------------------------------------------------------------------ device
// C(r, c) = sum_k A[r][k] B[k][c] for the lane's (r, c); A, B row-major 6x6 in LDS.  TA / TB: use the transpose.
template <bool TA, bool TB>
__device__ __forceinline__ double mm6(const double* __restrict__ A, const double* __restrict__ B, int r, int c)
{
    double s = 0.;
#pragma unroll
    for (int k = 0; k < 6; k++) s = fma(TA ? A[k * 6 + r] : A[r * 6 + k], TB ? B[c * 6 + k] : B[k * 6 + c], s);
    return s;
}
template <bool TA>
__device__ __forceinline__ double mv6(const double* __restrict__ A, const double* __restrict__ v, int r)
{
    double s = 0.;
#pragma unroll
    for (int k = 0; k < 6; k++) s = fma(TA ? A[k * 6 + r] : A[r * 6 + k], v[k], s);
    return s;
}

// TWO waves per run, eliminating from BOTH ENDS towards the run's middle vertex (round 4: a run is a chain of dependent 6 x 6 steps of
// ~3 us; one wave walking all of it took 67 us at 24 steps).  Wave 0 takes v_0 .. v_{j-1} with the recurrences of pgo_schur.hpp, wave 1
// the mirror image from the other end (v_{k-1} .. v_{j+1}: "left separator" = s1, "next" = the vertex before; E'_q = E_{q-1}^T), then
// wave 0 eliminates the middle vertex v_j, which by then couples to s0 through C_L and to s1 through C_R:
//     Dinv = (H_jj + lambda I + what both halves left)^-1;  u = Dinv g';  W^L = Dinv C_L^T;  W^R = Dinv C_R^T
//     S_L -= C_L W^L;  g_L -= C_L u;  S_R -= C_R W^R;  g_R -= C_R u;  F = -C_L W^R;      x_j = u - W^L x_s0 - W^R x_s1
// Records: u | W | T per vertex - W multiplies the x of the half's own separator, T the x of the vertex eliminated after it (towards the
// middle); the middle keeps W^L | W^R.  LDS matrices are private to a wave; the two waves' step counts differ by at most one and the
// shorter one idles through the barriers of the longer.
struct SchurWaveLds { double D[36], Di[36], C[36], E[36], T[36], W[36], g[6], u[6]; };
// The LDS matrices of a half are private to its wave: what one lane writes the wave's other lanes read a few instructions later.  A wave's
// LDS operations execute in program order, so all that is needed between the write and the reads is that the COMPILER keeps that order -
// no s_barrier.  (Round 5: the elimination's 19 workgroup barriers per step made its two waves march in lock step; without them 37.9 ->
// 34.5 us at 20k / 21.7k - tests/diag/r5_schur_ab.sh; the step is a chain of ~60 LDS round trips and six divisions either way.)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Di = D^-1 for the SPD 6 x 6 in L.D, by the wave's first 36 lanes TOGETHER: in-place Gauss-Jordan, lane (r, c) owns element (r, c), six
// pivot steps through LDS, then the mean with the transpose (the inverse of a symmetric matrix, symmetric to the last bit like the
// Cholesky-based routine's).  Every lane used to invert the same matrix on its own (spd_inverse6_rs: ~600 flops x 64 lanes): with 1285
// runs in flight that made the elimination a throughput problem - half the chain length per wave changed nothing.  Called by all lanes
// of the wave (`on` = this wave has a matrix).
__device__ __forceinline__ void schur_inverse6_coop(SchurWaveLds& L, bool on, bool act, int lane, int r, int c)
{
    double a = (on && act) ? L.D[lane] : 0.;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (on && act) L.Di[lane] = a;
        wave_sync();
        if (on && act) {
            const double p = 1. / L.Di[k * 6 + k], rk = L.Di[r * 6 + k], kc = L.Di[k * 6 + c];
            a = (r == k) ? ((c == k) ? p : kc * p) : ((c == k) ? -rk * p : fma(-rk * p, kc, a));
        }
        wave_sync();
    }
    if (on && act) L.Di[lane] = a;
    wave_sync();
    if (on && act) a = 0.5 * (a + L.Di[c * 6 + r]);
    wave_sync();
    if (on && act) L.Di[lane] = a;
}
__device__ __forceinline__ void schur_eliminate_kernel_body(PgoDev D, SchurDev S)
{
    __shared__ SchurWaveLds sw[2];
    __shared__ double xch[2][36 + 36 + 6 + 36 + 6];          // per half, for the middle step: C | dupd | gupd | acc S | acc g
    const int run = blockIdx.x;
    if (run >= S.n_runs) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane / 6, c = lane % 6;
    const bool act = lane < 36, vec = lane < 6;
    SchurWaveLds& L = sw[wv];
    const double lambda = D.scal[3];
    const int p0 = S.run_ptr[run], p1 = S.run_ptr[run + 1], len = p1 - p0;
    const bool hasL = S.endL[run] >= 0, hasR = S.endR[run] >= 0;
    const int j = len / 2;                                    // the middle vertex; wave 0: q = 0 .. j - 1, wave 1: q = len - 1 .. j + 1
    const int my_steps = wv == 0 ? j : len - 1 - j, steps = max(j, len - 1 - j);
    const bool has_sep = wv == 0 ? hasL : hasR;               // this half's own separator exists
    // the rows and slots of the whole run once (lane q: vertex q; runs are <= 64 long, schur_plan)
    const int vq = lane < len ? S.run_rows[p0 + lane] : 0, sq = lane < len ? S.slotN[p0 + lane] : -1;
    auto eblk = [&](int q, int idx) -> double {               // element idx of E_q = H_{v_q, next} (sharded solve: summed over the ranks)
        return S.runblk ? S.runblk[(size_t)(p0 + q) * 36 + idx] : D.blk[(size_t)__shfl(sq, q) * 36 + idx];
    };
    // coupling of this half's first vertex to its own separator: C = H_{s0,v_0} = (H_{v_0,s0})^T, or H_{s1,v_{k-1}} = E_{k-1}^T
    double cval = 0.;
    if (wv == 0) {
        const double* __restrict__ cblk = S.runblk ? S.runblk + (size_t)(S.n_int + run) * 36 : D.blk + (size_t)(hasL ? S.slotP[p0] : 0) * 36;
        if (hasL && act) cval = cblk[c * 6 + r];
    } else if (hasR) {
        const double e = eblk(len - 1, act ? c * 6 + r : 0);
        if (act) cval = e;
    }
    double dupd = 0., gupd = 0., accS = 0., accg = 0.;
    auto fetch = [&](int it, double& hd, double& ev, double& bv) {          // operands of this half's step `it`
        const int q = wv == 0 ? it : len - 1 - it;
        const int v = __shfl(vq, q);
        hd = act ? D.hdiag[(size_t)v * 36 + lane] : 0.;
        const double e = wv == 0 ? eblk(q, act ? lane : 0) : eblk(q - 1, act ? c * 6 + r : 0);      // wave 1: E'_q = E_{q-1}^T
        ev = act ? e : 0.;
        bv = vec ? D.b[(size_t)v * 6 + lane] : 0.;
    };
    double hd = 0., ev = 0., bv = 0.;
    if (my_steps > 0) fetch(0, hd, ev, bv);
    for (int it = 0; it < steps; it++) {
        const bool on = it < my_steps;                       // (wave-uniform)
        const int q = wv == 0 ? it : len - 1 - it;
        if (on) {
            if (act) { L.D[lane] = hd + ((r == c) ? lambda : 0.) + dupd; L.E[lane] = ev; L.C[lane] = cval; }
            if (vec) L.g[lane] = bv + gupd;
            if (it + 1 < my_steps) fetch(it + 1, hd, ev, bv);
        }
        wave_sync();
        schur_inverse6_coop(L, on, act, lane, r, c);
        wave_sync();
        if (on) {
            double tval = 0., wval = 0., uval = 0.;
            if (act) {
                tval = mm6<false, false>(L.Di, L.E, r, c);                    // T = Dinv E
                if (has_sep) wval = mm6<false, true>(L.Di, L.C, r, c);        // W = Dinv C^T
                L.T[lane] = tval; L.W[lane] = wval;
            }
            if (vec) { uval = mv6<false>(L.Di, L.g, lane); L.u[lane] = uval; }
            double* __restrict__ out = S.elim + (size_t)(p0 + q) * kSchurElim;
            if (vec) out[lane] = uval;
            if (act) { out[6 + lane] = wval; out[42 + lane] = tval; }
        }
        wave_sync();
        if (on) {
            if (act) {
                if (has_sep) { accS -= mm6<false, false>(L.C, L.W, r, c); cval = -mm6<false, false>(L.C, L.T, r, c); }      // S -= C W;  C' = -C T
                dupd = -mm6<true, false>(L.E, L.T, r, c);                      // -E^T T
            }
            if (vec) {
                if (has_sep) accg -= mv6<false>(L.C, L.u, lane);
                gupd = -mv6<true>(L.E, L.u, lane);
            }
        }
        wave_sync();                                               // LDS is rewritten at the top of the next step
    }
    // ---- the middle vertex: both halves hand over what they left on it
    if (act) { xch[wv][lane] = cval; xch[wv][36 + lane] = dupd; xch[wv][78 + lane] = accS; }
    if (vec) { xch[wv][72 + lane] = gupd; xch[wv][114 + lane] = accg; }
    __syncthreads();
    if (wv != 0) return;
    {
        const int v = __shfl(vq, j);
        if (act) {
            L.D[lane] = D.hdiag[(size_t)v * 36 + lane] + ((r == c) ? lambda : 0.) + (xch[0][36 + lane] + xch[1][36 + lane]);
            L.C[lane] = xch[0][lane];                                   // C_L
            L.E[lane] = xch[1][lane];                                   // C_R
        }
        if (vec) L.g[lane] = D.b[(size_t)v * 6 + lane] + (xch[0][72 + lane] + xch[1][72 + lane]);
        wave_sync();                                               // (one wave left in the workgroup: orders its LDS traffic)
        schur_inverse6_coop(L, true, act, lane, r, c);
        wave_sync();
        double wl = 0., wr = 0., uval = 0.;
        if (act) {
            if (hasL) wl = mm6<false, true>(L.Di, L.C, r, c);           // W^L = Dinv C_L^T
            if (hasR) wr = mm6<false, true>(L.Di, L.E, r, c);           // W^R = Dinv C_R^T
            L.W[lane] = wl; L.T[lane] = wr;
        }
        if (vec) { uval = mv6<false>(L.Di, L.g, lane); L.u[lane] = uval; }
        double* __restrict__ out = S.elim + (size_t)(p0 + j) * kSchurElim;
        if (vec) out[lane] = uval;
        if (act) { out[6 + lane] = wl; out[42 + lane] = wr; }
        wave_sync();
        double* __restrict__ ro = S.runout + (size_t)run * kSchurRunOut;
        if (act) {
            ro[lane] = hasL ? xch[0][78 + lane] - mm6<false, false>(L.C, L.W, r, c) : 0.;                 // S_L
            ro[42 + lane] = hasR ? xch[1][78 + lane] - mm6<false, false>(L.E, L.T, r, c) : 0.;            // S_R
            ro[84 + lane] = (hasL && hasR) ? -mm6<false, false>(L.C, L.T, r, c) : 0.;                     // F = -C_L W^R
        }
        if (vec) {
            ro[36 + lane] = hasL ? xch[0][114 + lane] - mv6<false>(L.C, L.u, lane) : 0.;                  // g_L
            ro[78 + lane] = hasR ? xch[1][114 + lane] - mv6<false>(L.E, L.u, lane) : 0.;                  // g_R
        }
    }
}

// Reduced system: item t / 36 = off-diagonal block (copy or fill), then diagonal block, then (6 lanes) right-hand side
__device__ __forceinline__ void schur_assemble_kernel_body(PgoDev D, PgoDev R, SchurDev S)
{
    const int t = blockIdx.x * kBlk + threadIdx.x;
    const int item = t / 36, k = t % 36, kt = (k % 6) * 6 + k / 6;
    if (item < S.nslots_r) {
        const int src = S.rsrc[item];
        double v;
        if (src >= 0) v = D.blk[(size_t)src * 36 + k];
        else {
            const int code = -src - 1, run = code >> 1, side = code & 1;
            const double* __restrict__ F = S.runout + (size_t)run * kSchurRunOut + 84;
            v = side == 0 ? F[k] : F[kt];
            if (!D.diag_owner) v = 0.;                 // sharded solve: every rank eliminated the run - its fill block enters the summed A p once
        }
        R.blk[(size_t)item * 36 + k] = v;
        return;
    }
    const int i = item - S.nslots_r;
    if (i >= S.nbr) return;
    const int a = S.sep_rows[i];
    if (a < 0) {                                        // an empty row of a strong-aggregate numbering: x_i = 0
        R.hdiag[(size_t)i * 36 + k] = (k % 7 == 0) ? 1. : 0.;
        if (k < 6) R.b[(size_t)i * 6 + k] = 0.;
        return;
    }
    double h = D.hdiag[(size_t)a * 36 + k];
    double g = (k < 6) ? D.b[(size_t)a * 6 + k] : 0.;
    for (int q = S.inc_ptr[i]; q < S.inc_ptr[i + 1]; q++) {
        const int code = S.inc[q], run = code >> 2, side = code & 3;
        const double* __restrict__ ro = S.runout + (size_t)run * kSchurRunOut;
        if (side == 0) { h += ro[k]; if (k < 6) g += ro[36 + k]; }
        else if (side == 1) { h += ro[42 + k]; if (k < 6) g += ro[78 + k]; }
        else { h += (ro[k] + ro[42 + k]) + (ro[84 + k] + ro[84 + kt]); if (k < 6) g += ro[36 + k] + ro[78 + k]; }
    }
    R.hdiag[(size_t)i * 36 + k] = h;
    if (k < 6) R.b[(size_t)i * 6 + k] = g;
}

// blocks [0, n_runs): one wave per run, backwards; blocks behind: x of the separators to their full-system rows
__device__ __forceinline__ void schur_backsub_kernel_body(PgoDev D, PgoDev R, SchurDev S)
{
    __shared__ double sxs[2][6], sxn[2][6], sp[2][36];          // per wave: x of its own separator, x of the vertex solved last, products
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= S.n_runs) {
        const int t = ((int)blockIdx.x - S.n_runs) * 64 + lane;
        if (wv == 0 && t < S.nbr * 6 && S.sep_rows[t / 6] >= 0) D.x[(size_t)S.sep_rows[t / 6] * 6 + t % 6] = R.x[t];
        return;
    }
    const int run = blockIdx.x, c = lane % 6;
    const bool act = lane < 36, vec = lane < 6;
    const int p0 = S.run_ptr[run], p1 = S.run_ptr[run + 1], len = p1 - p0, j = len / 2;
    const int eL = S.endL[run], eR = S.endR[run];
    const int my_steps = wv == 0 ? j : len - 1 - j, steps = max(j, len - 1 - j);
    if (vec) { const int es = wv == 0 ? eL : eR; sxs[wv][lane] = es >= 0 ? R.x[(size_t)es * 6 + lane] : 0.; }
    // the operands of this half's first vertex are fetched before the middle's chain link: they do not depend on it
    auto rec = [&](int it) { return S.elim + (size_t)(p0 + (wv == 0 ? j - 1 - it : j + 1 + it)) * kSchurElim; };
    double w = 0., t = 0., u = 0.;
    if (my_steps > 0) { const double* __restrict__ e = rec(0); if (act) { w = e[6 + lane]; t = e[42 + lane]; } if (vec) u = e[lane]; }
    __syncthreads();
    if (wv == 0) {                                            // the middle vertex: x_j = u - W^L x_s0 - W^R x_s1
        const double* __restrict__ e = S.elim + (size_t)(p0 + j) * kSchurElim;
        if (act) sp[0][lane] = fma(e[6 + lane], sxs[0][c], e[42 + lane] * sxs[1][c]);
    }
    __syncthreads();
    if (wv == 0 && vec) {
        const double* __restrict__ q = sp[0] + lane * 6;
        const double x = S.elim[(size_t)(p0 + j) * kSchurElim + lane] - (((q[0] + q[1]) + (q[2] + q[3])) + (q[4] + q[5]));
        D.x[(size_t)S.run_rows[p0 + j] * 6 + lane] = x;
        sxn[0][lane] = x; sxn[1][lane] = x;
    }
    __syncthreads();
    for (int it = 0; it < steps; it++) {                      // outwards from the middle, both halves at once
        const bool on = it < my_steps;
        double wn = 0., tn = 0., un = 0.;
        if (on && it + 1 < my_steps) {
            const double* __restrict__ en = rec(it + 1);
            if (act) { wn = en[6 + lane]; tn = en[42 + lane]; }
            if (vec) un = en[lane];
        }
        if (on && act) sp[wv][lane] = fma(w, sxs[wv][c], t * sxn[wv][c]);
        wave_sync();                                          // (sp, sxn, sxs of a half are its wave's own)
        if (on && vec) {
            const double* __restrict__ q = sp[wv] + lane * 6;
            const double x = u - (((q[0] + q[1]) + (q[2] + q[3])) + (q[4] + q[5]));
            D.x[(size_t)S.run_rows[p0 + (wv == 0 ? j - 1 - it : j + 1 + it)] * 6 + lane] = x;
            sxn[wv][lane] = x;
        }
        wave_sync();
        w = wn; t = tn; u = un;
    }
}

// sharded solve: this rank's share of the runs' chain blocks (zero where another rank linearised the edge) into one contiguous buffer
__global__ __launch_bounds__(kBlk) void schur_gather_kernel(PgoDev D, SchurDev S)
{
    const long t = (long)blockIdx.x * kBlk + threadIdx.x;
    const long item = t / 36; const int k = (int)(t % 36);
    if (item >= S.n_int + S.n_runs) return;
    int slot;
    if (item < S.n_int) slot = S.slotN[item];
    else { const int run = (int)(item - S.n_int); slot = S.endL[run] >= 0 ? S.slotP[S.run_ptr[run]] : -1; }
    S.runblk[item * 36 + k] = slot >= 0 ? D.blk[(size_t)slot * 36 + k] : 0.;
}
__global__ __launch_bounds__(128) void schur_eliminate_kernel(PgoDev D, SchurDev S) { schur_eliminate_kernel_body(D, S); }
__global__ __launch_bounds__(kBlk) void schur_assemble_kernel(PgoDev D, PgoDev R, SchurDev S) { schur_assemble_kernel_body(D, R, S); }
__global__ __launch_bounds__(128) void schur_backsub_kernel(PgoDev D, PgoDev R, SchurDev S) { schur_backsub_kernel_body(D, R, S); }

// slot twins of the device-resident LM loop (pgo_types.hpp): graph = blockIdx.z; the reduction runs in the pass lm_head_kernel stamped
// (a new lambda), the back-substitution with the evaluation of a trial
__global__ __launch_bounds__(128) void schur_eliminate_lm_kernel(const LmSlot* __restrict__ slots)
{
    const LmSlot& S = slots[blockIdx.z];
    if (!S.red || S.lm->schur_pass != S.lm->pass) return;
    schur_eliminate_kernel_body(S.D, S.SD);
}
__global__ __launch_bounds__(kBlk) void schur_assemble_lm_kernel(const LmSlot* __restrict__ slots)
{
    const LmSlot& S = slots[blockIdx.z];
    if (!S.red || S.lm->schur_pass != S.lm->pass) return;
    schur_assemble_kernel_body(S.D, S.Dp, S.SD);
}
__global__ __launch_bounds__(128) void schur_backsub_lm_kernel(const LmSlot* __restrict__ slots)
{
    const LmSlot& S = slots[blockIdx.z];
    const LmDev* lm = S.lm;
    if (!S.red || !(lm->phase == kLmSolve && lm->flags[0] != 0 && lm->flags[2] == 0)) return;
    if ((int)blockIdx.x >= S.SD.n_runs + (S.SD.nbr * 6 + 63) / 64) return;
    schur_backsub_kernel_body(S.D, S.Dp, S.SD);
}

}  // namespace uzl

// The plan uzl_pgo_optimize works with, for a caller-supplied block structure: pure host code (no device needed).
extern "C" int uzl_pgo_schur_plan(int32_t nb, const int32_t* row_ptr, const int32_t* col, int32_t cap, int32_t* red_row, int32_t* run_id,
                                  int32_t* run_pos, int32_t* red_row_ptr, int32_t* red_col, int32_t cap_slots, int32_t* n_reduced,
                                  int32_t* n_runs)
{
    if (nb < 0 || !row_ptr || (nb > 0 && (!red_row || !run_id || !run_pos)) || !red_row_ptr || !n_reduced || !n_runs) return UZL_ERR_BAD_ARG;
    for (int a = 0; a < nb; a++) if (row_ptr[a + 1] < row_ptr[a]) return UZL_ERR_BAD_ARG;
    if (nb > 0 && row_ptr[nb] > 0 && !col) return UZL_ERR_BAD_ARG;
    try {
        const std::vector<int32_t> rp(row_ptr, row_ptr + nb + 1), cl(col, col + (nb > 0 ? row_ptr[nb] : 0));
        for (int32_t c : cl) if (c < -1 || c >= nb) return UZL_ERR_BAD_ARG;
        const uzl::SchurPlan P = uzl::schur_plan(nb, rp, cl, cap, nullptr, 0, 0., 2., 0);
        if (P.nslots_r > cap_slots || (P.nslots_r > 0 && !red_col)) return UZL_ERR_BAD_ARG;
        for (int a = 0; a < nb; a++) { red_row[a] = P.full2red[a]; run_id[a] = -1; run_pos[a] = -1; }
        for (int r = 0; r < P.n_runs; r++)
            for (int q = P.run_ptr[r]; q < P.run_ptr[r + 1]; q++) { run_id[P.run_rows[q]] = r; run_pos[P.run_rows[q]] = q - P.run_ptr[r]; }
        for (int i = 0; i <= P.nbr; i++) red_row_ptr[i] = P.row_ptr[i];
        for (int k = 0; k < P.nslots_r; k++) red_col[k] = P.col[k];
        *n_reduced = P.nbr; *n_runs = P.n_runs;
        return UZL_OK;
    } catch (...) { return UZL_ERR_OOM; }
}

// The same with the strong-aggregate numbering (SchurPlan::strong): red_row = full row -> reduced row, sep_rows = reduced row -> full row
// or -1 for an empty row; counts = {reduced rows, separators, groups of <= 8, blocks of <= 4 groups}.  Pure host code.
extern "C" int uzl_pgo_schur_plan_strong(int32_t nb, const int32_t* row_ptr, const int32_t* col, int32_t cap, const double* slot_w, int32_t strong_min,
                                         double theta, int32_t one_level_max, int32_t* red_row, int32_t* sep_rows, int32_t cap_rows, int32_t* counts)
{
    if (nb < 0 || !row_ptr || !slot_w || (nb > 0 && !red_row) || !sep_rows || !counts) return UZL_ERR_BAD_ARG;
    for (int a = 0; a < nb; a++) if (row_ptr[a + 1] < row_ptr[a]) return UZL_ERR_BAD_ARG;
    if (nb > 0 && row_ptr[nb] > 0 && !col) return UZL_ERR_BAD_ARG;
    try {
        const std::vector<int32_t> rp(row_ptr, row_ptr + nb + 1), cl(col, col + (nb > 0 ? row_ptr[nb] : 0));
        for (int32_t c : cl) if (c < -1 || c >= nb) return UZL_ERR_BAD_ARG;
        const uzl::SchurPlan P = uzl::schur_plan(nb, rp, cl, cap, slot_w, strong_min, theta, 2., one_level_max);
        if (P.nbr > cap_rows) return UZL_ERR_BAD_ARG;
        for (int a = 0; a < nb; a++) red_row[a] = P.full2red[a];
        for (int i = 0; i < P.nbr; i++) sep_rows[i] = P.sep_rows[i];
        counts[0] = P.nbr; counts[1] = P.n_sep; counts[2] = P.strong ? P.n_strong1 : 0; counts[3] = P.strong ? P.n_strong2 : 0;
        counts[4] = (int32_t)(1000. * P.strong_contiguous + 0.5);
        return UZL_OK;
    } catch (...) { return UZL_ERR_OOM; }
}

namespace uzl {

void k_schur_gather(const PgoDev& D, const SchurDev& S, hipStream_t s)
{
    const long items = (long)(S.n_int + S.n_runs) * 36;
    if (items > 0 && S.runblk) hipLaunchKernelGGL(schur_gather_kernel, dim3((unsigned)((items + kBlk - 1) / kBlk)), dim3(kBlk), 0, s, D, S);
}
void k_schur_eliminate(const PgoDev& D, const SchurDev& S, hipStream_t s)
{
    if (S.n_runs > 0) hipLaunchKernelGGL(schur_eliminate_kernel, dim3(S.n_runs), dim3(128), 0, s, D, S);
}
void k_schur_assemble(const PgoDev& D, const PgoDev& R, const SchurDev& S, hipStream_t s)
{
    const long items = (long)(S.nslots_r + S.nbr) * 36;
    if (items > 0) hipLaunchKernelGGL(schur_assemble_kernel, dim3((unsigned)((items + kBlk - 1) / kBlk)), dim3(kBlk), 0, s, D, R, S);
}
void k_schur_backsub(const PgoDev& D, const PgoDev& R, const SchurDev& S, hipStream_t s)
{
    const int g = S.n_runs + (S.nbr * 6 + 63) / 64;
    if (g > 0) hipLaunchKernelGGL(schur_backsub_kernel, dim3(g), dim3(128), 0, s, D, R, S);
}
// grids: the largest over the slots of a pass (a twin leaves at once past its own graph's extent - the bodies check run / item counts)
void kl_schur_reduce(const LmSlot* sl, int nslots, int max_runs, long max_items, hipStream_t s)
{
    if (max_runs > 0) hipLaunchKernelGGL(schur_eliminate_lm_kernel, dim3(max_runs, 1, nslots), dim3(128), 0, s, sl);
    if (max_items > 0) hipLaunchKernelGGL(schur_assemble_lm_kernel, dim3((unsigned)((max_items + kBlk - 1) / kBlk), 1, nslots), dim3(kBlk), 0, s, sl);
}
void kl_schur_backsub(const LmSlot* sl, int nslots, int max_grid, hipStream_t s)
{
    if (max_grid > 0) hipLaunchKernelGGL(schur_backsub_lm_kernel, dim3(max_grid, 1, nslots), dim3(128), 0, s, sl);
}

}  // namespace uzl
