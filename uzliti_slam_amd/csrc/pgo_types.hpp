// pgo_types.hpp — device-side view of one pose-graph problem (shared by host code and kernels).
//
// HBM layout (all f64 unless noted):
//   pose / pose_trial  [n][8]   (tx,ty,tz, qw,qx,qy,qz, pad): one 64-byte line per vertex gather
//   edge arrays, SoA   ei/ej [e] i32; zinv [7][e] = Z^-1 as (t, q); info [36][e]; robust [e] u8
//   block-CSR of H over the free vertices ("half-edge slots"): row a holds one slot per incident
//   system edge, sorted by edge index; slot s carries
//       blk [s][36]   H_{a,col[s]} = J_a^T W J_col      (col = -1 when the neighbour is fixed)
//   and contributes its edge's share of H_aa = J_a^T W J_a and of -b_a = J_a^T W e; the Hessian build (pgo_kernels.hip: hessian_kernel)
//   computes the shares of a row's slots side by side and adds them in slot order: no atomics, bit-reproducible, nothing but H itself in HBM.
#pragma once
#include <cstdint>
#include "../../include/uzl_mi355x.h"

namespace uzl {
// LDS budget of the PCG kernels and what ml_cg stages when the dense level-2 operator is present (pgo_handle.hpp: ml_comp4_fits)
constexpr size_t kMlLdsLimit = (size_t)140 * 1024;
__host__ __device__ inline size_t ml_comp4_lds(int n2) { return (size_t)6 * (size_t)n2 * 8 + 64; }


constexpr int kMaxPartials = 8192;        // block partials of a reduction = workgroups of a launch that leaves some (4096 until round 5: ml_spmv's 16-row workgroups ended at 65k vertices)
constexpr int kProgressEvery = 2;    // PCG iterations between two looks of the stop test (see pgo_device.hpp, progress_decide)
constexpr int kRowHdr = 24;          // ints per row header (96 B: one and a half cache lines)

struct PgoDev {
    int32_t n, nb, e, nslots;
    int32_t e_begin, e_end;   // system edges linearised by this rank (sharded solve); [0, e) otherwise
    int32_t diag_owner;       // 1 on the rank that adds the (H_aa + lambda) p term and the diagonal Galerkin parts
    int32_t sibling0;         // 1: the level-0 smoother couples the 8 rows of an aggregate (0 in the sharded solve, whose
                              //    ranks hold only their own edges' off-diagonal blocks)
    double* pose;
    double* pose_trial;
    const int32_t* v2b;      // [n]  free-block index or -1
    const int32_t* b2v;      // [nb]
    const int32_t* ei;
    const int32_t* ej;
    const double* zinv;      // [7][e]
    const double* info;      // [36][e]
    const uint8_t* robust;   // [e]
    const int32_t* row_ptr;  // [nb+1]
    const int32_t* col;      // [nslots]
    const int32_t* rowhdr;   // [nb][kRowHdr] = {row_ptr[a], row_ptr[a+1], col of the first 20 slots (-1 past the end), pad}: one hop instead of two
    const int32_t* rb_ptr;    // [n_rb + 1] row blocks of the Hessian build: consecutive rows with <= 256 slots, <= 42 rows where the graph allows
    int32_t n_rb, pad_rb;
    const double* srec;       // [22][nslots][2] the inputs of the slot's edge, slot-major in 16-byte pieces: Z^-1 (7) | Omega (36) | pad - a wave of the Hessian build reads 64 slots' k-th piece as one contiguous kilobyte
    const int4* smeta;        // [nslots] {2 * edge + side, pose index of the edge's first vertex, of its second, (slot in the other endpoint's row + 1) | robust << 30}
    double* blk;
    double* hdiag;           // [nb][36]
    double* minv;            // [nb][36]  (H_aa + lambda I)^-1   (block-Jacobi path only; the multilevel path uses MlLevel::Winv)
    double* b;               // [nb][6]
    double* x;               // PCG vectors [nb][6]
    double* xs;              // x as it was at the last look of the stop test (progress_decide, pgo_device.hpp)
    double* r;
    double* z;
    double* p;
    double* ap;
    double* part_a;          // [kMaxPartials] block partials (p.Ap, chi2, ...)
    double* part_b;          // [kMaxPartials] block partials (r.z, scale, ...)
    double* part_c;          // [kMaxPartials] block partials (max |H_jj|)
    double* scal;            // [16]: 0 rz, 1 rz threshold, 2 rz_prev, 3 lambda, 4 chi2, 5 scale, 6 diagmax, 7 |r|^2 / |b|^2 after PCG,
                             //       8 factor on pcg_tol^2 for this LM iteration's solves (host: do_optimize), 9 alpha and 10 breakdown of the current PCG
                             //       iteration (ml_alpha_kernel -> ml_cg_kernel); 11 r.z at the last progress check (first: of r_0), 12 / 13 the
                             //       step accuracy asked for [m] / [rad], 14 |b|^2 and 15 the movement of x over the last look's window in units of 12 / 13 (multilevel path; block-Jacobi path: 14 / 15 the estimates in m / rad);
                             //       0..7 go back to the host
    int32_t* flags;          // [4]: 0 done, 1 iterations, 2 breakdown, 3 this iteration's ml_cg leaves the stop test's partials (set by ml_spmv)
};

// ---- multilevel preconditioner (aggregation hierarchy with rigid-body-mode coarse spaces) -------------
// Level 0 = free vertices.  Level l+1 aggregates 8 consecutive level-l entities (index order = time order =
// the odometry chain), until at most 8 aggregates remain; that top level is solved exactly (dense), every
// level below contributes a block-diagonal smoother (additive multilevel, BPX style):
//     z = W0^-1 r + P1 ( W1^-1 r1 + P2 ( W2^-1 r2 + ... + P_L A_L^-1 r_L ) ),   r_l = P_l^T r_{l-1}
// where W_l is A_l(lambda) restricted to SIBLINGS (the children of one level-(l+1) aggregate: a dense block of
// (6 fan)^2 <= 48^2), inverted per LM trial.  Sibling blocks instead of 6x6 diagonal blocks cost one 48x48 matvec
// slice per workgroup and cut the PCG iterations by a third (the 8 consecutive vertices of an aggregate are
// tied by the stiff odometry chain).
// The coarse unknown of an aggregate is a world-frame twist (v, w) about the aggregate's centroid c, the
// exact null space of a pose graph; for a vertex i (R_i, t_i) in local MQT coordinates
//     P_i = [[R_i^T, -R_i^T [t_i - c]x], [0, 1/2 R_i^T]],  and between levels  P = [[I, -[c_child - c]x], [0, I]].
constexpr int kMlMaxLevels = 8;
constexpr int kMlFanout = 8;         // level 1: 8 vertices per aggregate; levels >= 3: 8 children
constexpr int kMlFanout2 = 4;        // level 2: 4 level-1 aggregates = 32 vertices = one workgroup of the PCG kernels
constexpr int kGalItems = 256;        // contributions one workgroup of ml_galerkin_kernel transforms per pass (LDS: 36 doubles each)
constexpr int kGalOutputs = 64;       // output blocks per chunk
constexpr int kMlTopMax = 8;          // aggregates at the top level (<= 48 dof dense) when the PCG kernels walk the hierarchy themselves
constexpr int kMlTopWide = 16;        // ... when they apply the dense composite operator instead: the top level is then only ever touched by
                                      // the rebuild, whose one-workgroup inverse (ml_top_kernel) takes 96 rows as readily as 48

struct MlLevel {
    int32_t n;                 // entities at this level (level 0: nb)
    int32_t fan;               // children per aggregate (levels >= 1)
    int32_t span;              // level-0 blocks under one aggregate
    int32_t nslots;            // off-diagonal blocks of A_l (level 0: the block-CSR above)
    const int32_t* row_ptr;    // [n+1]  (levels >= 1)
    const int32_t* col;        // [nslots]
    const int32_t* srow;       // [nslots] row of each slot (levels >= 1)
    // contribution ranges used when this level is the COARSE side of a Galerkin product A_{l+1} = P^T A_l P
    const int32_t* off_ptr;    // [nslots+1] contributions of each off-diagonal block
    const int32_t* diag_ptr;   // [n+1]      same-aggregate contributions of each diagonal block
    int32_t n_off_contrib;     // diag contributions start here in the contribution array
    const int32_t* cslot;      // [contributions] the fine-level slot behind each (ml_galerkin_kernel gathers)
    const int32_t* chunk;      // [n_chunks][5] = {0 off-diagonal / 1 diagonal outputs, first output, outputs, first contribution, contributions}
    int32_t n_chunks;
    double* blk;               // [nslots][36]
    double* G;                 // [n][36]   diagonal blocks of A_l(lambda = 0)
    double* M;                 // [n][36]   diagonal blocks of P^T P chain (lambda multiplier)
    double* geo;               // level 0: [n][12] = R^T (9), d (3); levels >= 1: [n][3] = d = c_self - c_parent
    double* cen;               // [n][4]  centroid, vertices underneath (levels >= 1)
    double* r;                 // [n][6]  restricted residual
    double* y;                 // [n][6]  coarse correction
    double* Winv;              // [n_{l+1}][(6 fan_{l+1})^2]  inverse sibling blocks of THIS level's entities (levels < L)
};

struct MlDev {
    int32_t levels;            // number of coarse levels L (0 = plain block-Jacobi)
    int32_t comp_level;        // composite path: level whose dense operator Ydense[comp_level] the PCG kernel applies (1 or 2); 0 = off
    MlLevel lv[kMlMaxLevels + 1];
    double* top_inv;           // [(6 n_top)^2] dense inverse of A_L(lambda)
    const int32_t* grp_beg[kMlMaxLevels + 1];   // composite path, levels comp_level .. L-1: [n_l][n_{l+1}] first / one-past-last slot of
    const int32_t* grp_end[kMlMaxLevels + 1];   // level-l row i whose column lies in level-(l+1) aggregate p (slots of a row are sorted by
                                                // column, so the range is contiguous)
    double* mQ; double* mQY;    // composite path, level 1: [n_1][n_2][36] Q and Q Y_2 of the multiplicative operator
    double* nsT; double* nsX;   // composite path: (6 n_1)^2 scratch of the Newton-Schulz refinement of Y_1
    double* Ydense[kMlMaxLevels + 1];   // composite path: Y_l = dense (6 n_l)^2 operator "residual of level l -> correction of
                               // level l" of the whole hierarchy above, 1 <= l < L (Y_L = top_inv); null otherwise
    double* Sg;                // [n_g][6] restriction of A p at the gather level (written by ml_spmv)
};

// Hot subset of MlDev passed BY VALUE to the per-iteration kernels (kernel arguments are preloaded; going
// through the MlDev pointer costs one extra dependent scalar-load round trip per first use).
struct MlHot {
    int32_t levels;
    int32_t n[kMlMaxLevels + 1];
    int32_t fan[kMlMaxLevels + 1];
    const double* geo0;                    // [nb][12]
    const double* geo[kMlMaxLevels + 1];   // [n_l][3], l >= 1
    const double* Winv[kMlMaxLevels + 1];  // [n_{l+1}][(6 fan_{l+1})^2], 0 <= l < levels
    const double* top_inv;
    const double* Cmat;                    // composite path: Y_cl as built (f64), [6 n_cl][6 n_cl]; null = no dense operator
    const float* Cmat32;                   // the copy the PCG kernels apply: Y_cl rounded to f32, [6 n_cl][c32_stride] (rows 6A..6A+5 belong to
    int32_t c32_stride;                    // workgroup A; stride = 6 n_cl rounded up to 4, pad = 0).  A preconditioner needs no more, the
    int32_t c32_pad;                       // operator is still one fixed linear map per solve, and it is the kernels' largest stream.
    double* Sg;                            // [n_g][6] restriction of A p at the gather level g = min(2, levels) (gather level 2: [n_2][6][2], see sg_at)
    double* Vg;                            // [6 n_2] gather-level residual estimate rg - alpha Sg prepared by ml_alpha_kernel (graphs of 12k .. 21.8k vertices)
};

constexpr int kBatchMax = 256;      // graphs per uzl_pgo_batch
constexpr int kBatchLaneMin = 12;   // from this many graphs on a batch runs as two launch sequences (uzl_pgo_batch, uzl_pgo.hip)

// ---- Schur reduction of chain interiors: device view (recurrences and host-side plan in pgo_schur.hpp) ----------------------------
constexpr int kSchurElim = 78;        // doubles kept per eliminated vertex: u (6) | W (36) | T (36)
constexpr int kSchurRunOut = 120;     // doubles written per run: S_L (36) | g_L (6) | S_R (36) | g_R (6) | F (36)

// device view (by-value kernel argument)
struct SchurDev {
    int32_t n_runs, n_int, nbr, nslots_r;
    const int32_t* run_ptr;     // [n_runs+1] into run_rows / slotP / slotN
    const int32_t* run_rows;    // [n_int] full-system row of every eliminated vertex, in chain order
    const int32_t* slotP;       // [n_int] slot (full block-CSR) of the block H_{v, previous element of the run / s0}; -1 = none
    const int32_t* slotN;       // [n_int] slot of H_{v, next element / s1}; -1 = none
    const int32_t* endL;        // [n_runs] reduced row of s0, -1 = none
    const int32_t* endR;        // [n_runs] reduced row of s1, -1 = none
    const int32_t* sep_rows;    // [nbr] full-system row of every reduced row (ascending)
    const int32_t* rsrc;        // [nslots_r] >= 0: slot of the full system whose block is copied; < 0: -(2 run + side) - 1, fill block F (side 0) / F^T (side 1)
    const int32_t* inc_ptr;     // [nbr+1] runs incident to each reduced row
    const int32_t* inc;         // 4 run + side: 0 = row is s0 (S_L, g_L), 1 = row is s1 (S_R, g_R), 2 = s0 == s1 (S_L + S_R + F + F^T, g_L + g_R)
    double* runblk;             // sharded solve: [n_int + n_runs][36] the runs' chain blocks (E_m per eliminated vertex, then C_1 per run) summed
                                //                over the ranks - every rank holds only its own edges' blocks; null = read them from PgoDev::blk
    double* elim;               // [n_int][kSchurElim]
    double* runout;             // [n_runs][kSchurRunOut]
};

// ---- device-resident Levenberg-Marquardt loop (uzl_pgo_lm.hip) --------------------------------------------------------------------
// The accept / reject decisions of g2o's OptimizationAlgorithmLevenberg::solve [EXT] (graph_optimization/src/g2o_optimizer.cpp:148 calls
// it) are taken ON THE DEVICE: the state of the loop lives in an LmDev per graph, two one-workgroup kernels (lm_head_kernel in front of a
// trial's solve, lm_tail_kernel behind its evaluation) advance it, and every other kernel of an LM iteration is a "slot twin" that reads
// its arguments from an LmSlot and predicates itself on that state.  A PASS is a fixed launch sequence
//     linearise | head | [Schur reduction] | [set-up of the hierarchy copy in use] | [rebuild of the other copy, second stream]
//     | PCG init | PCG iterations x K | residual guard, [back-substitution], retraction, chi2 | tail (decide + publish)
// whose kernels no-op unless the graph is in the phase they serve - so a pass can be captured once per structure and replayed, several
// graphs at different LM iterations can share one (blockIdx.z = slot: the batched solve), and the host looks at the outcome once per pass.
enum LmPhase : int32_t {
    kLmLin = 0,        // the next pass linearises (start of an LM iteration)
    kLmSolve = 1,      // a trial's PCG is running (flags[0] = done tells whether it still iterates)
    kLmRetry = 2,      // the last trial was rejected: the next pass starts another one on the same linearisation
    kLmNeedSetup = 3,  // the head wants a set-up segment the pass did not carry (LmDev::need): the host enqueues a pass that has it
    kLmDone = 4,       // iterations exhausted or Terminate
    kLmAnomaly = 5     // PCG breakdown / not converged / residual guard: the host-driven loop solves this graph again from its start poses
};
enum LmNeed : int32_t { kNeedNumeric = 1, kNeedTrial = 2, kNeedRebuild = 4 };
enum LmPassFlags : int32_t { kPassSetup = 1 /* the pass carries the set-up segment of the copy in use */, kPassRebuild = 2 /* ... and a rebuild of the other copy */ };
struct LmDev {
    int32_t flags[4];          // = PgoDev::flags of the graph: 0 done (PCG kernels no-op), 1 PCG iterations, 2 breakdown, 3 look
    int32_t phase;             // LmPhase
    int32_t cur;               // pose buffer (LmSlot::pose) that holds the current estimate
    int32_t ix;                // hierarchy copy the PCG applies
    int32_t pass;              // passes this graph has seen (bumped by lm_head_kernel); the *_pass stamps below mean "due in that pass"
    int32_t init_pass, schur_pass, numeric_pass, trial_pass;
    int32_t build_pass, build_ix, build_cur, need;          // rebuild of copy build_ix from the poses in buffer build_cur; need: LmNeed bits
    int32_t it, qmax, iterations, max_it;                   // LM iteration, trial of it, iterations asked for, PCG iteration cap per solve
    int32_t pending, adopted, fresh, pcg_last;              // a rebuilt copy waits to be adopted / was adopted this iteration / this trial runs on fresh inverses
    int32_t always_refresh, sync_rebuild, guarded, tails;   // constants of the solve; tails: lm_tail_kernel launches seen (= sequence word of LmHost)
    int32_t st_pcg_iterations, st_lm_trials, st_precond_builds, st_iterations_done, st_terminated_early, anomaly_code, pad0, pad1;
    double lambda, ni, chi_cur, last_rel;
    double lambda_setup[2];    // lambda the inverses of each hierarchy copy were taken at
    double rate_ref, rate_last;
    double chi2_initial, tol_f2, eps_t, eps_r, refresh_rel, tol2, lambda_retake, delta;
    double scal2[8];           // [3]: lambda of a rebuild that runs ahead of the trial loop (the set-up kernels read scal[3])
    double rate_drop;          // share of its fresh PCG rate below which the hierarchy is rebuilt (lm_refresh)
};
// what the host sees after a pass: an image of the LM state and of PgoDev::scal[0..8), written by lm_tail_kernel into pinned coherent
// memory (one 8-byte word per lane); seq_begin with the fields, seq (= LmDev::tails) after every lane's stores have been acknowledged
// (publish_wait_own_stores, uzl_common.hpp - no fence on gfx9).  The host copies a snapshot before it enqueues the pass whose tail writes
// the next one: that ordering, not the two words, keeps a copy whole; seq_begin == seq around it is the cross-check
struct LmHost {
    LmDev lm;
    double scal[8];            // as the host-driven loop fetches them (verbose logs, residual ratio)
    uint32_t seq_begin, seq;
};
// everything a slot twin needs, per graph; uploaded when the structure of the graph changes
struct LmSlot {
    PgoDev D;                                  // the full system (pose / pose_trial unused: LmSlot::pose + LmDev::cur)
    PgoDev Dp;                                 // the system the PCG solves: D, or the Schur complement over the separator vertices
    SchurDev SD;                               // the reduction (valid when red)
    LmDev* lm;
    LmHost* pub;                               // device address of the pinned snapshot
    MlHot hot[2];                              // hot subset of the two hierarchy copies
    const MlDev* dml[2];
    double* rg[2][2];                          // per copy: double-buffered gather-level residual
    double* dense[2][kMlMaxLevels + 1];        // per copy: Ydense[l]
    double* nsT[2];
    double* nsX[2];
    double* pbuf[2];                           // PCG direction, ping-pong
    double* pose[2];
    int32_t g_edges, g_asm, g_oplus, g_rows, g_spmv, red, pad0, pad1;      // grids (= partial counts) of this graph's launches; red: Schur-reduced
    int64_t copy_stride;                       // bytes from an array of hierarchy copy 0 to the same array of copy 1 (one arena, two halves)
};

// launch geometry of a pass: what the host needs besides the slot table (one structure, or the common shape of a batch)
enum LmCgVariant : int32_t { kCgPlain1 = 0, kCgComp1 = 1, kCgPlain4 = 2, kCgComp4 = 3, kCgComp4Ypre = 4, kCgComp4Vpre = 5 };
struct LmShape {
    int32_t nslots;                          // graphs of the pass (blockIdx.z)
    int32_t batch_geometry;                  // 0: the single solve's workgroups (shortest chain), 1: the batch's (most bytes in flight)
    int32_t levels, cl, agg;                 // hierarchy: coarse levels, level of the dense operator (0 = none), level-1 aggregates per PCG workgroup
    int32_t mult, ns_steps, upper_ns;        // multiplicative cycle + Newton-Schulz steps (composite level / the dense levels above)
    int32_t cg_variant, comp_u;              // LmCgVariant; kCgComp1: gather-level values per lane (5 / 8 / 12 / 16)
    int32_t n_lv[kMlMaxLevels + 2];          // entities per level (level 0: the largest graph)
    int32_t work_t[kMlMaxLevels + 2];        // Galerkin transform / reduce work per level (largest graph): nslots_l + n_l
    int32_t chunks[kMlMaxLevels + 2];        // ml_galerkin_kernel's workgroups per coarse level (largest graph)
    int32_t inner_aggs;                      // sibling blocks (largest graph)
    int32_t g_edges, g_asm, g_oplus, g_rows, g_spmv;      // largest grids
    int32_t red, schur_runs, schur_backsub_grid;
    int64_t schur_items;
    uint64_t cg_lds;                         // dynamic LDS of the ml_cg variant
};

// scalars handed back to the host after each LM trial / PCG chunk.  The struct lives in pinned, host-coherent memory
// that a one-workgroup kernel (publish_kernel) writes directly; `seq` is stored last with system-scope release, the
// host spins on it - a few microseconds instead of the copy + stream-synchronise round trip.
struct PgoHostScal {
    double scal[8];
    int32_t flags[4];
    uint32_t seq;
    uint32_t pad;
};

}  // namespace uzl
