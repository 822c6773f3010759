// pgo_schur.hpp — Schur reduction of the chain interiors of a pose graph (north_star's "Schur-reduced PCG").
//
// The reference's graphs are an odometry chain (addOdometryEdge, graph_optimization/src/g2o_optimizer.cpp:190-259) plus loop
// closures (addFeatureEdge, :261-299).  In an online run most vertices carry nothing but their two odometry edges: in the block
// system (H + lambda I) dx = b such a vertex couples to its two chain neighbours only.  Maximal runs of those vertices are
// eliminated EXACTLY by block-tridiagonal elimination (one wave per run, runs are independent), the PCG then runs on the Schur
// complement over the remaining "separator" vertices (loop-closure endpoints, hubs, and every (cap+1)-th vertex of a long run so
// that no run is longer than `cap`), and the interiors follow by back-substitution.  Exact linear algebra on an SPD system: the
// LM trajectory is the one of the full solve up to the PCG tolerance.
//
//   run:  s0 - v1 - v2 - ... - vk - s1        (s0 / s1: separator, or absent = fixed vertex / end of the chain)
//   forward sweep, m = 1..k, with D'_1 = H_11 + lambda I, g'_1 = b_1, C_1 = H_{s0,v1}:
//       Dinv = D'_m^-1;  u_m = Dinv g'_m;  W_m = Dinv C_m^T;  T_m = Dinv E_m          (E_m = H_{vm, next})
//       S_L -= C_m W_m;  g_L -= C_m u_m;  C_{m+1} = -C_m T_m;  D'_{m+1} = H_{m+1,m+1} + lambda I - E_m^T T_m;  g'_{m+1} = b_{m+1} - E_m^T u_m
//   after vk:  S_R = -E_k^T T_k,  g_R = -E_k^T u_k,  F = C_{k+1} = fill block H'_{s0,s1}
//   back-substitution, m = k..1:   x_m = u_m - W_m x_{s0} - T_m x_{next}
// Round 4: a run is eliminated from BOTH ends at once (two waves per run): the sweep above over v1 .. v(j-1), its mirror image (s1 in the
// role of s0, E'_m = E_(m-1)^T) over vk .. v(j+1), then the middle vertex vj, which couples to s0 through C_L and to s1 through C_R:
//       W^L = Dinv C_L^T;  W^R = Dinv C_R^T;  S_L -= C_L W^L;  S_R -= C_R W^R;  g_L -= C_L u;  g_R -= C_R u;  F = -C_L W^R
//   and x_j = u_j - W^L x_{s0} - W^R x_{s1} first, then both halves outwards.  Half the chain of dependent steps; the same Schur complement.
//
// Numbering of the reduced system (round 4).  Small systems keep the separators in row (= trajectory) order.  From kSchurStrongMin
// separators on they can be numbered by STRONG AGGREGATES: the multilevel preconditioner gives every aggregate of 8 consecutive rows the six
// rigid-body modes as its coarse space, which only helps if those rows DO move nearly rigidly together - and in row order a group of 8
// consecutive separators holds the two ends of long soft runs while loop-closure partners sit in different groups (config 5's last
// re-optimisation: 70 - 140 PCG iterations per LM iteration).  schur_plan therefore weighs the reduced graph's edges (trace of the
// information matrix; a removed run = its edges as springs in series) and groups the separators by size-capped heavy-edge matching along
// edges that are at least theta = 0.25 x the stiffest edge at either end (<= 8 separators per group).  Up to one_level_max groups every
// group becomes one aggregate of the level-1 path (blocks of 8 rows); beyond, the groups are matched once more (<= 4 to a block) and laid
// out in blocks of 4 x 8 rows - the AGG = 4 geometry of the PCG kernels.  Either way groups and blocks are padded with EMPTY rows
// (sep_rows = -1: identity diagonal block, zero right-hand side and prolongation block; x stays 0).  Same linear system, same solution;
// which numbering a handle takes: uzl_pgo_cfg::reduced_numbering, build_structure (uzl_pgo.hip), DESIGN.md section 6.
#pragma once
#include <cstdint>
#include <vector>
#include "pgo_types.hpp"

namespace uzl {

// (SchurDev, kSchurElim, kSchurRunOut: pgo_types.hpp - the device view is part of LmSlot)

// host-side plan: which rows are eliminated, the runs, and the block-CSR of the reduced system
struct SchurPlan {
    int32_t nb = 0, nbr = 0, n_sep = 0, n_int = 0, n_runs = 0, nslots_r = 0, longest_run = 0;      // nbr: rows of the reduced system (n_sep separators + empty rows)
    bool strong = false; int32_t n_strong1 = 0, n_strong2 = 0;                                    // numbered by strong aggregates: groups, blocks of <= 4 groups (0: blocks of one group)
    double strong_contiguous = 1.;                           // share of the separators whose strong group is a run of consecutive separators anyway
    std::vector<int32_t> full2red;                            // [nb] reduced row or -1   (sep_rows: [nbr] full row, or -1 = an empty row)
    std::vector<int32_t> run_ptr, run_rows, slotP, slotN, endL, endR, sep_rows, rsrc, inc_ptr, inc;
    std::vector<int32_t> row_ptr, col;                        // reduced block-CSR
};

// Plans the reduction of a block-CSR (row_ptr / col over free vertices, col = -1 for a fixed neighbour).  `cap` = longest run.
SchurPlan schur_plan(int nb, const std::vector<int32_t>& row_ptr, const std::vector<int32_t>& col, int cap, const double* slot_w = nullptr,
                     int strong_min = 0, double theta = 0.25, double max_contiguous = 2., int one_level_max = 0, int min_interiors = 0);
// (min_interiors: the caller reduces only when at least so many rows are eliminated; below, the plan stops at the counts - the strong
//  grouping of the separators of a loopy 10k / 50k graph that has no chain interiors at all took 8 of its first solve's 13 ms of structure)

}  // namespace uzl
