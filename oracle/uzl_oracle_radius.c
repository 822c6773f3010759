/*
 * uzl_oracle_radius.c — CPU ORACLE (test infrastructure, NOT product code), see uzl_oracle.h.
 *
 * Restatement of the distance loop-closure candidate producer:
 *   SlamGraph::getNodesWithinRadius   graph_slam_common/src/slam_graph.cpp:266-278
 *   its caller                        graph_slam/src/graph_slam_node.cpp:272-289
 * For a query node q: every other node c (in node-map order = index order) with ||t_c - t_q|| < radius,
 * |stamp_q - stamp_c| > new_edge_time and a relative rotation below 30 degrees yields the job (from = c, to = q).
 * PARITY UNPINNED (no reference tests).  Rotation angle as Eigen::AngleAxisd [EXT], see uzl_oracle_gate.c.
 */
#include "uzl_oracle.h"

#include <math.h>
#include <stddef.h>

static double angle_of(const double R[9])
{
    double q[4];
    uzlo_quat_from_R(R, q);
    const double n2 = (q[1] * q[1] + q[2] * q[2]) + q[3] * q[3];
    if (n2 < 1e-12 * 1e-12) return 0.;
    double w = q[0];
    if (w < -1.) w = -1.;
    if (w > 1.) w = 1.;
    return 2. * acos(w);
}

int64_t uzlo_radius_candidates(int32_t n, const double* poses, const int64_t* stamp_front_ns, double radius, double new_edge_time,
                               double max_rotation_deg, int32_t nq, const int32_t* queries, int64_t cap, int32_t* out_from,
                               int32_t* out_to, int32_t* count_per_query)
{
    int64_t total = 0;
    for (int32_t j = 0; j < nq; j++) {
        const int32_t q = queries[j];
        int32_t cnt = 0;
        if (q >= 0 && q < n) {
            const double* Q = poses + 12 * (size_t)q;
            for (int32_t c = 0; c < n; c++) {                                              /* slam_graph.cpp:271-275 */
                if (c == q) continue;
                const double* C = poses + 12 * (size_t)c;
                const double dx = C[3] - Q[3], dy = C[7] - Q[7], dz = C[11] - Q[11];
                if (!(sqrt((dx * dx + dy * dy) + dz * dz) < radius)) continue;
                const double dts = fabs((double)(stamp_front_ns[q] - stamp_front_ns[c]) * 1e-9);   /* graph_slam_node.cpp:277 */
                if (!(dts > new_edge_time)) continue;
                double Rd[9];                                                               /* close^-1 * current: R = Rc^T Rq (:278) */
                for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++)
                    Rd[r * 3 + k] = (C[0 * 4 + r] * Q[0 * 4 + k] + C[1 * 4 + r] * Q[1 * 4 + k]) + C[2 * 4 + r] * Q[2 * 4 + k];
                const double diff_rotation = 180. * angle_of(Rd) / M_PI;                    /* :279-280 */
                if (!(fabs(diff_rotation) < max_rotation_deg)) continue;                    /* :282 */
                if (total < cap) { out_from[total] = c; out_to[total] = q; }                /* estimateEdge(close_node, current_node) :283 */
                total++; cnt++;
            }
        }
        if (count_per_query) count_per_query[j] = cnt;
    }
    return total;
}
