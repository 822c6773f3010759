/*
 * uzl_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's CPU algorithm for the hot path, used only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / reported
 * CPU baseline.  Nothing under uzliti_slam_amd/ may include, link or call this.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for this path
 *  and cannot be built here (ROS, Eigen, OpenCV, PCL, g2o, CSparse, boost are
 * all absent), so this oracle is pinned only by (a) line-by-line traceability to the cited
 * reference lines, (b) known-answer tests derived from the in-tree g2o excerpt
 * graph_slam_common/thirdparty/src/isometry3d_mappings.cpp, and (c) an independent
 * NumPy/SciPy second implementation in tests/np_reference.py.
 *
 * Third-party arithmetic restated from its published algorithm [EXT]:
 *   OpenCV 2.4  cv::BFMatcher(NORM_HAMMING)::knnMatch       (call site feature_transformation_estimator.cpp:38,58)
 *   PCL 1.7     pcl::TransformationFromCorrespondences       (call site :301-312)
 *   Eigen 3.2   JacobiSVD<Matrix3f>, Quaterniond(Matrix3d)   (inside PCL / isometry3d_mappings.cpp)
 *   g2o master (~2014-15) EdgeSE3, RobustKernelHuber, BlockSolver<6,3>, OptimizationAlgorithmLevenberg,
 *               LinearSolverCSparse                           (call sites g2o_optimizer.cpp:36-40,139,148,276-296)
 *
 * Deliberate, documented choices where the reference is unspecified (SURVEY §8c last row):
 *   - equal-distance matches are ordered by (distance, queryIdx)      (std::sort at :114 is unstable)
 *   - the RNG is a counter-based hash keyed by (seed, job_id, iteration) (std::random_shuffle/std::rand
 *     at :217 is unseeded, process-global and shared between threads)
 *   - the 3-point pose runs in float with a fixed operation order (two-sided Jacobi SVD) that the
 *     HIP kernels follow operation for operation; compile with -ffp-contract=off.
 */
#ifndef UZL_ORACLE_H
#define UZL_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- matching half ---------------- */

/* M1: cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, k=2)  (feature_transformation_estimator.cpp:58).
 * Ties: lower train index first.  idx = -1 when the train set is too small. */
void uzlo_knn2(const uint8_t* query, int32_t nq, const uint8_t* train, int32_t nt, int32_t bytes,
               int32_t* idx0, int32_t* d0, int32_t* idx1, int32_t* d1);

/* M2 + M4: ratio test d0 < 0.99*d1 (:65-71), 3-D validity filter (:101-112), sort by
 * (distance, queryIdx) (:114).  Returns M; *n_ratio = matches.size() (:78). */
int32_t uzlo_filter_sort(int32_t nq, const int32_t* idx0, const int32_t* d0, const int32_t* idx1,
                         const int32_t* d1, const uint8_t* valid_train, const uint8_t* valid_query,
                         int32_t* out_query, int32_t* out_train, int32_t* out_dist, int32_t* n_ratio);

/* M6a: sample of PROSAC iteration `iter` (3 indices into the sorted correspondence list). */
void uzlo_sample3(uint64_t seed, uint64_t job_id, int32_t iter, int32_t iterations, int32_t m,
                  int32_t do_prosac, int32_t out[3]);
/* PROSAC prefix length min(ceil((i+3.)/iterations*M), M)  (:217). */
int32_t uzlo_prosac_prefix(int32_t iter, int32_t iterations, int32_t m);

/* M7: estimatePoseSVD (:299-314) on columns idx[0..k) of P,Q (3 x M column-major); T = 12 doubles
 * row-major [R|t] with Q ~= T*P. idx may be NULL (use columns 0..k). */
void uzlo_pose_svd(const double* P, const double* Q, const int32_t* idx, int32_t k, double T[12]);
/* the float 3x3 SVD inside it (row-major A = U diag(s) V^T, s descending) */
void uzlo_svd3f(const float A[9], float U[9], float S[3], float V[9]);

/* M8: consensus3D (:337-347). Returns the count, fills set[M]. */
int32_t uzlo_consensus3d(const double* P, const double* Q, int32_t m, const double T[12],
                         double thresh, uint8_t* set);

/* M6: estimateSVD -> prosac (:178-297). */
void uzlo_prosac(const double* P, const double* Q, int32_t m, double max_error, int32_t iterations,
                 double break_percentage, int32_t do_prosac, uint64_t seed, uint64_t job_id,
                 double T[12], int32_t* consensus, double* mse, uint8_t* mask,
                 int32_t* iterations_run, int32_t* best_iteration);

/* M9: information matrix (:133-137), 36 doubles row-major. */
void uzlo_information(int32_t consensus, double mse, double info[36]);

typedef struct uzlo_frame {
    const uint8_t* desc; int32_t n; int32_t bytes_per_desc;
    const double* pos_xyz; const uint8_t* valid3d;
    int32_t feature_type; int32_t sensor_frame;
} uzlo_frame;

typedef struct uzlo_edge_result {
    int32_t ok, consensus, n_matches, n_corr, frame_from, frame_to, iterations_run, best_iteration;
    double mse, T[12], information[36];
} uzlo_edge_result;

/* estimateEdgeDirect (:32-159) for one node pair: `from` and `to` are arrays of FeatureData.
 * Optional outputs (capacity max_corr): corr_query/corr_train/corr_dist, inlier mask. */
void uzlo_estimate_edge(const uzlo_frame* from, int32_t n_from, const uzlo_frame* to, int32_t n_to,
                        double ransac_threshold, int32_t ransac_iteration, double break_percentage,
                        int32_t do_prosac, uint64_t seed, uint64_t job_id,
                        uzlo_edge_result* res, int32_t max_corr,
                        int32_t* corr_query, int32_t* corr_train, int32_t* corr_dist, uint8_t* mask);

/* Vote recipe of consensus3D: 0 = fused (default; what the HIP kernels compute), 1 = the reference build's unfused order
 * (uzl_oracle_match.c, "Vote recipes").  Process-global: tests only. */
void uzlo_set_vote_recipe(int32_t recipe);
void uzlo_vote_recipe_diff(const double* P, const double* Q, int32_t m, double max_error, int32_t iterations, int32_t do_prosac,
                           uint64_t seed, uint64_t job_id, int64_t* n_tests, int64_t* n_diff, double* min_margin);

/* Baseline builds only (-fopenmp): n_pairs independent single-frame pairs, `threads` estimator threads (the reference runs one
 * estimator thread per plugin instance; "all cores" = one instance per core).  Pair k uses frames from[k] / to[k], job id
 * job_id0 + k; results[k] as uzlo_estimate_edge leaves them.  Without OpenMP the loop is serial. */
void uzlo_estimate_edge_batch(int32_t n_pairs, const uzlo_frame* from, const uzlo_frame* to,
                              double ransac_threshold, int32_t ransac_iteration, double break_percentage,
                              int32_t do_prosac, uint64_t seed, uint64_t job_id0, int32_t threads, uzlo_edge_result* results);

/* ---------------- pose-graph half ---------------- */

/* in-tree g2o excerpt: graph_slam_common/thirdparty/src/isometry3d_mappings.cpp */
void uzlo_quat_from_R(const double R[9], double q[4]);            /* Eigen Quaterniond(R): (w,x,y,z)    */
void uzlo_R_from_quat(const double q[4], double R[9]);            /* Quaterniond::toRotationMatrix      */
void uzlo_to_vector_mqt(const double T[12], double v[6]);         /* :94-99                              */
void uzlo_from_vector_mqt(const double v[6], double T[12]);       /* :117-122                            */
void uzlo_to_euler(const double R[9], double rpy[3]);             /* :47-57                              */
void uzlo_from_euler(const double rpy[3], double R[9]);           /* :59-75                              */

/* g2o EdgeSE3::computeError [EXT]: e = toVectorMQT(Z^-1 * Xi^-1 * Xj) */
void uzlo_edge_error(const double Xi[12], const double Xj[12], const double Z[12], double e[6]);
/* g2o EdgeSE3::linearizeOplus [EXT]: analytic Ji, Jj (6x6 row-major) w.r.t. X <- X*fromVectorMQT(d) */
void uzlo_edge_jacobians(const double Xi[12], const double Xj[12], const double Z[12],
                         double Ji[36], double Jj[36]);
/* g2o RobustKernelHuber::robustify [EXT] */
void uzlo_huber(double e2, double delta, double rho[3]);

typedef struct uzlo_node { double pose[12]; int32_t fixed; } uzlo_node;
typedef struct uzlo_edge {
    int32_t from, to, type, sensor_from, sensor_to, valid;
    double transform[12], displacement_from[12], displacement_to[12], information[36];
    double diff_time;             /* |SlamEdge::diff_time_| [s] (g2o_optimizer.cpp:211) */
} uzlo_edge;

typedef struct uzlo_pgo_stats {
    int32_t iterations_done, lm_trials, terminated_early, n_vertices, n_edges, n_gauge_fixed;
    double chi2_initial, chi2_final, lambda_final;
    double t_order_ms, t_symbolic_ms, t_numeric_ms, t_linearize_ms, t_total_ms;
    int64_t factor_blocks;
} uzlo_pgo_stats;

/* G1: addGraphImpl flattening (g2o_optimizer.cpp:55-104,160-299). Outputs sized n / e:
 * poses (n x 12), fixed (n), ij (e x 2), meas (e x 12), info (e x 36), robust (e), src_edge (e) =
 * index of the input edge each system edge came from. Returns number of system edges. */
int32_t uzlo_flatten_graph(int32_t n_nodes, const uzlo_node* nodes, int32_t n_edges, const uzlo_edge* edges,
                           int32_t n_sensors, const double* sensors, int32_t optimize_xy_only,
                           int32_t use_odometry_parameters,
                           double* poses, uint8_t* fixed, int32_t* ij, double* meas, double* info,
                           uint8_t* robust, int32_t* src_edge);

/* g2o sclam2d OdomConvert [EXT] (g2o/types/sclam2d/odometry_measurement.cpp, called at g2o_optimizer.cpp:212-214):
 * motion (x, y, theta, dt) -> differential-drive wheel velocities (wheel base 1) -> motion.  Identity on exact
 * circular arcs; any other motion is projected onto the arc with the same heading change. */
void uzlo_odom_convert(double x, double y, double theta, double dt, double out_xyt[3]);

/* G2: setFixedNodes (g2o_optimizer.cpp:301-349): fixes the smallest-index vertex of every component
 * not reachable from a fixed vertex. Returns how many were fixed. */
int32_t uzlo_set_fixed_nodes(int32_t n, uint8_t* fixed, int32_t e, const int32_t* ij);

/* G3-G9: optimizer_.optimize(iterations) on the flattened problem: LM (g2o
 * OptimizationAlgorithmLevenberg) + sparse direct Cholesky (mirrors LinearSolverCSparse).
 * poses updated in place. */
int32_t uzlo_pgo_optimize(int32_t n, double* poses, const uint8_t* fixed, int32_t e, const int32_t* ij,
                          const double* meas, const double* info, const uint8_t* robust,
                          double huber_delta, int32_t iterations, uzlo_pgo_stats* stats);

/* G10: storeImpl edge error ||e||_2 (g2o_optimizer.cpp:124-131) */
void uzlo_edge_error_norms(int32_t n, const double* poses, int32_t e, const int32_t* ij,
                           const double* meas, double* err);

/* chi2 (activeRobustChi2) of the flattened problem */
/* baseline builds only (oracle/Makefile target `native`, -fopenmp): threads over the edges as g2o does when built with OpenMP;
 * the sparse Cholesky stays serial like CSparse.  Results do not depend on the thread count. */
void    uzlo_set_threads(int32_t threads);
int32_t uzlo_has_openmp(void);

double uzlo_chi2(int32_t n, const double* poses, int32_t e, const int32_t* ij, const double* meas,
                 const double* info, const uint8_t* robust, double huber_delta);

/* Dense assembly of the normal equations for cross-checks (tests only, small n): H is (6n)x(6n)
 * row-major, b is 6n; rows/cols of fixed vertices are zero. */
void uzlo_build_dense(int32_t n, const double* poses, const uint8_t* fixed, int32_t e, const int32_t* ij,
                      const double* meas, const double* info, const uint8_t* robust, double huber_delta,
                      double* H, double* b);

/* ---------------- edge filter (uzl_oracle_filter.c): TransformationFilter / EdgeCluster ----------------
 * transformation_filter.cpp:43-350.  Same POD layouts as include/uzl_mi355x.h so tests can share buffers. */
typedef struct uzlo_filter uzlo_filter;
typedef struct uzlo_filter_cfg {
    double max_dt, min_size; int32_t max_cluster_size, ransac_iterations; double max_error, min_time_span;
    int32_t max_edges, device; uint64_t seed;
} uzlo_filter_cfg;
typedef struct uzlo_filter_edge {
    uint64_t key; double matching_score; int32_t valid, sensor_from, sensor_to, n_stamps_from, n_stamps_to, _pad;
    const int64_t* stamps_from_ns; const int64_t* stamps_to_ns;
    double transform[12], displacement_from[12], displacement_to[12], pose_from[12], pose_to[12];
} uzlo_filter_edge;
typedef struct uzlo_cluster_info {
    uint64_t uid; int64_t from_start_ns, from_end_ns, to_start_ns, to_end_ns; int32_t size, consensus, changed, evaluations;
} uzlo_cluster_info;
void uzlo_filter_cfg_default(uzlo_filter_cfg* c);
uzlo_filter* uzlo_filter_create(const uzlo_filter_cfg* cfg);
void uzlo_filter_destroy(uzlo_filter* f);
void uzlo_filter_set_sensors(uzlo_filter* f, int32_t n, const double* sensors);
void uzlo_filter_add(uzlo_filter* f, int32_t n_edges, const uzlo_filter_edge* edges);
void uzlo_filter_remove(uzlo_filter* f, int32_t n_keys, const uint64_t* keys);
int32_t uzlo_filter_all_edges(const uzlo_filter* f, int32_t cap, uint64_t* keys);
int32_t uzlo_filter_calc_valid_edges(uzlo_filter* f);
int32_t uzlo_filter_valid_edges(const uzlo_filter* f, int32_t cap, uint64_t* keys);
int32_t uzlo_filter_cluster_count(const uzlo_filter* f);
void uzlo_filter_cluster_info(const uzlo_filter* f, int32_t idx, uzlo_cluster_info* o);
void uzlo_filter_cluster_edges(const uzlo_filter* f, int32_t idx, uint64_t* keys, uint8_t* valid);
int32_t uzlo_filter_cluster_last_eval(const uzlo_filter* f, int32_t idx, double* P, double* Q, double* T, int32_t* ransac_consensus);

/* ---------------- edge acceptance gate (uzl_oracle_gate.c): newEdgeCallback / checkEdgeHeuristic / astar ----------------
 * graph_slam_node.cpp:779-829,1064-1085; slam_graph.cpp:838-890.  Same POD layouts as include/uzl_mi355x.h. */
typedef struct uzlo_gate uzlo_gate;
typedef struct uzlo_gate_cfg {
    double min_matching_score, max_edge_distance_T, max_edge_distance_R, scope_size_factor, min_accept_valid;
    int32_t device, pad;
} uzlo_gate_cfg;
typedef struct uzlo_gate_edge { int32_t from, to, type, valid; double matching_score; double transform[12]; } uzlo_gate_edge;
void uzlo_gate_cfg_default(uzlo_gate_cfg* c);
uzlo_gate* uzlo_gate_create(const uzlo_gate_cfg* cfg);
void uzlo_gate_destroy(uzlo_gate* g);
void uzlo_gate_set_graph(uzlo_gate* g, int32_t n, const double* poses, const uint8_t* merged, int32_t ne, const uzlo_gate_edge* edges);
double uzlo_gate_astar(uzlo_gate* g, int32_t source, int32_t target);
void uzlo_gate_check(uzlo_gate* g, int32_t nc, const uzlo_gate_edge* cand, uint8_t* accept, uint8_t* valid, double* astar_dist);
int32_t uzlo_gate_edge_count(const uzlo_gate* g);
int64_t uzlo_gate_last_expansions(const uzlo_gate* g);

/* ---------------- distance loop-closure candidates (uzl_oracle_radius.c) ----------------
 * slam_graph.cpp:266-278 + graph_slam_node.cpp:272-289.  Returns the total number of jobs (may exceed cap). */
int64_t uzlo_radius_candidates(int32_t n, const double* poses, const int64_t* stamp_front_ns, double radius, double new_edge_time,
                               double max_rotation_deg, int32_t nq, const int32_t* queries, int64_t cap, int32_t* out_from,
                               int32_t* out_to, int32_t* count_per_query);

/* ---------------- appearance-based candidates (uzl_oracle_places.c): FastLshSet / LshSetRecognizer / PlaceRecognizer ---------
 * lsh_set_recognizer.cpp:46-310, place_recognizer.cpp:71-215.  Same cfg layout as include/uzl_mi355x.h. */
typedef struct uzlo_places uzlo_places;
typedef struct uzlo_places_cfg {
    int32_t key_width, min_rows_to_add; double T; int32_t k_nearest_neighbors, device; double min_time_gap;
} uzlo_places_cfg;
void uzlo_places_cfg_default(uzlo_places_cfg* c);
uzlo_places* uzlo_places_create(const uzlo_places_cfg* cfg);
void uzlo_places_destroy(uzlo_places* h);
int32_t uzlo_places_search_and_add(uzlo_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns, int32_t cap,
                                   int32_t* neighbors, int32_t* place_index);
int32_t uzlo_places_add(uzlo_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns);
int32_t uzlo_places_search(uzlo_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns, int32_t id_q,
                           int32_t cap, int32_t* neighbors);
void uzlo_places_remove(uzlo_places* h, int32_t id, const uint8_t* desc, int32_t rows, int32_t bytes);
int32_t uzlo_places_count(const uzlo_places* h);
int32_t uzlo_places_num_tables(const uzlo_places* h);
int32_t uzlo_places_last_counts(const uzlo_places* h, int32_t cap, int32_t* counts);

#ifdef __cplusplus
}
#endif
#endif
