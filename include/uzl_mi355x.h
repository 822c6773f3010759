/*
 * uzl_mi355x.h — C ABI of libuzl_mi355x.so, the MI355X (gfx950) back end for the
 * one data-parallel hot path of uzliti_slam:
 *
 *   (1) feature edge estimation  = 2-NN Hamming match + ratio test + 3-D filter +
 *       PROSAC/RANSAC 3-point pose + refit + information matrix
 *       (reference: transformation_estimation/src/feature_transformation_estimator.cpp:32-347)
 *   (2) SE(3) pose-graph solve   = graph flattening, gauge fixing, Levenberg-Marquardt
 *       with Huber kernel, write-back
 *       (reference: graph_optimization/src/g2o_optimizer.cpp:55-349 + the g2o semantics
 *        it delegates to)
 *
 * Every entry point is extern "C", takes plain pointers and sizes and returns an int
 * status (0 = ok, <0 = error; never throws).  Inputs are borrowed for the duration of the
 * call and copied to HBM before the call returns; outputs go to caller-provided buffers.
 * Handles are opaque and thread-safe at handle granularity (one mutex per handle).
 *
 * The reference-side classes these functions sit under are
 *   TransformationEstimator / FeatureTransformationEstimator
 *     (transformation_estimation/include/transformation_estimation/transformation_estimator.h:45-67,
 *      .../feature_transformation_estimator.h:33-60)
 *   GraphOptimizer / G2oOptimizer
 *     (graph_optimization/include/graph_optimization/graph_optimizer.h:28-56,
 *      .../g2o_optimizer.h:38-68)
 * INTEGRATION.md shows the C++ subclasses a maintainer would add on the ROS side.
 *
 * Matrix conventions: an SE(3) transform is 12 doubles, row-major 3x4 [R|t]
 * (the top three rows of Eigen::Isometry3d::matrix()).  A 6x6 information matrix is 36
 * doubles row-major, parameter order (x,y,z,qx,qy,qz) as in SlamEdge::information_
 * (graph_slam_common/include/graph_slam_common/slam_edge.h:84).
 */
#ifndef UZL_MI355X_H
#define UZL_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the declarations between this push and the pop at the end of the file are its
 * whole dynamic symbol table (tests/test_capi_exports.py holds `nm -D --defined-only` to exactly this list). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define UZL_ABI_VERSION 3

/* ---- status codes (reference: bool returns + ROS_ERROR, SURVEY §8b "Errors") ---- */
#define UZL_OK                 0
#define UZL_ERR_BAD_ARG       -1
#define UZL_ERR_NO_DEVICE     -2
#define UZL_ERR_HIP           -3
#define UZL_ERR_NOT_CONVERGED -4   /* PCG hit pcg_max_iter in some LM trial (result still written) */
#define UZL_ERR_BUSY          -5   /* GraphOptimizer::optimize() returns false while a solve is in flight */
#define UZL_ERR_OOM           -6
#define UZL_ERR_NOT_FOUND     -7
#define UZL_ERR_STATE         -8   /* call order violated (e.g. optimize before set_graph) */

/* feature types: graph_slam_msgs/msg/Features.msg:1-4 */
#define UZL_FEATURE_BRIEF 1
#define UZL_FEATURE_ORB   2
#define UZL_FEATURE_BRISK 3
#define UZL_FEATURE_FREAK 4

/* edge types: the values of graph_slam_msgs/msg/Edge.msg:1-9, so that SlamEdge::type_ passes through unchanged
 * (TYPE_2D_WHEEL_ODOMETRY is the only one the
 * optimizer treats specially, g2o_optimizer.cpp:78) */
#define UZL_EDGE_TYPE_3D_FULL           1
#define UZL_EDGE_TYPE_3D_ROTATION       2
#define UZL_EDGE_TYPE_3D_TRANSLATION    3
#define UZL_EDGE_TYPE_3D_GPS            4
#define UZL_EDGE_TYPE_2D_FULL           101
#define UZL_EDGE_TYPE_2D_ROTATION       102
#define UZL_EDGE_TYPE_2D_TRANSLATION    103
#define UZL_EDGE_TYPE_2D_WHEEL_ODOMETRY 104
#define UZL_EDGE_TYPE_2D_LASER          105

int         uzl_abi_version(void);
/* The library's long-lived HIP streams on `device`.  Streams that have to run side by side (a solver handle's solver / rebuild pair,
 * the launch sequences of a batch and their rebuild streams) are leased from one pool per device and process: a pair of streams is measured against each
 * other at most once per process (0.1 - 0.5 ms: chains of short kernels timed on the device), its verdict is remembered, streams go back
 * to the pool when their handle is destroyed.  UZL_STREAM_PROBE=0 in the environment skips every measurement (a batch then runs as one
 * launch sequence).  Out (any may be NULL): streams in the pool / leased right now / made by handles for themselves and registered;
 * pairs measured so far / found independent; leases that found no independent stream within the budget; host time spent measuring
 * [ms] (without the one-time set-up - two small allocations and the process's first kernel launches - and without the first launch
 * on a fresh stream, which whoever uses the stream first pays). */
int         uzl_stream_stats(int32_t device, int32_t* n_pooled, int32_t* n_leased, int32_t* n_registered, int32_t* pairs_measured,
                             int32_t* pairs_independent, int32_t* fallbacks, double* probe_ms);
/* Number of visible HIP devices, or <0 (UZL_ERR_NO_DEVICE) when there is none. */
int         uzl_device_count(void);
/* Static string for a status code. */
const char* uzl_status_string(int status);

/* ======================================================================================
 *  Edge estimation  (TransformationEstimator family)
 * ====================================================================================== */

typedef struct uzl_match uzl_match;

/* Mirrors transformation_estimation/cfg/FeatureLinkEstimation.cfg:9-13 field for field,
 * then the back-end additions.  uzl_match_cfg_default() fills the cfg-file defaults. */
typedef struct uzl_match_cfg {
    double   ransac_threshold;         /* 0.2   max 3-D distance of an inlier [m]              */
    double   link_covariance;          /* 0.01  (unused by the live reference code)           */
    int32_t  ransac_iteration;         /* 100   number of PROSAC iterations                   */
    double   ransac_break_percentage;  /* 0.6   early exit when consensus > pct * M           */
    int32_t  use_epnp;                 /* 1     (unused by the live reference code)           */
    /* ---- back-end additions ---- */
    int32_t  do_prosac;                /* 1     growing-prefix sampling (estimateSVD default) */
    int32_t  device;                   /* HIP device ordinal                                  */
    uint64_t seed;                     /* counter-based RNG seed; replaces the reference's
                                          unseeded process-global std::rand      */
} uzl_match_cfg;

/* One FeatureData (graph_slam_common/include/graph_slam_common/sensor_data.h:49-70). */
typedef struct uzl_frame {
    const uint8_t* desc;            /* n rows x bytes_per_desc, row-major (cv::Mat CV_8U)       */
    int32_t        n;               /* number of keypoints (features_.rows)                     */
    int32_t        bytes_per_desc;  /* 32 = ORB/BRIEF-256, 64 = BRISK/FREAK-512; multiple of 4  */
    const double*  pos_xyz;         /* 3 x n column-major (Eigen::MatrixXd feature_positions_)  */
    const uint8_t* valid3d;         /* n flags (std::vector<bool> valid_3d_)                    */
    int32_t        feature_type;    /* UZL_FEATURE_*                                            */
    int32_t        sensor_frame;    /* integer key standing for the sensor_frame_ string        */
    double         displacement[12];/* SensorData::displacement_                                */
} uzl_frame;

/* One node-pair job = one call of estimateEdgeImpl(from, to, edge)
 * (feature_transformation_estimator.cpp:161-171).  A node may carry several FeatureData;
 * frame ids index the handle's resident frame store and are given through the flat
 * frame_ids array passed next to the jobs. */
typedef struct uzl_pair_job {
    uint64_t job_id;       /* keys the RNG stream; echoed in the result                       */
    int32_t  from_begin;   /* frames of node `from`: frame_ids[from_begin .. +from_count)     */
    int32_t  from_count;
    int32_t  to_begin;     /* frames of node `to`                                             */
    int32_t  to_count;
} uzl_pair_job;

/* What estimateEdgeDirect() leaves in the SlamEdge (feature_transformation_estimator.cpp:127-156)
 * plus the diagnostics the parity tests compare. */
typedef struct uzl_edge_result {
    uint64_t job_id;
    int32_t  ok;               /* return value of estimateEdgeImpl (1 iff a sensor pair matched
                                  and >= 3 correspondences survived)                          */
    int32_t  consensus;        /* SlamEdge::matching_score_ (0 when !ok, transformation_estimator.cpp:53-55) */
    int32_t  n_matches;        /* ratio-test survivors of the chosen sensor pair (score, :78) */
    int32_t  n_corr;           /* M: survivors of the 3-D validity filter (:101-112)          */
    int32_t  frame_from;       /* chosen FeatureData pair (frame ids), -1 if none             */
    int32_t  frame_to;
    int32_t  iterations_run;   /* PROSAC iterations executed before the early exit            */
    int32_t  best_iteration;   /* iteration whose hypothesis won                              */
    double   mse;              /* mean inlier distance (:285-290)                             */
    double   T[12];            /* SlamEdge::transform_  (from_T_to)                           */
    double   information[36];  /* SlamEdge::information_ (:133-137)                           */
} uzl_edge_result;

void uzl_match_cfg_default(uzl_match_cfg* cfg);

/* FeatureTransformationEstimator::FeatureTransformationEstimator (…estimator.cpp:27-30). */
int  uzl_match_create(const uzl_match_cfg* cfg, uzl_match** out);
void uzl_match_destroy(uzl_match* h);
/* FeatureTransformationEstimator::setConfig (…estimator.cpp:350-353). */
int  uzl_match_set_config(uzl_match* h, const uzl_match_cfg* cfg);
const char* uzl_match_last_error(uzl_match* h);

/* Upload one FeatureData into the handle's HBM-resident frame store (the reference deep-copies
 * both SlamNodes per enqueue, transformation_estimator.cpp:39; here a frame is uploaded once and
 * referenced by every pair job that uses it).  Returns the frame id through *frame_id.  The frame's arrays are
 * borrowed for the duration of the call only: they are packed into pinned staging memory and go up as one
 * asynchronous copy on the handle's stream, in front of whatever the handle is asked to do next. */
int  uzl_match_add_frame(uzl_match* h, const uzl_frame* frame, int32_t* frame_id);
/* n FeatureData in one call (what the adapter's worker has queued; the reference copies one node pair per estimateEdge call,
 * transformation_estimator.cpp:35-43): one contiguous extent of the frame store, packed into pinned staging by several host
 * threads and uploaded half by half (32 MB per DMA) while the next half is being packed.  frame_ids: n_frames entries. */
int  uzl_match_add_frames(uzl_match* h, int32_t n_frames, const uzl_frame* frames, int32_t* frame_ids);
/* Hands the frame's extent back to the store's free list (first fit, neighbours merged; a frame removed while a batch is in
 * flight is freed by that batch's collect): a node that adds and removes frames for hours - the reference merges and deletes
 * nodes continuously, graph_slam_node.cpp:665-777 - holds what is alive, not what was ever uploaded. */
int  uzl_match_remove_frame(uzl_match* h, int32_t frame_id);
int  uzl_match_frame_count(uzl_match* h);
/* bytes of the frame store: held by live frames / high-water mark of the arena / allocated.  Any of the three may be NULL. */
int  uzl_match_arena_bytes(uzl_match* h, uint64_t* live, uint64_t* high_water, uint64_t* capacity);

/* Batched estimateEdgeImpl: n_jobs independent node pairs in one launch sequence.
 * Optional diagnostics (may each be NULL): per job, at stride max_corr,
 *   corr_query / corr_train : the sorted correspondence list (DMatch queryIdx / trainIdx
 *                             after std::sort, :114), first n_corr entries valid
 *   corr_dist               : their Hamming distances
 *   inlier_mask             : final consensus set (maxConsensusSet after the refit, :258)
 * Blocks until the results are in `results`. */
int  uzl_match_estimate(uzl_match* h,
                        int32_t n_jobs, const uzl_pair_job* jobs,
                        const int32_t* frame_ids, int32_t n_frame_ids,
                        uzl_edge_result* results,
                        int32_t max_corr,
                        int32_t* corr_query, int32_t* corr_train, int32_t* corr_dist,
                        uint8_t* inlier_mask);

/* Split form of uzl_match_estimate for pipelining: launch enqueues every kernel and the
 * D2H copies on the handle's stream and returns; collect waits for them. One batch may be
 * in flight per handle (UZL_ERR_BUSY otherwise). */
int  uzl_match_launch(uzl_match* h, int32_t n_jobs, const uzl_pair_job* jobs,
                      const int32_t* frame_ids, int32_t n_frame_ids, int32_t max_corr);
int  uzl_match_collect(uzl_match* h, uzl_edge_result* results,
                       int32_t* corr_query, int32_t* corr_train, int32_t* corr_dist,
                       uint8_t* inlier_mask);

/* Stage M1 alone, for parity tests: cv::BFMatcher(NORM_HAMMING).knnMatch(query=to, train=from, k=2)
 * (feature_transformation_estimator.cpp:38,58).  Outputs have n(to) entries; an index is -1 when
 * the train set has fewer rows than the rank asks for. */
int  uzl_match_knn2(uzl_match* h, int32_t frame_from, int32_t frame_to,
                    int32_t* idx0, int32_t* dist0, int32_t* idx1, int32_t* dist1);

/* FeatureTransformationEstimator::estimateSVD (…estimator.cpp:178-184) on caller-supplied
 * correspondences, batched: problem b uses columns [offsets[b], offsets[b+1]) of P and Q
 * (3 x total column-major).  This is the entry TransformationFilter::EdgeCluster uses
 * (transformation_filter.cpp:272-275).  Outputs per problem: T (12), consensus, mse,
 * iterations_run; mask is per column.  job_ids key the RNG streams. */
int  uzl_ransac_points(uzl_match* h, int32_t n_problems, const int32_t* offsets,
                       const double* P, const double* Q,
                       double max_error, int32_t iterations, double break_percentage,
                       int32_t do_prosac, const uint64_t* job_ids,
                       double* T, int32_t* consensus, double* mse, int32_t* iterations_run,
                       uint8_t* mask);

/* Per-kernel timing of the last estimate/launch, measured with HIP events on the handle's
 * stream when profiling is on.  names/ms arrays of capacity cap; returns the number filled. */
int  uzl_match_set_profiling(uzl_match* h, int32_t on);
int  uzl_match_kernel_times(uzl_match* h, int32_t cap, const char** names, double* ms, int32_t* launches);

/* ======================================================================================
 *  Pose-graph optimisation  (GraphOptimizer family)
 * ====================================================================================== */

typedef struct uzl_pgo uzl_pgo;

/* Mirrors graph_optimization/cfg/GraphOptimizer.cfg:10-12, then the back-end additions. */
typedef struct uzl_pgo_cfg {
    int32_t iterations;               /* 20   LM outer iterations (optimizer_.optimize(iterations), g2o_optimizer.cpp:148) */
    int32_t use_odometry_parameters;  /* 0    differential-drive round trip of odometry edges (g2o_optimizer.cpp:209-227)  */
    int32_t optimize_xy_only;         /* 0    project poses/measurements to (x,y,yaw) (g2o_optimizer.cpp:164-170)         */
    /* ---- back-end additions ---- */
    int32_t device;
    double  pcg_tol;                  /* 1e-5  accuracy asked of every linear solve.  pcg_stop = 0 (default): the solve ends when the
                                         estimated error left in the LM step is below pcg_tol metres in every translation component
                                         and 0.1 * pcg_tol in every quaternion-vector component (~0.2 * pcg_tol rad) AND the
                                         residual has come down (see pcg_stop); r.M^-1 r <= 1e-4 * pcg_tol^2 * (r0.M^-1 r0) is kept
                                         as a floor.  pcg_stop = 1: the plain relative test r.M^-1 r <= pcg_tol^2 * (r0.M^-1 r0)
                                         (g2o's LinearSolverPCG stops at 1e-6 on that squared norm, i.e. pcg_tol = 1e-3 [EXT]) */
    int32_t pcg_max_iter;             /* per linear solve                                             */
    int32_t schur_reduce;             /* 0 = auto: vertices that carry nothing but their two chain (odometry, g2o_optimizer.cpp:190-259)
                                         edges are eliminated exactly from (H + lambda I) per LM trial and PCG runs on the Schur
                                         complement over the rest, when they are a third or more of the free vertices; -1 = never.
                                         (occupies what used to be padding: layout unchanged)           */
    double  huber_delta;              /* 1.0  (g2o_optimizer.cpp:293)                                 */
    int32_t verbose;
    int32_t preconditioner;           /* 1 = additive multilevel (8-vertex aggregates, rigid-body modes), 0 = block-Jacobi.  The multilevel
                                         hierarchy serves systems of up to ~95 000 free vertices (after the elimination of chain interiors:
                                         a 400 000-node chain-like graph reduces far below that); larger ones are solved with block-Jacobi
                                         whatever this says - correct, but an order of magnitude slower on loopy graphs */
    int32_t pcg_stop;                 /* 0 = step-error estimate (above), 1 = relative residual test only                  */
    int32_t lm_loop;                  /* 0 = Levenberg-Marquardt decisions on the device, one host look per trial (captured passes);
                                         1 = host-driven loop (the one sharded and profiled solves always take); same results */
    int32_t reduced_numbering;        /* how the Schur-reduced system (schur_reduce) of a graph with >= 32 separators is laid out:
                                         1 = row (trajectory) order, 8 consecutive separators per aggregate; 2 = by strong aggregates
                                         (separators that are stiffly tied - loop-closure partners, short runs - share an aggregate);
                                         0 = the handle chooses: strong aggregates while they are few enough for the level-1 path
                                         (<= 256 groups) or differ from the row order (fewer than 60 % of the separators in groups that
                                         are consecutive anyway), and it changes its mind when the PCG iterations per LM trial of its
                                         own last solves say so.  Same linear system either way.
                                         (was reserved0: layout unchanged)                                              */
    int32_t pass_history;             /* The device-resident loop (lm_loop = 0) enqueues one pass per LM trial and has to size its PCG
                                         segment before the solve runs.  0 (default): a handle that is asked to optimise the SAME
                                         structure again (uzl_pgo_reset, a timer-driven re-optimisation, graph_slam_node.cpp:1138-1150)
                                         also uses the PCG iteration count every trial took in its previous uzl_pgo_optimize;
                                         1: sizes come from the running optimize alone (the previous solve's count), as on a fresh
                                         handle.  Results are the same either way - a pass that is too short is followed by another -
                                         only the number of passes and idle launches changes.
                                         (occupies the struct's tail padding: layout unchanged)                          */
} uzl_pgo_cfg;

/* SlamNode as the optimizer sees it (slam_node.h:89-107). Array order = std::map iteration order
 * of SlamGraph (lexicographic id), which is also the order g2o vertex ids are assigned in
 * (g2o_optimizer.cpp:64-66,180) and the order setFixedNodes() picks gauge vertices in (:338). */
typedef struct uzl_node {
    double  pose[12];   /* SlamNode::pose_   */
    int32_t fixed;      /* SlamNode::fixed_  */
} uzl_node;

/* SlamEdge as the optimizer sees it (slam_edge.h:78-92). */
typedef struct uzl_edge {
    int32_t from;                 /* index into nodes[] (id_from_), -1 if the node is missing  */
    int32_t to;                   /* index into nodes[] (id_to_)                               */
    int32_t type;                 /* UZL_EDGE_TYPE_*                                           */
    int32_t sensor_from;          /* index into sensors[] (sensor_from_), -1 = identity        */
    int32_t sensor_to;
    int32_t valid;                /* passes the TransformationFilter (g2o_optimizer.cpp:97-103);
                                     non-odometry edges with valid==0 are not optimised        */
    double  transform[12];        /* transform_           */
    double  displacement_from[12];/* displacement_from_   */
    double  displacement_to[12];  /* displacement_to_     */
    double  information[36];      /* information_         */
    double  diff_time;            /* |diff_time_| in seconds: read for odometry edges when
                                     use_odometry_parameters is set (g2o_optimizer.cpp:211)     */
} uzl_edge;

typedef struct uzl_pgo_stats {
    int32_t iterations_done;   /* LM outer iterations completed (return value of optimize())  */
    int32_t lm_trials;         /* total inner trials (linear solves)                          */
    int32_t pcg_iterations;    /* total PCG iterations over all solves                        */
    int32_t terminated_early;  /* LM returned Terminate (10 rejections or rho == 0)           */
    int32_t n_vertices;        /* vertices in the system                                      */
    int32_t n_edges;           /* edges in the system (after the skip rules)                  */
    int32_t n_gauge_fixed;     /* vertices fixed by setFixedNodes()                           */
    int32_t pcg_not_converged; /* solves that hit pcg_max_iter                                */
    double  chi2_initial;      /* activeRobustChi2 before the first iteration                 */
    double  chi2_final;
    double  lambda_final;
    double  solve_ms;          /* wall time of uzl_pgo_optimize, device-resident graph        */
    int32_t precond_builds;    /* LM iterations that rebuilt the multilevel preconditioner    */
    int32_t exchange_calls;    /* sharded solve: all-reduce calls issued by this optimize()   */
    double  structure_ms;      /* part of solve_ms spent on what the reference's full rebuild per addGraphImpl (:57) implies here:
                                  setFixedNodes, block-CSR structure, aggregation hierarchy, (first solve) PCG graph capture */
    double  exchange_ms;       /* sharded solve: host time inside the exchange step (callback) or enqueueing it (native RCCL) */
    int32_t structure_reused;  /* 1: the structure of the previous graph was kept (same vertices / edge endpoints / fixed flags) */
    int32_t n_eliminated;      /* free vertices Schur-eliminated ahead of the PCG (chain interiors), 0 = full system */
    int32_t lm_passes;         /* device-resident loop: passes enqueued = host looks at the state; 0 = the host-driven loop ran */
    int32_t reduced_strong;    /* 1: the Schur-reduced system was numbered by strong aggregates (uzl_pgo_cfg::reduced_numbering); was reserved0 */
} uzl_pgo_stats;

void uzl_pgo_cfg_default(uzl_pgo_cfg* cfg);

/* G2oOptimizer::G2oOptimizer (g2o_optimizer.cpp:34-49). */
int  uzl_pgo_create(const uzl_pgo_cfg* cfg, uzl_pgo** out);
void uzl_pgo_destroy(uzl_pgo* h);
/* GraphOptimizer::setConfig (graph_optimizer.cpp:54-57). */
int  uzl_pgo_set_config(uzl_pgo* h, const uzl_pgo_cfg* cfg);
const char* uzl_pgo_last_error(uzl_pgo* h);

/* G2oOptimizer::addGraphImpl (g2o_optimizer.cpp:55-104): full rebuild.  Applies the skip rules
 * (:77, :203-206, :270-274), composes the measurements (:229, :281), the optional xy-only
 * projection (:164-170, :231-237, :282-288) and marks non-odometry edges robust (:292-294).
 * sensors: n_sensors x 12 doubles (SlamGraph sensor transforms, :68-71).  Only copies.
 * The poses a solve returns do not depend on what the handle solved before, up to the accuracy of the linear solve (pcg_tol): with
 * reduced_numbering = 0 a handle remembers how many PCG iterations its last solves took in either numbering of the Schur-reduced
 * system and lays the next one out accordingly (another preconditioner for the same system).  That memory is kept while the graph is
 * the previous one, unchanged or grown (at least as many nodes, the old nodes' fixed flags in front, nine in ten of the old system
 * edges still present in their order: an online session, graph_slam_node.cpp:1138-1150), and dropped for any other graph. */
int  uzl_pgo_add_graph(uzl_pgo* h,
                       int32_t n_nodes, const uzl_node* nodes,
                       int32_t n_edges, const uzl_edge* edges,
                       int32_t n_sensors, const double* sensors);

/* The graph of the last uzl_pgo_add_graph / uzl_pgo_append_graph, GROWN in place: n_new_nodes nodes and n_new_edges edges are appended
 * (node ids continue: the first new node is node n_nodes of the graph so far; edges may name any node), and the `valid` flag of the
 * old edges edge_index[0 .. n_flags) is set to edge_valid[.].  An online session re-optimises a graph that gained a few hundred nodes
 * and edges since the last time (graph_slam_node.cpp:1138-1150 -> g2o_optimizer.cpp:55-104 rebuilds it from the SlamGraph every time);
 * with the graph resident in HBM only the new part crosses PCIe and only the new part is flattened.
 * The result is what uzl_pgo_add_graph on the same handle gives for the grown arrays with the old nodes' poses as uzl_pgo_store last returned them - the
 * handle's current estimates, i.e. what storeImpl wrote back into the SlamGraph (:106-135) - and the same skip rules (node `fixed`
 * flags of old nodes stay as given).  Sensors stay as given to uzl_pgo_add_graph.  Only copies. */
int  uzl_pgo_append_graph(uzl_pgo* h,
                          int32_t n_new_nodes, const uzl_node* new_nodes,
                          int32_t n_new_edges, const uzl_edge* new_edges,
                          int32_t n_flags, const int32_t* edge_index, const uint8_t* edge_valid);

/* Already-flattened form of the same problem (what addGraphImpl leaves inside g2o):
 * poses n x 12, fixed n, ij e x 2, meas e x 12, info e x 36, robust e. */
int  uzl_pgo_set_graph(uzl_pgo* h, int32_t n, const double* poses, const uint8_t* fixed,
                       int32_t e, const int32_t* ij, const double* meas, const double* info,
                       const uint8_t* robust);

/* Restore the vertex estimates to what the last add_graph/set_graph left (device-to-device copy):
 * the reference gets the same effect by calling addGraphImpl again (full rebuild, :57); with the
 * graph resident in HBM a repeated solve does not need the upload. */
int  uzl_pgo_reset(uzl_pgo* h);

/* G2oOptimizer::optimizeImpl (g2o_optimizer.cpp:137-149): initializeOptimization, setFixedNodes
 * (:301-349) and optimize(iterations).  iterations <= 0 uses cfg.iterations.  Blocks. */
int  uzl_pgo_optimize(uzl_pgo* h, int32_t iterations, uzl_pgo_stats* stats);

/* G2oOptimizer::storeImpl (g2o_optimizer.cpp:106-135): poses out n x 12 (node order of the last
 * add_graph/set_graph); edge_error out = ||e||_2 per input edge (un-weighted 6-norm, :126),
 * NaN for edges that were skipped; edge_in_system out = 1 for edges that entered the solve.
 * Any of the three may be NULL. */
int  uzl_pgo_store(uzl_pgo* h, double* poses, double* edge_error, uint8_t* edge_in_system);

/* Vertices fixed after the last optimize (input fixed flags + setFixedNodes()), n flags. */
int  uzl_pgo_get_fixed(uzl_pgo* h, uint8_t* fixed);

int  uzl_pgo_set_profiling(uzl_pgo* h, int32_t on);
int  uzl_pgo_kernel_times(uzl_pgo* h, int32_t cap, const char** names, double* ms, int32_t* launches);

/* The Schur plan of uzl_pgo_cfg::schur_reduce for a block structure given as CSR over nb free vertices (col = -1: fixed neighbour):
 * which rows are chain interiors (one or two incident edges, to different neighbours; g2o_optimizer.cpp:190-259 builds that chain),
 * how they group into runs of at most `cap` vertices, and the block structure of the Schur complement over the rest.  Host code
 * only (no device).  red_row / run_id / run_pos: nb entries (-1 where not applicable); red_row_ptr: nb + 1 entries; red_col:
 * cap_slots entries (UZL_ERR_BAD_ARG if the reduced system has more blocks). */
int  uzl_pgo_schur_plan(int32_t nb, const int32_t* row_ptr, const int32_t* col, int32_t cap, int32_t* red_row, int32_t* run_id,
                        int32_t* run_pos, int32_t* red_row_ptr, int32_t* red_col, int32_t cap_slots, int32_t* n_reduced, int32_t* n_runs);
/* The same plan with the reduced system numbered by strong aggregates (uzl_pgo_cfg::reduced_numbering = 2): slot_w = one weight per
 * block of `col` (trace of the edge's information matrix), strong_min = separators from which on the numbering applies, theta = how
 * stiff an edge must be against the stiffest at either end to tie two separators (0.25), one_level_max = groups up to which the layout
 * is one level (256).  red_row: full row -> reduced row (-1: eliminated); sep_rows (cap_rows entries): reduced row -> full row, -1 for
 * an EMPTY padding row; counts[5] = {reduced rows, separators, groups of <= 8, blocks of <= 4 groups, 1000 x the share of separators in
 * groups that are consecutive in row order anyway}.  Host code only. */
int  uzl_pgo_schur_plan_strong(int32_t nb, const int32_t* row_ptr, const int32_t* col, int32_t cap, const double* slot_w, int32_t strong_min,
                               double theta, int32_t one_level_max, int32_t* red_row, int32_t* sep_rows, int32_t cap_rows, int32_t* counts);

/* ---- sharded single-graph solve (BASELINE config 4): one handle per rank ------------------
 * The graph is edge-partitioned: every rank holds all vertices and the edges
 * [e_begin, e_end) of the flattened problem.  The caller supplies the exchange step: a
 * function that sums `count` doubles in place across all ranks (RCCL all-reduce on the
 * device buffer `dev_ptr`, issued on `hip_stream`).  A NULL callback means unsharded. */
typedef int (*uzl_allreduce_fn)(void* dev_ptr, int64_t count, void* hip_stream, void* user);
int  uzl_pgo_set_shard(uzl_pgo* h, int32_t rank, int32_t world_size,
                       uzl_allreduce_fn allreduce, void* user);

/* The same exchange owned by the handle (SURVEY section 8b "Threading": the multi-GPU handle owns its RCCL communicator).
 * One process per GPU; rank 0 creates an id with uzl_rccl_unique_id and hands the bytes to the other ranks by whatever channel the
 * caller has (the reference's ROS parameter server, a file, MPI, torch.distributed ...); then EVERY rank calls
 * uzl_pgo_set_shard_rccl with the same id (collective: returns when all world_size ranks have joined).  The handle then issues
 * ncclAllReduce(sum, f64, in place) on its own HIP stream between its kernels - stream-ordered, no host synchronisation, no
 * callback - and destroys the communicator in uzl_pgo_destroy (or when the shard setting changes).  librccl.so is loaded on the
 * first call only (dlopen), so processes that never shard a graph do not pay for it.
 *   id buffer: UZL_RCCL_UNIQUE_ID_BYTES bytes.  world_size 1 is allowed (every exchange step still runs: a one-GPU test of the path). */
#define UZL_RCCL_UNIQUE_ID_BYTES 128
int  uzl_rccl_unique_id(void* id_out, int32_t cap);
int  uzl_pgo_set_shard_rccl(uzl_pgo* h, int32_t rank, int32_t world_size, const void* unique_id, int32_t id_bytes);
/* Ranks of the handle's communicator as RCCL itself counts them (ncclCommCount): what a multi-GPU run reports next to its numbers to
 * show that the exchange really spans the devices; 0 = no communicator (unsharded, or the callback form), < 0 = error code. */
int  uzl_pgo_rccl_ranks(uzl_pgo* h);

/* ---- batched solve: many small graphs through shared launches ---------------------------
 * A 1k-node graph uses ~3 % of an MI355X (125 workgroups per launch, two dependent launches per PCG iteration).  Independent
 * graphs - the disjoint subgraphs / local scopes / per-robot graphs of SURVEY section 8e row 2, or the disconnected components
 * setFixedNodes() finds (g2o_optimizer.cpp:301-349) given as separate graphs - are therefore solved together: every kernel is
 * launched once for the whole batch (blockIdx.z = graph), the host runs the Levenberg-Marquardt decisions of all graphs in
 * lock step.  Each graph's result (poses, chi2, iteration counts) is bit-identical to uzl_pgo_optimize on that graph alone.
 * The batch owns n_graphs ordinary handles: fill them with uzl_pgo_add_graph / uzl_pgo_set_graph, read them with uzl_pgo_store.
 * Graphs are launched together when they are of the small-graph class (<= 2048 free vertices) and have the same hierarchy shape
 * (same number of free vertices per level, e.g. same-size graphs); otherwise, and for any graph whose solve meets an anomaly, the
 * call falls back to one uzl_pgo_optimize per graph - same results, no batching.  *n_batched = graphs solved in the batch.
 * stats[g].solve_ms of a batched graph is the wall time of the whole batch call.  The graphs' handles must not be used from other
 * threads while uzl_pgo_batch_optimize runs (it drives them without taking their mutexes).  From 12 graphs on the call solves the
 * second half of the graphs as a launch sequence of its own, from a helper thread that it starts and joins before it returns
 * (results per graph are the same either way). */
typedef struct uzl_pgo_batch uzl_pgo_batch;
int  uzl_pgo_batch_create(const uzl_pgo_cfg* cfg, int32_t n_graphs, uzl_pgo_batch** out);
void uzl_pgo_batch_destroy(uzl_pgo_batch* b);
const char* uzl_pgo_batch_last_error(uzl_pgo_batch* b);
int  uzl_pgo_batch_size(uzl_pgo_batch* b);
uzl_pgo* uzl_pgo_batch_graph(uzl_pgo_batch* b, int32_t i);            /* borrowed: destroyed with the batch */
int  uzl_pgo_batch_optimize(uzl_pgo_batch* b, int32_t iterations, uzl_pgo_stats* stats /* n_graphs entries, may be NULL */,
                            int32_t* n_batched /* may be NULL */);
/* How many of the batch's graphs are solved at a time (0 = all of them, the default).  With fewer resident slots than graphs the
 * batch is a queue worked off in cohorts: the resident graphs advance in step (they linearise, rebuild their preconditioners and
 * evaluate in the same launches) and the slots are refilled when all of them are through.  Results per graph do not depend on it. */
int  uzl_pgo_batch_set_resident(uzl_pgo_batch* b, int32_t n_resident);
/* per-kernel timing of the two PCG kernels of the last batch solve (as uzl_pgo_set_profiling / uzl_pgo_kernel_times) */
int  uzl_pgo_batch_set_profiling(uzl_pgo_batch* b, int32_t on);
int  uzl_pgo_batch_kernel_times(uzl_pgo_batch* b, int32_t cap, const char** names, double* ms, int32_t* launches);

/* ======================================================================================
 *  Edge filter  (TransformationFilter / EdgeCluster)
 *
 *  The step between the two halves: every non-odometry edge passes through it before it
 *  reaches the solver (g2o_optimizer.cpp:74-103).  Edges are grouped into clusters by the
 *  time stamps of their end nodes (transformation_filter.cpp:144-207); a cluster that
 *  changed is validated by a 3-point RANSAC over the translations of its edges' world-frame
 *  end poses (:222-291, 200 hypotheses, 0.3 m, no PROSAC prefix); validEdges() (:293-337)
 *  thins each cluster's valid edges.  The cluster bookkeeping is sequential host logic; the
 *  pose chains, the RANSAC and the consensus run on the GPU, batched over all changed
 *  clusters of one calcValidEdges() call.
 *
 *  String ids stay in the adapter: edges are addressed by a caller-chosen 64-bit key.
 *  Where the reference iterates an unordered_map (order unspecified) this build uses
 *  insertion order; equal matching scores keep insertion order (the reference's std::sort
 *  is unstable).  RANSAC stream of a cluster evaluation: job id = cluster_uid * 2^20 +
 *  evaluation counter, keyed with cfg.seed like every other job.
 * ====================================================================================== */

typedef struct uzl_filter uzl_filter;

typedef struct uzl_filter_cfg {
    double  max_dt;               /* 5.0   TransformationFilter(max_dt, ...)  transformation_filter.h:82, g2o_optimizer.cpp:46 */
    double  min_size;             /* 8.0   "cluster_size" ROS parameter (g2o_optimizer.cpp:43-46); header default 10          */
    int32_t max_cluster_size;     /* 100   transformation_filter.h:82                                                          */
    int32_t ransac_iterations;    /* 200   transformation_filter.cpp:273                                                       */
    double  max_error;            /* 0.3   transformation_filter.cpp:272                                                       */
    double  min_time_span;        /* 2.0   seconds, transformation_filter.cpp:240-241                                          */
    int32_t max_edges;            /* 5     validEdges(): transformation_filter.cpp:310                                         */
    int32_t device;
    uint64_t seed;
} uzl_filter_cfg;

void uzl_filter_cfg_default(uzl_filter_cfg* cfg);

/* One SlamEdge with its end nodes as TransformationFilter::add(edge, from, to) sees them
 * (transformation_filter.cpp:138).  3x4 row-major [R|t] like everywhere in this ABI. */
typedef struct uzl_filter_edge {
    uint64_t key;                 /* stands for SlamEdge::id_                                        */
    double   matching_score;      /* SlamEdge::matching_score_                                       */
    int32_t  valid;               /* SlamEdge::valid_ (initial EdgeData::valid_)                     */
    int32_t  sensor_from;         /* index into the sensor table, -1 = identity                      */
    int32_t  sensor_to;
    int32_t  n_stamps_from;       /* SlamNode::stamps_ of the from node                              */
    int32_t  n_stamps_to;
    int32_t  _pad;
    const int64_t* stamps_from_ns;/* ros::Time as nanoseconds                                        */
    const int64_t* stamps_to_ns;
    double   transform[12];       /* SlamEdge::transform_                                            */
    double   displacement_from[12];
    double   displacement_to[12];
    double   pose_from[12];       /* SlamNode::pose_ of the end nodes                                */
    double   pose_to[12];
} uzl_filter_edge;

int  uzl_filter_create(const uzl_filter_cfg* cfg, uzl_filter** out);
void uzl_filter_destroy(uzl_filter* h);
const char* uzl_filter_last_error(uzl_filter* h);

/* sensor_transforms_ (transformation_filter.h:92): n_sensors x 12 doubles. */
int  uzl_filter_set_sensors(uzl_filter* h, int32_t n_sensors, const double* sensors);
/* TransformationFilter::add for each edge in order: a known key only refreshes the stored
 * edge and end poses (:140-146); a new key is clustered by its stamp pairs (:148-206). */
int  uzl_filter_add(uzl_filter* h, int32_t n_edges, const uzl_filter_edge* edges);
/* TransformationFilter::remove (:209-220). */
int  uzl_filter_remove(uzl_filter* h, int32_t n_keys, const uint64_t* keys);
/* TransformationFilter::allEdges (:343-350): keys in ascending order; *n = total count. */
int  uzl_filter_all_edges(uzl_filter* h, int32_t cap, uint64_t* keys, int32_t* n);
/* TransformationFilter::calcValidEdges (:222-291); n_evaluated = clusters sent to the GPU. */
int  uzl_filter_calc_valid_edges(uzl_filter* h, int32_t* n_evaluated);
/* TransformationFilter::validEdges (:293-337): keys in ascending order (std::set order). */
int  uzl_filter_valid_edges(uzl_filter* h, int32_t cap, uint64_t* keys, int32_t* n);

/* Introspection for parity tests: clusters in clusters_ order. */
int  uzl_filter_cluster_count(uzl_filter* h);
typedef struct uzl_cluster_info {
    uint64_t uid;
    int64_t  from_start_ns, from_end_ns, to_start_ns, to_end_ns;
    int32_t  size, consensus, changed, evaluations;
} uzl_cluster_info;
int  uzl_filter_cluster_info(uzl_filter* h, int32_t index, uzl_cluster_info* info);
/* keys and EdgeData::valid_ flags of one cluster, in cluster order; cap >= size. */
int  uzl_filter_cluster_edges(uzl_filter* h, int32_t index, int32_t cap, uint64_t* keys, uint8_t* valid);
/* The P / Q columns (3 x size, column-major) and the transform of the cluster's LAST GPU evaluation. */
int  uzl_filter_cluster_last_eval(uzl_filter* h, int32_t index, int32_t cap, double* P, double* Q, double* T,
                                  int32_t* ransac_consensus);

/* ======================================================================================
 *  Edge acceptance gate  (GraphSlamNode::newEdgeCallback)
 *
 *  The step right after the estimator (graph_slam/src/graph_slam_node.cpp:779-829): an
 *  estimated edge enters the graph only if no edge of its type joins the two nodes yet,
 *  its score reaches min_matching_score, its transform stays within max_edge_distance_T/R
 *  and checkEdgeHeuristic (:1064-1085) finds it plausible: the length of the path that
 *  SlamGraph::astar (slam_graph.cpp:843-890, a greedy best-first search over the valid
 *  edges) finds between the nodes bounds how far apart their current poses may be.
 *  One batch of candidates = one kernel launch, one search per lane; the sequential
 *  semantics of the callback (an accepted edge is in the graph for the next candidate)
 *  are replayed on the host over the search results.  Node / edge ids are indices.
 * ====================================================================================== */

typedef struct uzl_gate uzl_gate;

typedef struct uzl_gate_cfg {
    double  min_matching_score;   /* 20    graph_slam/cfg/GraphSlam.cfg:18            */
    double  max_edge_distance_T;  /* 1.0   m,   GraphSlam.cfg:19                      */
    double  max_edge_distance_R;  /* 20.0  deg, GraphSlam.cfg:20                      */
    double  scope_size_factor;    /* 0.1   GraphSlam.cfg:34                           */
    double  min_accept_valid;     /* DBL_MAX  "min_accept_valid" (graph_slam_node.cpp:139) */
    int32_t device;
    int32_t _pad;
} uzl_gate_cfg;

typedef struct uzl_gate_edge {
    int32_t from, to;             /* node indices (id_from_, id_to_)                  */
    int32_t type;                 /* UZL_EDGE_TYPE_*                                  */
    int32_t valid;                /* SlamEdge::valid_ (graph edges; ignored for candidates) */
    double  matching_score;       /* candidates only                                  */
    double  transform[12];        /* candidates only: transform_                      */
} uzl_gate_edge;

void uzl_gate_cfg_default(uzl_gate_cfg* cfg);
int  uzl_gate_create(const uzl_gate_cfg* cfg, uzl_gate** out);
void uzl_gate_destroy(uzl_gate* h);
const char* uzl_gate_last_error(uzl_gate* h);
/* The graph the callback sees: node poses (n x 12), merged flags (isMerged, may be NULL),
 * existing edges (from, to, type, valid). */
int  uzl_gate_set_graph(uzl_gate* h, int32_t n_nodes, const double* poses, const uint8_t* merged,
                        int32_t n_edges, const uzl_gate_edge* edges);
/* newEdgeCallback for every candidate in order.  accept[k] = the edge was added to the graph,
 * valid[k] = its valid_ flag (score >= min_accept_valid), astar_dist[k] = path length found
 * (-1: the search was not reached, DBL_MAX: target not reachable).  Outputs may be NULL
 * except accept.  Accepted edges stay in the handle's graph.
 * astar_dist == NULL also means that only the verdicts are wanted: checkEdgeHeuristic's tests are
 * monotone in the path length, so a lower bound of it that passes them (the straight line between
 * the nodes, then a shortest-path search that stops at the radius the tests need) decides without
 * the reference's greedy search; that one runs only for candidates no bound settles.  The verdicts
 * are the reference's either way. */
int  uzl_gate_check(uzl_gate* h, int32_t n_candidates, const uzl_gate_edge* candidates,
                    uint8_t* accept, uint8_t* valid, double* astar_dist);
int  uzl_gate_edge_count(uzl_gate* h);
/* Introspection for parity tests: searches run so far by the reference's greedy search (one wave per candidate) and by the
 * deciding lane-per-candidate search of the verdicts-only form.  Either may be NULL. */
int  uzl_gate_search_counts(uzl_gate* h, int64_t* n_wave, int64_t* n_lane);

/* ======================================================================================
 *  Distance loop-closure candidates  
 *
 *  The step before the estimator: SlamGraph::getNodesWithinRadius (slam_graph.cpp:266-278)
 *  and the filters of its caller (graph_slam_node.cpp:272-289) turn a new node into the list
 *  of (close node, new node) pairs handed to estimateEdge.  Here for a batch of query nodes:
 *  one workgroup per query scans all node positions (an HBM-bound streaming scan) and appends
 *  its hits in node order.  Output jobs are ordered by query, then by node index.
 * ====================================================================================== */
typedef struct uzl_radius uzl_radius;
typedef struct uzl_radius_cfg {
    double  radius;               /* 0.5   distance_loop_closure_radius, GraphSlam.cfg:15 */
    double  new_edge_time;        /* 5.0   s, GraphSlam.cfg:21                             */
    double  max_rotation_deg;     /* 30.0  graph_slam_node.cpp:282                        */
    int32_t device;
    int32_t _pad;
} uzl_radius_cfg;
void uzl_radius_cfg_default(uzl_radius_cfg* cfg);
int  uzl_radius_create(const uzl_radius_cfg* cfg, uzl_radius** out);
void uzl_radius_destroy(uzl_radius* h);
const char* uzl_radius_last_error(uzl_radius* h);
/* node poses (n x 12) and stamps_.front() of every node in nanoseconds */
int  uzl_radius_set_nodes(uzl_radius* h, int32_t n_nodes, const double* poses, const int64_t* stamp_front_ns);
/* jobs (from = close node, to = query node) for every query node; *n_jobs = total found (may exceed cap:
 * then only the first cap are written); count_per_query may be NULL. */
int  uzl_radius_query(uzl_radius* h, int32_t n_queries, const int32_t* query_nodes, int64_t cap,
                      int32_t* out_from, int32_t* out_to, int32_t* count_per_query, int64_t* n_jobs);

/* ======================================================================================
 *  Appearance-based candidate pairs  (second half)
 *
 *  LshSetRecognizer / FastLshSet (place_recognition/src/lsh_set_recognizer.cpp:46-310) behind
 *  PlaceRecognizer (place_recognizer.cpp:71-215): every place's binary descriptors are cut into
 *  key_width-byte keys (one exact-match table per byte offset 0, kw, 2 kw, ... below 32); a
 *  query counts, per earlier place, how many (descriptor, table) keys it shares; places whose
 *  count / tables reaches T are neighbours, best first, subject to a time gap, a k-nearest
 *  cut and a reported-once filter.  Here the tables are open-addressing hash tables in HBM
 *  with per-key entry lists in an append-only arena; a query is two launches (count, insert),
 *  one lane per (descriptor, table).  Counts are integers: results equal the CPU checker's.
 * ====================================================================================== */
typedef struct uzl_places uzl_places;
typedef struct uzl_places_cfg {
    int32_t key_width;            /* 8     FastLshSet(key_width = 8), lsh_set_recognizer.h:66           */
    int32_t min_rows_to_add;      /* 150   a frame is indexed only with more rows (:66, :111)            */
    double  T;                    /* 10    cfg/PlaceRecognizer.cfg "T": minimum count / tables           */
    int32_t k_nearest_neighbors;  /* 10    cfg "k_nearest_neighbors"                                     */
    int32_t device;
    double  min_time_gap;         /* 5.0   s, place_recognizer.cpp:90                                    */
} uzl_places_cfg;
void uzl_places_cfg_default(uzl_places_cfg* cfg);
int  uzl_places_create(const uzl_places_cfg* cfg, uzl_places** out);
void uzl_places_destroy(uzl_places* h);
const char* uzl_places_last_error(uzl_places* h);
/* PlaceRecognizer::searchAndAddPlace: desc = rows x bytes (bytes >= 32) descriptors of the node's FeatureData,
 * stamp = node.stamps_.front().  neighbors (capacity cap) receives the place indices, *n_neighbors their number,
 * *place_index the index given to this place. */
int  uzl_places_search_and_add(uzl_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns,
                               int32_t cap, int32_t* neighbors, int32_t* n_neighbors, int32_t* place_index);
/* PlaceRecognizer::addPlace */
int  uzl_places_add(uzl_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns, int32_t* place_index);
/* PlaceRecognizer::searchPlace; query_place = the querying node's place index (for the reported-once filter), -1 if none */
int  uzl_places_search(uzl_places* h, const uint8_t* desc, int32_t rows, int32_t bytes, int64_t stamp_ns, int32_t query_place,
                       int32_t cap, int32_t* neighbors, int32_t* n_neighbors);
/* PlaceRecognizer::removePlace: needs the descriptors the place was added with (as the reference does) */
int  uzl_places_remove(uzl_places* h, int32_t place_index, const uint8_t* desc, int32_t rows, int32_t bytes);
int  uzl_places_count(uzl_places* h);
/* collision counts per place of the last search / search_and_add (parity tests); returns their number */
int  uzl_places_last_counts(uzl_places* h, int32_t cap, int32_t* counts);

/* ======================================================================================
 *  Wire and disk formats  
 *
 *  The data formats either side of the path: graph_slam_msgs/{Edge,Node,SensorData,Features,
 *  Feature}.msg in ROS 1 serialisation (little-endian; string = u32 length + bytes; T[] = u32 count
 *  + elements; T[N] = elements; time / duration = two 32-bit words; bool = one byte), converted
 *  to and from the graph objects as Conversions does (graph_slam_common/src/conversions.cpp:43-70,
 *  217-322) and FeatureData::toMsg / fromMsg do (graph_slam_common/src/sensor_data.cpp:78-167),
 *  and the one-message-per-file rosbag 2.0 container RosbagStorage writes and reads
 *  (graph_slam_common/src/rosbag_storage.cpp:62-209).
 *
 *  Message headers and strings are host work (a few dozen fields).  The bulk of a stored graph is
 *  the Feature[] arrays - descriptors travel as one float32 per descriptor BYTE
 *  (sensor_data.cpp:93-110), 41 + 4 D bytes per keypoint on the wire for 25 + D bytes of content -
 *  and those are unpacked / packed on the device, straight into / out of the estimator's frame
 *  arena: an HBM-bound byte shuffle, one launch for any number of frames.
 *
 *  Poses: toMsg writes position + Eigen's Quaterniond(R) as (x,y,z,w), un-normalised sign
 *  (conversions.cpp:57-70); fromMsg is g2o::internal::fromVectorQT = Quaterniond(w,x,y,z)
 *  .toRotationMatrix() without normalisation (conversions.cpp:229-240,
 *  graph_slam_common/thirdparty/src/isometry3d_mappings.cpp:131-136).
 * ====================================================================================== */
#define UZL_ERR_TRUNCATED   -9    /* message / file ends inside a field, or output capacity too small */
#define UZL_ERR_UNSUPPORTED -10   /* e.g. compressed rosbag chunk, ragged descriptor lengths           */

/* sensor types: graph_slam_msgs/msg/SensorData.msg:2-6 */
#define UZL_SENSOR_TYPE_UNKNOWN     0
#define UZL_SENSOR_TYPE_FEATURE     1
#define UZL_SENSOR_TYPE_DEPTH_IMAGE 2
#define UZL_SENSOR_TYPE_BINARY_GIST 3
#define UZL_SENSOR_TYPE_LASERSCAN   4

/* borrowed bytes (a string or a sub-message); not NUL-terminated */
typedef struct uzl_span { const char* p; uint64_t n; } uzl_span;

/* graph_slam_msgs/Edge <-> SlamEdge  (Conversions::fromMsg / toMsg, conversions.cpp:242-274) */
typedef struct uzl_wire_edge {
    uzl_span id, id_from, id_to, sensor_from, sensor_to;
    int32_t  type;                 /* uint8 on the wire                                   */
    int32_t  valid;                /* bool on the wire                                    */
    double   transform[12];        /* transformation.pose                                 */
    double   information[36];      /* transformation.covariance, row-major (:48-52,:221-226) */
    double   displacement_from[12], displacement_to[12];
    double   error, age, matching_score;
    int32_t  diff_time_sec, diff_time_nsec;   /* ros::Duration                            */
} uzl_wire_edge;
/* bytes uzl_wire_edge_encode will write */
uint64_t uzl_wire_edge_size(const uzl_wire_edge* e);
int  uzl_wire_edge_encode(const uzl_wire_edge* e, uint8_t* buf, uint64_t cap, uint64_t* written);
/* spans of *out point into buf */
int  uzl_wire_edge_decode(const uint8_t* buf, uint64_t len, uzl_wire_edge* out, uint64_t* consumed);

/* One graph_slam_msgs/SensorData inside a Node message.  Decode fills every field; `raw` is the
 * whole sub-message (copy-through for sensor types this back end does not touch).  Encode: when
 * raw.p != NULL the bytes are copied verbatim, otherwise a FEATURE message is written from the
 * fields below with records = n_features Feature records (uzl_match_frame_to_wire or
 * uzl_wire_features_pack) and camera_info (raw sensor_msgs/CameraInfo bytes; NULL = a
 * default-constructed one); depth_image / gist / scan are written empty as SensorData::toMsg
 * leaves them (sensor_data.cpp:40-49). */
typedef struct uzl_wire_sensor {
    uzl_span raw;
    int32_t  sensor_type;
    uint32_t stamp_sec, stamp_nsec;   /* header.stamp = SensorData::stamp_                               */
    uzl_span sensor_frame;            /* header.frame_id: fromMsg takes sensor_frame_ from here (:56)   */
    double   displacement[12];
    int32_t  descriptor_type;         /* features.descriptor_type = FeatureData::feature_type_          */
    int32_t  n_features;
    int32_t  desc_len;                /* descriptor elements of the first feature (= bytes per row)     */
    int32_t  uniform;                 /* 1 iff every record has desc_len elements (constant stride)     */
    uzl_span records;                 /* the n_features Feature records, 41 + 4 desc_len bytes each     */
    uzl_span camera_info;             /* features.camera_model                                          */
} uzl_wire_sensor;

/* graph_slam_msgs/Node <-> SlamNode  (Conversions::fromMsg / toMsg, conversions.cpp:276-322) */
typedef struct uzl_wire_node {
    uzl_span id;
    double   pose[12], odom_pose[12];     /* SlamNode::pose_, sub_pose_                    */
    int32_t  fixed;
    int32_t  n_stamps, n_edge_ids, n_sensors;
    double   uncertainty;
} uzl_wire_node;
/* Variable parts go to caller arrays: stamps (ns since epoch), edge ids, sensors; counts are always reported in
 * *out, entries beyond a capacity are parsed but not stored. */
int  uzl_wire_node_decode(const uint8_t* buf, uint64_t len, uzl_wire_node* out,
                          int32_t stamp_cap, int64_t* stamps_ns, int32_t edge_cap, uzl_span* edge_ids,
                          int32_t sensor_cap, uzl_wire_sensor* sensors, uint64_t* consumed);
uint64_t uzl_wire_node_size(const uzl_wire_node* n, const uzl_span* edge_ids, const uzl_wire_sensor* sensors);
int  uzl_wire_node_encode(const uzl_wire_node* n, const int64_t* stamps_ns, const uzl_span* edge_ids,
                          const uzl_wire_sensor* sensors, uint8_t* buf, uint64_t cap, uint64_t* written);

/* graph_slam_msgs/GraphMeta <-> the graph's meta data: SlamGraph::toMetaData / updateMetaData
 * (graph_slam_common/src/slam_graph.cpp:592-633), written by RosbagStorage::storeMetaData
 * (graph_slam_common/src/rosbag_storage.cpp:94-107, file <path>/meta/meta, topic "meta") and read back by loadGraph (:187-207).
 * Field order of GraphMeta.msg: header, name, map_transform, sensor_transforms[], sensor_transforms_initial[],
 * odometry_parameters[6]; a SensorTransform is (string sensor_name, geometry_msgs/Pose transform).  The sensor
 * transforms and odometry parameters are exactly what G2oOptimizer::addGraphImpl takes from the graph
 * (graph_optimization/src/g2o_optimizer.cpp:209-227,281), i.e. uzl_pgo_add_graph's sensor table. */
typedef struct uzl_wire_sensor_transform {
    uzl_span sensor_name;
    double   transform[12];
} uzl_wire_sensor_transform;
typedef struct uzl_wire_meta {
    uint32_t stamp_sec, stamp_nsec;       /* header.stamp (header.seq is written 0, ignored on decode)  */
    uzl_span frame_id;                    /* header.frame_id = SlamGraph::frame_                        */
    uzl_span name;                        /* SlamGraph::name_                                           */
    double   map_transform[12];           /* /map -> /base_footprint at store time; sub_transform_ on load */
    int32_t  n_sensor_transforms, n_sensor_transforms_initial;
    double   odometry_parameters[6];
} uzl_wire_meta;
uint64_t uzl_wire_meta_size(const uzl_wire_meta* m, const uzl_wire_sensor_transform* sensor_transforms,
                            const uzl_wire_sensor_transform* sensor_transforms_initial);
int  uzl_wire_meta_encode(const uzl_wire_meta* m, const uzl_wire_sensor_transform* sensor_transforms,
                          const uzl_wire_sensor_transform* sensor_transforms_initial, uint8_t* buf, uint64_t cap,
                          uint64_t* written);
/* Counts are always reported in *out, entries beyond a capacity are parsed but not stored; spans point into buf. */
int  uzl_wire_meta_decode(const uint8_t* buf, uint64_t len, uzl_wire_meta* out, int32_t cap,
                          uzl_wire_sensor_transform* sensor_transforms, int32_t cap_initial,
                          uzl_wire_sensor_transform* sensor_transforms_initial, uint64_t* consumed);

/* bytes of n Feature records with desc_len descriptor elements each */
uint64_t uzl_wire_features_size(int32_t n, int32_t desc_len);

/* FeatureData::fromMsg (sensor_data.cpp:123-167) on the device for a batch of frames: the Feature records of
 * frame k (sensors[k].records, n_features, desc_len, descriptor_type; must be uniform and a binary descriptor
 * type) are uploaded as they are and unpacked by one kernel into the frame arena: descriptor byte =
 * (unsigned char) of the float (truncation, low 8 bits of the integer), position, is_3d.  sensor_frame_keys[k]
 * stands for the sensor_frame_ string as in uzl_frame.  uv (optional, 2 x n_features int32 per frame,
 * concatenated) receives u,v (feature_positions_2d_). */
int  uzl_match_add_frames_wire(uzl_match* h, int32_t n_frames, const uzl_wire_sensor* sensors,
                               const int32_t* sensor_frame_keys, int32_t* frame_ids, int32_t* uv);
/* FeatureData::toMsg (sensor_data.cpp:78-121) on the device: Feature records of a resident frame
 * (keypoint_strength = -1 as :96; uv = 2 x n int32 or NULL for zeros). */
int  uzl_match_frame_to_wire(uzl_match* h, int32_t frame_id, const int32_t* uv, uint8_t* records, uint64_t cap,
                             uint64_t* written);
/* The arena content of a frame (parity tests; desc n x bytes, pos 3 x n, valid n; any may be NULL). */
int  uzl_match_get_frame(uzl_match* h, int32_t frame_id, uint8_t* desc, double* pos_xyz, uint8_t* valid3d,
                         int32_t* n, int32_t* bytes_per_desc);

/* ---- rosbag 2.0, as RosbagStorage uses it: one message per file (rosbag_storage.cpp:62-107) ---- */
typedef struct uzl_bag_msg {
    uzl_span topic, datatype, md5sum, definition;   /* from the message's connection record */
    uzl_span data;                                  /* the serialised message                 */
    uint32_t time_sec, time_nsec;                   /* the record's time                      */
} uzl_bag_msg;
/* Every message-data record of an uncompressed bag image, in file order; *n_msgs = number found (may exceed cap). */
int  uzl_bag_read(const uint8_t* file, uint64_t len, int32_t cap, uzl_bag_msg* msgs, int32_t* n_msgs);
/* bag.open(Write); bag.write(topic, time, msg); bag.close(): header (4096-byte padded), one chunk with the
 * connection and the message, its index record, the connection and chunk-info records.  md5sum / definition
 * are ros::message_traits::{MD5Sum,Definition}<M>::value() of the caller's message type. */
uint64_t uzl_bag_single_size(const uzl_bag_msg* m);
int  uzl_bag_write_single(const uzl_bag_msg* m, uint8_t* out, uint64_t cap, uint64_t* written);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* UZL_MI355X_H */
