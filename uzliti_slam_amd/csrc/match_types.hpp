// match_types.hpp — structs shared by the host side and the HIP kernels of the edge-estimation half.
#pragma once
#include <cstdint>
#include "../../include/uzl_mi355x.h"

namespace uzl {

// One (FeatureData_from, FeatureData_to) candidate of a job: the body of the double loop at
// feature_transformation_estimator.cpp:40-91.  Offsets address the frame arena (one HBM allocation).
struct Combo {
    uint64_t desc_from_off;   // u32-word offset of the train descriptors  (from)
    uint64_t desc_to_off;     // u32-word offset of the query descriptors  (to)
    uint64_t pos_from_off;    // double offset of from.feature_positions_ (3 x n col-major)
    uint64_t pos_to_off;
    uint64_t valid_from_off;  // byte offset of from.valid_3d_
    uint64_t valid_to_off;
    int32_t  nt;              // train rows (from)
    int32_t  nq;              // query rows (to)
    int32_t  words;           // u32 words per descriptor
    int32_t  knn_off;         // offset (in uint2) of this combo's 2-NN output
    int32_t  frame_from, frame_to;
};

struct Job {
    uint64_t job_id;
    int32_t  combo_begin, combo_count;
    int32_t  pq_off;          // column offset into caller-supplied P/Q (uzl_ransac_points mode)
    int32_t  pq_count;        // number of columns (uzl_ransac_points mode)
};

struct RansacParams {
    double   thresh;
    double   break_pct;
    uint64_t seed;
    int32_t  iterations;
    int32_t  do_prosac;
    int32_t  max_corr;        // stride of the per-job diagnostic / scratch arrays
    int32_t  lds_points;      // capacity (points) of the LDS-resident correspondence tile
};

// keys: (distance << 20) | index   — distance <= 4095 bits, index < 2^20
constexpr int      kIdxBits = 20;
constexpr uint32_t kIdxMask = (1u << kIdxBits) - 1u;
constexpr int      kMaxKeypoints = 16384;   // LDS sort capacity per frame
constexpr int      kMaxIterations = 4096;   // LDS vote array capacity

struct EstimateArgs {
    const uint8_t* arena;          // frame arena (bytes)
    const Combo* combos;
    const Job* jobs;
    const uint2* knn;
    RansacParams prm;
    uzl_edge_result* results;      // device copy, one per job
    // caller-supplied correspondences (uzl_ransac_points mode), 3 x total column-major; null otherwise
    const double* P_in;
    const double* Q_in;
    // optional diagnostics, stride prm.max_corr per job (may be null)
    int32_t* corr_query;
    int32_t* corr_train;
    int32_t* corr_dist;
    uint8_t* inlier_mask;
    // global scratch for correspondence tiles larger than the LDS tile (may be null if never needed)
    double* pq_scratch;            // n_jobs x max_corr x 6
    double* dist_scratch;          // n_jobs x max_corr
    uint8_t* mask_scratch;         // n_jobs x max_corr
    int32_t sort_cap;              // power of two >= max nq in the batch (LDS sort array length)
    int32_t vote_valu;             // 1: consensus votes on the vector ALU (A/B switch); 0: on the f64 matrix cores
};

}  // namespace uzl
