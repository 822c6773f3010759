// Internal (non-ABI) interface between the estimator handle and the edge filter, which owns one the way
// TransformationFilter owns a FeatureTransformationEstimator (transformation_filter.h:105).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/uzl_mi355x.h"

namespace uzl {

struct MatchDeviceResults {
    const uzl_edge_result* results;   // device, one per problem
    const uint8_t* mask;              // device, row `stride` per problem
    int stride;
    hipStream_t stream;               // everything above is ordered on this stream
};

hipStream_t match_stream(uzl_match* h);

// 3-point RANSAC over point sets resident on the device (3 x total column-major, problem b = columns
// [offsets[b], offsets[b+1])); enqueue only - the caller synchronises on out->stream.
int match_ransac_device(uzl_match* h, int32_t n_problems, const int32_t* offsets, const double* dP, const double* dQ,
                        double max_error, int32_t iterations, double break_percentage, int32_t do_prosac,
                        const uint64_t* job_ids, MatchDeviceResults* out);

}  // namespace uzl
