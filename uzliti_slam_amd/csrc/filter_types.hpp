// filter_types.hpp — POD shared by filter_kernels.hip and uzl_filter.hip
#pragma once
#include <cstdint>
#include "../../include/uzl_mi355x.h"

namespace uzl {

constexpr int kFilterBlock = 64;      // one wave per workgroup: a batch is a few hundred edges, spread them over CUs

// what filter_points_kernel needs of one EdgeData (transformation_filter.h:28-46): 3x4 row-major [R|t]
struct FilterEdgeDev {
    double pos_from[12], disp_from[12], transform[12], pos_to[12], disp_to[12];
    int32_t sensor_from, sensor_to;
};

}  // namespace uzl
