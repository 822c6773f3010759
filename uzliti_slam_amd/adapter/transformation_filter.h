// transformation_filter.h — host-side mirror of the reference's TransformationFilter
//   transformation_estimation/include/transformation_estimation/transformation_filter.h:80-106
// over the uzl_filter_* C ABI.  Same public methods and meaning; string ids are mapped to the ABI's 64-bit keys
// here, and the SlamEdge copies validEdges() hands back are kept here too.
#pragma once
#include <map>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/uzl_mi355x.h"
#include "slam_types.h"

namespace uzl_adapter {

class TransformationFilter {
public:
    TransformationFilter(double max_dt = 5., int min_size = 10, int max_cluster_size = 100, int device = 0, uint64_t seed = 0);
    ~TransformationFilter();
    TransformationFilter(const TransformationFilter&) = delete;
    TransformationFilter& operator=(const TransformationFilter&) = delete;

    void add(const SlamEdge& edge, const SlamNode& from, const SlamNode& to);   // transformation_filter.cpp:138-207
    void remove(std::string id);                                                 // :209-220
    void calcValidEdges();                                                       // :222-291 (GPU)
    std::vector<SlamEdge> validEdges(int skip = 1);                              // :293-337, in id order
    std::unordered_set<std::string> allEdges();                                  // :343-350

    std::map<std::string, Isometry3d> sensor_transforms_;                        // transformation_filter.h:92

    int lastStatus() const { return status_; }
    int lastEvaluated() const { return evaluated_; }

private:
    int32_t sensorIndex(const std::string& name);

    uzl_filter* h_ = nullptr;
    int status_ = 0, evaluated_ = 0;
    uint64_t next_key_ = 1;
    std::unordered_map<std::string, uint64_t> key_of_;
    std::unordered_map<uint64_t, SlamEdge> edge_of_;
    std::map<std::string, int32_t> sensor_index_;                                // grows only: indices stay valid
};

}  // namespace uzl_adapter
