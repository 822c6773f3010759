// transformation_filter.cpp — see transformation_filter.h
#include "transformation_filter.h"

#include <algorithm>
#include <cstring>

namespace uzl_adapter {

TransformationFilter::TransformationFilter(double max_dt, int min_size, int max_cluster_size, int device, uint64_t seed)
{
    uzl_filter_cfg c;
    uzl_filter_cfg_default(&c);
    c.max_dt = max_dt; c.min_size = min_size; c.max_cluster_size = max_cluster_size; c.device = device; c.seed = seed;
    status_ = uzl_filter_create(&c, &h_);
}

TransformationFilter::~TransformationFilter()
{
    if (h_) uzl_filter_destroy(h_);
}

int32_t TransformationFilter::sensorIndex(const std::string& name)
{
    auto it = sensor_index_.find(name);
    if (it != sensor_index_.end()) return it->second;
    if (sensor_transforms_.find(name) == sensor_transforms_.end()) return -1;    // unknown sensor: identity
    const int32_t idx = (int32_t)sensor_index_.size();
    sensor_index_[name] = idx;
    return idx;
}

void TransformationFilter::add(const SlamEdge& edge, const SlamNode& from, const SlamNode& to)
{
    if (!h_) return;
    auto k = key_of_.find(edge.id_);
    const bool known = k != key_of_.end();
    const uint64_t key = known ? k->second : next_key_;
    uzl_filter_edge e;
    std::memset(&e, 0, sizeof(e));
    e.key = key;
    e.matching_score = edge.matching_score_;
    e.valid = edge.valid_ ? 1 : 0;
    e.sensor_from = sensorIndex(edge.sensor_from_);
    e.sensor_to = sensorIndex(edge.sensor_to_);
    e.n_stamps_from = (int32_t)from.stamps_.size(); e.stamps_from_ns = from.stamps_.data();
    e.n_stamps_to = (int32_t)to.stamps_.size(); e.stamps_to_ns = to.stamps_.data();
    std::memcpy(e.transform, edge.transform_.m.data(), 96);
    std::memcpy(e.displacement_from, edge.displacement_from_.m.data(), 96);
    std::memcpy(e.displacement_to, edge.displacement_to_.m.data(), 96);
    std::memcpy(e.pose_from, from.pose_.m.data(), 96);
    std::memcpy(e.pose_to, to.pose_.m.data(), 96);
    status_ = uzl_filter_add(h_, 1, &e);
    if (status_ != UZL_OK) return;
    if (!known) {
        // an edge between nodes without stamps is never clustered and never known to the filter (:148-149)
        if (from.stamps_.empty() || to.stamps_.empty()) return;
        key_of_[edge.id_] = key;
        next_key_++;
    }
    edge_of_[key] = edge;                                   // EdgeData::edge_ = edge (:55, :72, :85)
}

void TransformationFilter::remove(std::string id)
{
    auto k = key_of_.find(id);
    if (!h_ || k == key_of_.end()) return;
    const uint64_t key = k->second;
    status_ = uzl_filter_remove(h_, 1, &key);
    edge_of_.erase(key);
    key_of_.erase(k);
}

void TransformationFilter::calcValidEdges()
{
    if (!h_) return;
    std::vector<double> table(12 * std::max<size_t>(sensor_index_.size(), 1), 0.0);
    for (const auto& s : sensor_index_) {
        auto t = sensor_transforms_.find(s.first);
        const Isometry3d T = t == sensor_transforms_.end() ? Isometry3d::Identity() : t->second;
        std::memcpy(&table[12 * (size_t)s.second], T.m.data(), 96);
    }
    status_ = uzl_filter_set_sensors(h_, (int32_t)sensor_index_.size(), table.data());
    if (status_ != UZL_OK) return;
    int32_t n = 0;
    status_ = uzl_filter_calc_valid_edges(h_, &n);
    evaluated_ = n;
}

std::vector<SlamEdge> TransformationFilter::validEdges(int)
{
    std::vector<SlamEdge> res;
    if (!h_) return res;
    int32_t n = 0;
    std::vector<uint64_t> keys(edge_of_.size() + 1);
    status_ = uzl_filter_valid_edges(h_, (int32_t)keys.size(), keys.data(), &n);
    if (status_ != UZL_OK) return res;
    for (int32_t i = 0; i < n; i++) {
        auto e = edge_of_.find(keys[i]);
        if (e != edge_of_.end()) res.push_back(e->second);
    }
    // std::set<std::string> order (:295, :335-337)
    std::sort(res.begin(), res.end(), [](const SlamEdge& a, const SlamEdge& b) { return a.id_ < b.id_; });
    return res;
}

std::unordered_set<std::string> TransformationFilter::allEdges()
{
    std::unordered_set<std::string> all;
    for (const auto& k : key_of_) all.insert(k.first);
    return all;
}

}  // namespace uzl_adapter
