// rosbag_storage.h — host-side mirror of RosbagStorage (graph_slam_common/include/graph_slam_common/rosbag_storage.h,
// src/rosbag_storage.cpp:36-235) over the C ABI: one rosbag 2.0 file per node / edge under <path>/nodes/<id> and
// <path>/edges/<id>, the graph's meta data under <path>/meta/meta.  Message bytes come from uzl_wire_* (Conversions::toMsg / fromMsg), the Feature[] payload of a node is
// packed / unpacked on the device through the estimator handle the storage is given.
#pragma once
#include <mutex>
#include <string>

#include "../../include/uzl_mi355x.h"
#include "slam_types.h"

namespace uzl_adapter {

class RosbagStorage {
public:
    // message traits of the caller's ROS build (ros::message_traits::MD5Sum<M>::value(), Definition<M>::value())
    struct Traits { std::string node_md5 = "*", node_def, edge_md5 = "*", edge_def, meta_md5 = "*", meta_def; };

    RosbagStorage(uzl_match* estimator, const std::string& storage_path, bool clear_storage = false);   // :36-44
    void setTraits(const Traits& t) { traits_ = t; }
    void clear();                                                                                       // :54-60
    bool storeNode(const SlamNode& node, int64_t now_ns = 0);                                           // :62-76
    bool storeEdge(const SlamEdge& edge, int64_t now_ns = 0);                                           // :78-92
    // the graph's meta data (sensor transforms, odometry parameters, name, frame) into <path>/meta/meta; map_pose = /map -> /base_footprint
    // as the caller's tf listener has it (toMetaData asks tf: slam_graph.cpp:599-602)
    bool storeMetaData(SlamGraph& graph, const Isometry3d& map_pose = Isometry3d::Identity(), int64_t now_ns = 0);   // :94-107
    void removeNode(const std::string& id);                                                             // :110-122
    void removeEdge(const std::string& id);                                                             // :124-136
    // nodes' FeatureData arrays are filled from the device-side unpack; returns false if a file could not be parsed
    bool loadGraph(SlamGraph& graph);                                                                   // :138-209
    const std::string& lastError() const { return last_error_; }

private:
    void initialize(const std::string& storage_path, bool clear_storage);                               // :211-235
    bool writeBag(const std::string& file, const char* topic, const char* type, const std::string& md5, const std::string& def,
                  int64_t now_ns, const std::vector<uint8_t>& msg);
    uzl_match* estimator_;
    std::string storage_path_, last_error_;
    Traits traits_;
    std::mutex rosbag_mutex_;
};

}  // namespace uzl_adapter
