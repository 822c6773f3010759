// rosbag_storage.cpp — see rosbag_storage.h.  std::filesystem stands in for boost::filesystem.
#include "rosbag_storage.h"

#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>

namespace fs = std::filesystem;

namespace uzl_adapter {

namespace {

uzl_span span_of(const std::string& s) { return uzl_span{s.data(), s.size()}; }
std::string str_of(const uzl_span& s) { return s.p ? std::string(s.p, s.n) : std::string(); }

bool read_file(const fs::path& p, std::vector<uint8_t>& out)
{
    std::ifstream f(p, std::ios::binary);
    if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

}  // namespace

RosbagStorage::RosbagStorage(uzl_match* estimator, const std::string& storage_path, bool clear_storage)
    : estimator_(estimator), storage_path_(storage_path)
{
    initialize(storage_path_, clear_storage);
}

void RosbagStorage::initialize(const std::string& storage_path, bool clear_storage)
{
    fs::path dir(storage_path);
    if (clear_storage && fs::is_directory(dir)) fs::remove_all(dir);
    fs::create_directories(dir / "nodes");
    fs::create_directories(dir / "edges");
    fs::create_directories(dir / "meta");
}

void RosbagStorage::clear()
{
    std::lock_guard<std::mutex> lock(rosbag_mutex_);
    initialize(storage_path_, true);
}

bool RosbagStorage::writeBag(const std::string& file, const char* topic, const char* type, const std::string& md5,
                             const std::string& def, int64_t now_ns, const std::vector<uint8_t>& msg)
{
    const int64_t t = now_ns + 1;                                      // ros::Time::now() + ros::Duration(0, 1)
    uzl_bag_msg m;
    m.topic = uzl_span{topic, strlen(topic)}; m.datatype = uzl_span{type, strlen(type)};
    m.md5sum = span_of(md5); m.definition = span_of(def);
    m.data = uzl_span{reinterpret_cast<const char*>(msg.data()), msg.size()};
    m.time_sec = (uint32_t)(t / 1000000000); m.time_nsec = (uint32_t)(t % 1000000000);
    std::vector<uint8_t> img(uzl_bag_single_size(&m));
    uint64_t w = 0;
    if (uzl_bag_write_single(&m, img.data(), img.size(), &w) != UZL_OK) { last_error_ = "bag_write_single failed"; return false; }
    std::ofstream f(file, std::ios::binary | std::ios::trunc);
    f.write(reinterpret_cast<const char*>(img.data()), (std::streamsize)w);
    return (bool)f;
}

bool RosbagStorage::storeNode(const SlamNode& node, int64_t now_ns)
{
    std::lock_guard<std::mutex> lock(rosbag_mutex_);
    // Conversions::toMsg(node): every FeatureData becomes a SensorData message whose Feature records are packed on the device
    std::vector<uzl_wire_sensor> sens;
    std::vector<std::vector<uint8_t>> records;
    std::vector<std::vector<double>> pos_cm;
    for (const SensorDataPtr& sd : node.sensor_data_) {
        const FeatureData* fd = dynamic_cast<const FeatureData*>(sd.get());
        if (!fd) continue;                                             // other sensor types: not carried by this stand-in data model
        uzl_frame f;
        memset(&f, 0, sizeof(f));
        std::vector<uint8_t> valid(fd->valid_3d_.begin(), fd->valid_3d_.end());
        f.desc = fd->features_.data(); f.n = fd->rows; f.bytes_per_desc = fd->bytes_per_row;
        f.pos_xyz = fd->feature_positions_.data(); f.valid3d = valid.data(); f.feature_type = fd->feature_type_;
        records.emplace_back(uzl_wire_features_size(fd->rows, fd->bytes_per_row));
        if (fd->rows > 0) {
            int32_t fid = -1;
            uint64_t w = 0;
            if (uzl_match_add_frame(estimator_, &f, &fid) != UZL_OK ||
                uzl_match_frame_to_wire(estimator_, fid, fd->feature_positions_2d_.empty() ? nullptr : fd->feature_positions_2d_.data(),
                                        records.back().data(), records.back().size(), &w) != UZL_OK) {
                last_error_ = uzl_match_last_error(estimator_);
                return false;
            }
            uzl_match_remove_frame(estimator_, fid);
        }
        uzl_wire_sensor s;
        memset(&s, 0, sizeof(s));
        s.sensor_type = UZL_SENSOR_TYPE_FEATURE;
        s.stamp_sec = (uint32_t)(fd->stamp_ / 1000000000); s.stamp_nsec = (uint32_t)(fd->stamp_ % 1000000000);
        s.sensor_frame = span_of(fd->sensor_frame_);
        memcpy(s.displacement, fd->displacement_.m.data(), 96);
        s.descriptor_type = fd->feature_type_; s.n_features = fd->rows; s.desc_len = fd->rows ? fd->bytes_per_row : 0; s.uniform = 1;
        s.records = uzl_span{reinterpret_cast<const char*>(records.back().data()), records.back().size()};
        sens.push_back(s);
    }
    uzl_wire_node n;
    memset(&n, 0, sizeof(n));
    n.id = span_of(node.id_);
    memcpy(n.pose, node.pose_.m.data(), 96); memcpy(n.odom_pose, node.sub_pose_.m.data(), 96);
    n.fixed = node.fixed_; n.uncertainty = node.uncertainty_;
    n.n_stamps = (int32_t)node.stamps_.size(); n.n_sensors = (int32_t)sens.size(); n.n_edge_ids = (int32_t)node.edges_.size();
    std::vector<uzl_span> eids;
    for (const std::string& e : node.edges_) eids.push_back(span_of(e));
    std::vector<uint8_t> msg(uzl_wire_node_size(&n, eids.data(), sens.data()));
    uint64_t w = 0;
    if (uzl_wire_node_encode(&n, node.stamps_.data(), eids.data(), sens.data(), msg.data(), msg.size(), &w) != UZL_OK) {
        last_error_ = "node_encode failed";
        return false;
    }
    return writeBag((fs::path(storage_path_) / "nodes" / node.id_).string(), "node", "graph_slam_msgs/Node", traits_.node_md5, traits_.node_def, now_ns, msg);
}

bool RosbagStorage::storeEdge(const SlamEdge& edge, int64_t now_ns)
{
    std::lock_guard<std::mutex> lock(rosbag_mutex_);
    uzl_wire_edge e;
    memset(&e, 0, sizeof(e));
    e.id = span_of(edge.id_); e.id_from = span_of(edge.id_from_); e.id_to = span_of(edge.id_to_);
    e.sensor_from = span_of(edge.sensor_from_); e.sensor_to = span_of(edge.sensor_to_);
    e.type = edge.type_; e.valid = edge.valid_;
    memcpy(e.transform, edge.transform_.m.data(), 96); memcpy(e.information, edge.information_.data(), 288);
    memcpy(e.displacement_from, edge.displacement_from_.m.data(), 96); memcpy(e.displacement_to, edge.displacement_to_.m.data(), 96);
    e.error = edge.error_; e.age = edge.age_; e.matching_score = edge.matching_score_;
    const int64_t dt = (int64_t)(edge.diff_time_ * 1e9 + (edge.diff_time_ >= 0 ? 0.5 : -0.5));       // ros::Duration(double): rounded to ns
    e.diff_time_sec = (int32_t)(dt / 1000000000); e.diff_time_nsec = (int32_t)(dt % 1000000000);
    if (e.diff_time_nsec < 0) { e.diff_time_nsec += 1000000000; e.diff_time_sec -= 1; }             // normalised: 0 <= nsec < 1e9
    std::vector<uint8_t> msg(uzl_wire_edge_size(&e));
    uint64_t w = 0;
    if (uzl_wire_edge_encode(&e, msg.data(), msg.size(), &w) != UZL_OK) { last_error_ = "edge_encode failed"; return false; }
    return writeBag((fs::path(storage_path_) / "edges" / edge.id_).string(), "edge", "graph_slam_msgs/Edge", traits_.edge_md5, traits_.edge_def, now_ns, msg);
}

bool RosbagStorage::storeMetaData(SlamGraph& graph, const Isometry3d& map_pose, int64_t now_ns)
{
    std::lock_guard<std::mutex> lock(rosbag_mutex_);
    // SlamGraph::toMetaData (slam_graph.cpp:592-619)
    uzl_wire_meta m;
    memset(&m, 0, sizeof(m));
    m.stamp_sec = (uint32_t)(now_ns / 1000000000); m.stamp_nsec = (uint32_t)(now_ns % 1000000000);
    m.frame_id = span_of(graph.frame_); m.name = span_of(graph.name_);
    memcpy(m.map_transform, map_pose.m.data(), 96);
    std::vector<uzl_wire_sensor_transform> st, sti;
    for (const auto& kv : graph.sensors()) { uzl_wire_sensor_transform t; t.sensor_name = span_of(kv.first); memcpy(t.transform, kv.second.m.data(), 96); st.push_back(t); }
    for (const auto& kv : graph.sensorsInitial()) { uzl_wire_sensor_transform t; t.sensor_name = span_of(kv.first); memcpy(t.transform, kv.second.m.data(), 96); sti.push_back(t); }
    m.n_sensor_transforms = (int32_t)st.size(); m.n_sensor_transforms_initial = (int32_t)sti.size();
    memcpy(m.odometry_parameters, graph.odom().data(), 48);
    std::vector<uint8_t> msg(uzl_wire_meta_size(&m, st.data(), sti.data()));
    uint64_t w = 0;
    if (uzl_wire_meta_encode(&m, st.data(), sti.data(), msg.data(), msg.size(), &w) != UZL_OK) { last_error_ = "meta_encode failed"; return false; }
    return writeBag((fs::path(storage_path_) / "meta" / "meta").string(), "meta", "graph_slam_msgs/GraphMeta", traits_.meta_md5, traits_.meta_def, now_ns, msg);
}

void RosbagStorage::removeNode(const std::string& id)
{
    std::lock_guard<std::mutex> lock(rosbag_mutex_);
    const fs::path p = fs::path(storage_path_) / "nodes" / id;
    if (fs::exists(p)) fs::remove(p);
}

void RosbagStorage::removeEdge(const std::string& id)
{
    std::lock_guard<std::mutex> lock(rosbag_mutex_);
    const fs::path p = fs::path(storage_path_) / "edges" / id;
    if (fs::exists(p)) fs::remove(p);
}

bool RosbagStorage::loadGraph(SlamGraph& graph)
{
    std::lock_guard<std::mutex> lock(rosbag_mutex_);
    bool ok = true;
    // ---- nodes: the first "node" message of every file (:149-156); files stay in memory until the device has unpacked them
    struct Pending { std::vector<uint8_t> file; SlamNode node; std::vector<uzl_wire_sensor> sensors; };
    std::vector<Pending> pend;
    for (const auto& it : fs::directory_iterator(fs::path(storage_path_) / "nodes")) {
        Pending p;
        if (!read_file(it.path(), p.file)) { ok = false; continue; }
        std::vector<uzl_bag_msg> msgs(4);
        int32_t nm = 0;
        if (uzl_bag_read(p.file.data(), p.file.size(), (int32_t)msgs.size(), msgs.data(), &nm) != UZL_OK) { ok = false; continue; }
        for (int32_t k = 0; k < nm && k < (int32_t)msgs.size(); k++) {
            if (str_of(msgs[k].topic) != "node") continue;
            const uint8_t* b = reinterpret_cast<const uint8_t*>(msgs[k].data.p);
            uzl_wire_node wn;
            if (uzl_wire_node_decode(b, msgs[k].data.n, &wn, 0, nullptr, 0, nullptr, 0, nullptr, nullptr) != UZL_OK) { ok = false; break; }
            std::vector<int64_t> stamps((size_t)wn.n_stamps);
            std::vector<uzl_span> eids((size_t)wn.n_edge_ids);
            p.sensors.resize((size_t)wn.n_sensors);
            uzl_wire_node_decode(b, msgs[k].data.n, &wn, wn.n_stamps, stamps.data(), wn.n_edge_ids, eids.data(), wn.n_sensors, p.sensors.data(), nullptr);
            SlamNode& n = p.node;                                   // Conversions::fromMsg(Node) (conversions.cpp:276-297)
            n.id_ = str_of(wn.id); n.stamps_ = stamps;
            memcpy(n.pose_.m.data(), wn.pose, 96); memcpy(n.sub_pose_.m.data(), wn.odom_pose, 96);
            n.fixed_ = wn.fixed != 0; n.uncertainty_ = wn.uncertainty;
            for (const uzl_span& e : eids) n.edges_.insert(str_of(e));
            pend.push_back(std::move(p));
            break;
        }
    }
    // ---- every feature frame of every node: one upload + one kernel (FeatureData::fromMsg on the device)
    std::vector<uzl_wire_sensor> batch;
    std::vector<std::pair<size_t, size_t>> owner;
    for (size_t i = 0; i < pend.size(); i++)
        for (size_t j = 0; j < pend[i].sensors.size(); j++)
            if (pend[i].sensors[j].sensor_type == UZL_SENSOR_TYPE_FEATURE) { batch.push_back(pend[i].sensors[j]); owner.emplace_back(i, j); }
    std::vector<int32_t> ids(batch.size(), -1), uv;
    int64_t total = 0;
    for (const uzl_wire_sensor& s : batch) total += s.n_features;
    uv.resize((size_t)std::max<int64_t>(2 * total, 1));
    if (!batch.empty() && uzl_match_add_frames_wire(estimator_, (int32_t)batch.size(), batch.data(), nullptr, ids.data(), uv.data()) != UZL_OK) {
        last_error_ = uzl_match_last_error(estimator_);
        return false;
    }
    int64_t row = 0;
    for (size_t k = 0; k < batch.size(); k++) {
        const uzl_wire_sensor& s = batch[k];
        auto fd = std::make_shared<FeatureData>();
        fd->type_ = s.sensor_type; fd->stamp_ = (int64_t)s.stamp_sec * 1000000000 + s.stamp_nsec;      // SensorData::fromMsg (sensor_data.cpp:52-58)
        fd->sensor_frame_ = str_of(s.sensor_frame);
        memcpy(fd->displacement_.m.data(), s.displacement, 96);
        fd->feature_type_ = s.descriptor_type;
        int32_t n = 0, bpd = 0;
        uzl_match_get_frame(estimator_, ids[k], nullptr, nullptr, nullptr, &n, &bpd);
        fd->rows = n; fd->bytes_per_row = n ? bpd : 0;
        fd->features_.resize((size_t)n * bpd); fd->feature_positions_.resize((size_t)n * 3);
        std::vector<uint8_t> valid((size_t)n);
        if (n) uzl_match_get_frame(estimator_, ids[k], fd->features_.data(), fd->feature_positions_.data(), valid.data(), &n, &bpd);
        fd->valid_3d_.assign(valid.begin(), valid.end());
        fd->feature_positions_2d_.assign(uv.begin() + 2 * row, uv.begin() + 2 * (row + n));
        row += n;
        uzl_match_remove_frame(estimator_, ids[k]);
        pend[owner[k].first].node.sensor_data_.push_back(fd);
    }
    for (Pending& p : pend) graph.addNode(p.node);
    // ---- edges: every "edge" message (:163-185)
    for (const auto& it : fs::directory_iterator(fs::path(storage_path_) / "edges")) {
        std::vector<uint8_t> file;
        if (!read_file(it.path(), file)) { ok = false; continue; }
        std::vector<uzl_bag_msg> msgs(16);
        int32_t nm = 0;
        if (uzl_bag_read(file.data(), file.size(), (int32_t)msgs.size(), msgs.data(), &nm) != UZL_OK) { ok = false; continue; }
        for (int32_t k = 0; k < nm && k < (int32_t)msgs.size(); k++) {
            if (str_of(msgs[k].topic) != "edge") continue;
            uzl_wire_edge we;
            if (uzl_wire_edge_decode(reinterpret_cast<const uint8_t*>(msgs[k].data.p), msgs[k].data.n, &we, nullptr) != UZL_OK) { ok = false; continue; }
            SlamEdge e;                                             // Conversions::fromMsg(Edge) (conversions.cpp:242-253)
            e.id_ = str_of(we.id); e.id_from_ = str_of(we.id_from); e.id_to_ = str_of(we.id_to);
            e.sensor_from_ = str_of(we.sensor_from); e.sensor_to_ = str_of(we.sensor_to);
            e.type_ = (unsigned char)we.type; e.valid_ = we.valid != 0;
            memcpy(e.transform_.m.data(), we.transform, 96); memcpy(e.information_.data(), we.information, 288);
            memcpy(e.displacement_from_.m.data(), we.displacement_from, 96); memcpy(e.displacement_to_.m.data(), we.displacement_to, 96);
            e.error_ = we.error; e.age_ = we.age; e.matching_score_ = we.matching_score;
            e.diff_time_ = (double)we.diff_time_sec + 1e-9 * (double)we.diff_time_nsec;
            graph.addEdge(e);
        }
    }
    // ---- meta: every "meta" message of every file under <path>/meta (:187-207) -> SlamGraph::updateMetaData (slam_graph.cpp:621-633)
    const fs::path meta_dir = fs::path(storage_path_) / "meta";
    if (fs::exists(meta_dir))
        for (const auto& it : fs::directory_iterator(meta_dir)) {
            std::vector<uint8_t> file;
            if (!read_file(it.path(), file)) { ok = false; continue; }
            std::vector<uzl_bag_msg> msgs(16);
            int32_t nm = 0;
            if (uzl_bag_read(file.data(), file.size(), (int32_t)msgs.size(), msgs.data(), &nm) != UZL_OK) { ok = false; continue; }
            for (int32_t k = 0; k < nm && k < (int32_t)msgs.size(); k++) {
                if (str_of(msgs[k].topic) != "meta") continue;
                const uint8_t* b = reinterpret_cast<const uint8_t*>(msgs[k].data.p);
                uzl_wire_meta wm;
                if (uzl_wire_meta_decode(b, msgs[k].data.n, &wm, 0, nullptr, 0, nullptr, nullptr) != UZL_OK) { ok = false; continue; }
                std::vector<uzl_wire_sensor_transform> st((size_t)wm.n_sensor_transforms), sti((size_t)wm.n_sensor_transforms_initial);
                uzl_wire_meta_decode(b, msgs[k].data.n, &wm, wm.n_sensor_transforms, st.data(), wm.n_sensor_transforms_initial, sti.data(), nullptr);
                graph.frame_ = str_of(wm.frame_id); graph.name_ = str_of(wm.name);
                memcpy(graph.sub_transform_.m.data(), wm.map_transform, 96);
                for (const auto& t : st) { Isometry3d T; memcpy(T.m.data(), t.transform, 96); graph.addSensor(str_of(t.sensor_name), T); }
                for (const auto& t : sti) { Isometry3d T; memcpy(T.m.data(), t.transform, 96); graph.sensorInitial(str_of(t.sensor_name)) = T; }
                memcpy(graph.odom().data(), wm.odometry_parameters, 48);
            }
        }
    return ok;
}

}  // namespace uzl_adapter
