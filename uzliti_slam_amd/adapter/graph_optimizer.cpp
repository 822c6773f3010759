#include "graph_optimizer.h"

#include <set>
#include <unordered_set>

#include <cmath>
#include <cstring>

namespace uzl_adapter {

GraphOptimizer::GraphOptimizer()
{
    graph_optimization_thread_ = std::thread(&GraphOptimizer::graphOptimizationThread, this);
}

GraphOptimizer::~GraphOptimizer() { stopThread(); }

void GraphOptimizer::stopThread()
{
    {   // under the mutex: a store + notify between the worker's predicate check and its block would be a lost wakeup
        std::lock_guard<std::mutex> lock(opt_mutex_);
        running = false;
    }
    opt_cv_.notify_all();
    if (graph_optimization_thread_.joinable()) graph_optimization_thread_.join();
}

bool GraphOptimizer::optimize(SlamGraph& graph, uzl_adapter::function<void()> callback)
{
    std::lock_guard<std::mutex> lock(opt_mutex_);
    if (do_optimization_) return false;        // a solve is already in flight
    callback_ = callback;
    addGraphImpl(graph);                       // only copies (caller holds its graph mutex)
    do_optimization_ = true;
    opt_cv_.notify_all();
    return true;
}

void GraphOptimizer::storeOptimizationResults(SlamGraph& graph) { storeImpl(graph); }

void GraphOptimizer::setConfig(GraphOptimizerConfig config)
{
    std::lock_guard<std::mutex> lock(opt_mutex_);
    config_ = config;
}

// The reference polls every 10 ms with plain bools; here a condition variable wakes the worker.
void GraphOptimizer::graphOptimizationThread()
{
    std::unique_lock<std::mutex> lock(opt_mutex_);
    while (running) {
        opt_cv_.wait(lock, [this] { return do_optimization_ || !running; });
        if (!running) break;
        uzl_adapter::function<void()> cb = callback_;
        lock.unlock();                         // no plugin lock while solving / calling back (graph_optimizer.cpp:63-67)
        optimizeImpl();
        if (cb) cb();
        lock.lock();
        do_optimization_ = false;
    }
}

Mi355xOptimizer::Mi355xOptimizer(int device, bool use_edge_filter, double cluster_size, uint64_t seed)
{
    uzl_pgo_cfg c;
    uzl_pgo_cfg_default(&c);
    c.device = device;
    status_ = uzl_pgo_create(&c, &h_);
    if (use_edge_filter) edge_filter_.reset(new TransformationFilter(5., (int)cluster_size, 100, device, seed));   // g2o_optimizer.cpp:46
}

Mi355xOptimizer::~Mi355xOptimizer()
{
    stopThread();                              // before the handle goes away
    if (h_) uzl_pgo_destroy(h_);
}

// Has the SlamGraph only GROWN since the handle got it - same nodes (ids, fixed flags, the poses storeImpl wrote back) and edges in front,
// new ones behind, same sensors and projection flags?  Then only the tail crosses the C ABI (uzl_pgo_append_graph).
bool Mi355xOptimizer::growsOnly(SlamGraph& graph, const std::vector<double>& sensors) const
{
    if (!have_graph_ || status_ < 0 || stored_poses_.size() != 12 * node_ids_.size()) return false;
    if (sensors != sent_sensors_ || sent_xy_ != config_.optimize_xy_only || sent_odom_ != config_.use_odometry_parameters) return false;
    if (graph.nodes().size() < node_ids_.size() || graph.edges().size() < edge_ids_.size()) return false;
    size_t i = 0;
    for (auto& kv : graph.nodes()) {
        if (i == node_ids_.size()) break;
        if (kv.first != node_ids_[i] || (kv.second.fixed_ ? 1 : 0) != sent_fixed_[i]) return false;
        if (std::memcmp(kv.second.pose_.m.data(), stored_poses_.data() + 12 * i, 12 * sizeof(double)) != 0) return false;
        i++;
    }
    size_t k = 0;
    for (auto& kv : graph.edges()) {
        if (k == edge_ids_.size()) break;
        if (kv.first != edge_ids_[k]) return false;
        k++;
    }
    return true;
}

// FNV-1a over the packed edge with its `valid` flag left out (the one field uzl_pgo_append_graph can change on an old edge)
uint64_t Mi355xOptimizer::edgeHash(const uzl_edge& e)
{
    uzl_edge u = e;
    u.valid = 0;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(&u);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < sizeof(u); i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

void Mi355xOptimizer::packEdge(const SlamEdge& e, const std::string& key, uzl_edge& u) const
{
    std::memset(&u, 0, sizeof(u));
    auto f = index_.find(e.id_from_), t = index_.find(e.id_to_);
    u.from = f == index_.end() ? -1 : f->second;        // missing endpoints are skipped by the back end (:77)
    u.to = t == index_.end() ? -1 : t->second;
    u.type = e.type_;
    auto sf = sensor_index_.find(e.sensor_from_), st = sensor_index_.find(e.sensor_to_);
    u.sensor_from = sf == sensor_index_.end() ? -1 : sf->second;
    u.sensor_to = st == sensor_index_.end() ? -1 : st->second;
    // TransformationFilter verdict (:97-103); odometry edges bypass the filter (:78-79)
    u.valid = (e.type_ == TYPE_2D_WHEEL_ODOMETRY) ? 1 : (edge_filter_ ? (filtered_.count(key) ? 1 : 0) : (e.valid_ ? 1 : 0));
    std::memcpy(u.transform, e.transform_.m.data(), sizeof(u.transform));
    std::memcpy(u.displacement_from, e.displacement_from_.m.data(), sizeof(u.displacement_from));
    std::memcpy(u.displacement_to, e.displacement_to_.m.data(), sizeof(u.displacement_to));
    std::memcpy(u.information, e.information_.data(), sizeof(u.information));
    u.diff_time = e.diff_time_;
}

void Mi355xOptimizer::addGraphImpl(SlamGraph& graph)
{
    if (!h_) return;
    uzl_pgo_cfg c;
    uzl_pgo_cfg_default(&c);
    c.iterations = config_.iterations;
    c.optimize_xy_only = config_.optimize_xy_only ? 1 : 0;
    c.use_odometry_parameters = config_.use_odometry_parameters ? 1 : 0;
    uzl_pgo_set_config(h_, &c);
    std::vector<double> sensors;
    std::map<std::string, int32_t> sensor_index;
    for (auto& kv : graph.sensors()) {          // sensor transforms (:68-71)
        sensor_index[kv.first] = (int32_t)(sensors.size() / 12);
        sensors.insert(sensors.end(), kv.second.m.begin(), kv.second.m.end());
    }
    // non-odometry edges go through the edge filter (g2o_optimizer.cpp:73-103)
    if (edge_filter_) {
        edge_filter_->sensor_transforms_ = graph.sensors();                                   // :71
        std::unordered_set<std::string> edges_to_remove = edge_filter_->allEdges();           // :74
        for (auto& kv : graph.edges()) {
            SlamEdge& e = kv.second;
            if (!graph.existsNode(e.id_from_) || !graph.existsNode(e.id_to_)) continue;       // :77
            if (e.type_ == TYPE_2D_WHEEL_ODOMETRY) continue;                                  // :78-79
            edge_filter_->add(e, graph.node(e.id_from_), graph.node(e.id_to_));               // :83
            edges_to_remove.erase(e.id_);                                                     // :84
        }
        for (const auto& id : edges_to_remove) edge_filter_->remove(id);                      // :89-92
        edge_filter_->calcValidEdges();                                                       // :96
    }
    filtered_.clear();
    if (edge_filter_)
        for (const SlamEdge& fe : edge_filter_->validEdges()) {                               // :97-102
            if (!graph.existsEdge(fe.id_)) continue;
            graph.edge(fe.id_).valid_ = true;
            filtered_.insert(fe.id_);
        }
    // growsOnly looks at ids, flags and poses; the reference also rewrites existing edges in place (mergeNodes moves displacement_from_ /
    // displacement_to_ and id_from_ / id_to_ under the same edge id, graph_slam_node.cpp:947-976), and an old edge whose end node was
    // missing when it was sent resolves once the node appears: every old edge is therefore packed again and compared (a hash of the
    // packed uzl_edge without its `valid` flag) with what was sent - any difference sends the whole graph (uzl_pgo_add_graph)
    bool grow = growsOnly(graph, sensors);
    std::vector<uzl_node> nodes;
    std::vector<uzl_edge> edges;
    std::vector<int32_t> flag_index;
    std::vector<uint8_t> flag_valid;
    for (int attempt = 0; attempt < 2; attempt++) {
        last_append_ = grow;
        if (!grow) { node_ids_.clear(); edge_ids_.clear(); index_.clear(); sent_fixed_.clear(); sent_valid_.clear(); sent_hash_.clear(); sensor_index_ = sensor_index; }
        nodes.clear(); edges.clear(); flag_index.clear(); flag_valid.clear();
        // vertices in std::map order = lexicographic id = the order g2o ids are assigned in (g2o_optimizer.cpp:64-66)
        const size_t n_old = node_ids_.size(), e_old = edge_ids_.size();
        size_t i = 0;
        for (auto& kv : graph.nodes()) {
            if (i++ < n_old) continue;
            uzl_node n;
            std::memcpy(n.pose, kv.second.pose_.m.data(), sizeof(n.pose));
            n.fixed = kv.second.fixed_ ? 1 : 0;
            index_[kv.first] = (int32_t)node_ids_.size();
            node_ids_.push_back(kv.first);
            sent_fixed_.push_back((uint8_t)n.fixed);
            nodes.push_back(n);
        }
        bool rewritten = false;
        size_t k = 0;
        for (auto& kv : graph.edges()) {
            uzl_edge u;
            packEdge(kv.second, kv.first, u);
            const uint64_t hash = edgeHash(u);
            if (k < e_old) {                         // an old edge: only the filter's verdict may have changed
                if (hash != sent_hash_[k]) { rewritten = true; break; }
                if ((uint8_t)u.valid != sent_valid_[k]) { flag_index.push_back((int32_t)k); flag_valid.push_back((uint8_t)u.valid); sent_valid_[k] = (uint8_t)u.valid; }
            } else {
                edge_ids_.push_back(kv.first);
                sent_valid_.push_back((uint8_t)u.valid);
                sent_hash_.push_back(hash);
                edges.push_back(u);
            }
            k++;
        }
        if (!rewritten) break;
        grow = false;                                // (second round: everything is sent)
    }
    if (grow)
        status_ = uzl_pgo_append_graph(h_, (int32_t)nodes.size(), nodes.data(), (int32_t)edges.size(), edges.data(), (int32_t)flag_index.size(),
                                       flag_index.data(), flag_valid.data());
    else
        status_ = uzl_pgo_add_graph(h_, (int32_t)nodes.size(), nodes.data(), (int32_t)edges.size(), edges.data(),
                                    (int32_t)(sensors.size() / 12), sensors.data());
    have_graph_ = status_ >= 0;
    sent_sensors_ = sensors; sent_xy_ = config_.optimize_xy_only; sent_odom_ = config_.use_odometry_parameters;
    stored_poses_.clear();                       // (valid again once storeImpl has run for this graph)
}

void Mi355xOptimizer::optimizeImpl()
{
    if (!h_ || status_ < 0) return;             // reference: ROS_ERROR + return (g2o_optimizer.cpp:139-142)
    status_ = uzl_pgo_optimize(h_, config_.iterations, &stats_);
    if (status_ == UZL_ERR_NOT_CONVERGED) status_ = 0;     // result is still the best available estimate
}

void Mi355xOptimizer::storeImpl(SlamGraph& graph)
{
    if (!h_ || status_ < 0) return;
    std::vector<double> poses(node_ids_.size() * 12), err(edge_ids_.size() + 1);
    std::vector<uint8_t> used(edge_ids_.size() + 1);
    if (uzl_pgo_store(h_, poses.data(), err.data(), used.data()) != UZL_OK) return;
    stored_poses_ = poses;                      // what the handle's estimates look like from outside: growsOnly compares the SlamGraph with it
    for (size_t i = 0; i < node_ids_.size(); i++) {        // :110-117: the graph may have changed meanwhile
        if (!graph.existsNode(node_ids_[i])) continue;
        SlamNode& n = graph.node(node_ids_[i]);
        std::memcpy(n.pose_.m.data(), poses.data() + 12 * i, 12 * sizeof(double));
        n.optimized_ = true;
    }
    for (size_t k = 0; k < edge_ids_.size(); k++) {        // :120-134
        if (!used[k] || !graph.existsEdge(edge_ids_[k])) continue;
        SlamEdge& e = graph.edge(edge_ids_[k]);
        e.age_ += 1;
        e.error_ = err[k];
    }
}

}  // namespace uzl_adapter
