// graph_optimizer.h — host-side mirror of the reference's GraphOptimizer plugin family
// (graph_optimization/include/graph_optimization/graph_optimizer.h:28-56, src/graph_optimizer.cpp:22-73) with the
// MI355X back end in the place of G2oOptimizer (g2o_optimizer.h:38-68).  Same public calls, same threading
// contract: optimize() copies the graph under the caller's lock and returns, the solve runs on the worker thread,
// the completion callback fires on the worker thread with no plugin lock held, optimize() returns false while a
// solve is in flight.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <set>
#include <thread>

#include "slam_types.h"
#include "transformation_filter.h"
#include "../../include/uzl_mi355x.h"

namespace uzl_adapter {

class GraphOptimizer {
public:
    GraphOptimizer();
    virtual ~GraphOptimizer();
    bool optimize(SlamGraph& graph, uzl_adapter::function<void()> callback);      // graph_optimizer.cpp:35-47
    void storeOptimizationResults(SlamGraph& graph);                      // :49-52
    void setConfig(GraphOptimizerConfig config);                          // :54-57

protected:
    void graphOptimizationThread();                                        // :59-73
    void stopThread();
    virtual void addGraphImpl(SlamGraph& graph) = 0;
    virtual void storeImpl(SlamGraph& graph) = 0;
    virtual void optimizeImpl() = 0;

    std::atomic<bool> running{true};
    std::thread graph_optimization_thread_;
    std::mutex opt_mutex_;
    std::condition_variable opt_cv_;
    bool do_optimization_ = false;
    uzl_adapter::function<void()> callback_;
    GraphOptimizerConfig config_;
};

// The back end behind the interface: what a maintainer registers instead of G2oOptimizer
// (graph_slam/src/graph_slam_node.cpp:46).
class Mi355xOptimizer : public GraphOptimizer {
public:
    // use_edge_filter: route non-odometry edges through TransformationFilter like G2oOptimizer::addGraphImpl
    // (g2o_optimizer.cpp:74-103); false = take SlamEdge::valid_ as the verdict.  cluster_size is the reference's
    // "cluster_size" ROS parameter (g2o_optimizer.cpp:43-46).
    explicit Mi355xOptimizer(int device = 0, bool use_edge_filter = true, double cluster_size = 8, uint64_t seed = 0);
    ~Mi355xOptimizer() override;
    const uzl_pgo_stats& lastStats() const { return stats_; }
    int lastStatus() const { return status_; }
    TransformationFilter* edgeFilter() { return edge_filter_.get(); }
    const std::set<std::string>& lastFiltered() const { return filtered_; }   // ids the filter passed to the solver in the last addGraph
    bool lastWasAppend() const { return last_append_; }                       // the last addGraph grew the resident graph (uzl_pgo_append_graph)

protected:
    void addGraphImpl(SlamGraph& graph) override;     // g2o_optimizer.cpp:55-104
    void storeImpl(SlamGraph& graph) override;        // :106-135
    void optimizeImpl() override;                     // :137-149

private:
    uzl_pgo* h_ = nullptr;
    std::unique_ptr<TransformationFilter> edge_filter_;          // g2o_optimizer.h:67
    std::set<std::string> filtered_;
    std::vector<std::string> node_ids_, edge_ids_;    // index <-> string id (the reference's boost::bimap, g2o_optimizer.h:35-36)
    uzl_pgo_stats stats_{};
    int status_ = 0;
    bool applied_xy_ = false;
    // what the handle holds, for the grown-only path of an online session (graph_slam_node.cpp:1138-1150 re-optimises a graph that gained
    // nodes and edges; ids are time-ordered, :294, so they sort behind the old ones)
    bool growsOnly(SlamGraph& graph, const std::vector<double>& sensors) const;
    void packEdge(const SlamEdge& e, const std::string& key, uzl_edge& u) const;
    static uint64_t edgeHash(const uzl_edge& u);
    std::map<std::string, int32_t> index_, sensor_index_;
    std::vector<double> stored_poses_, sent_sensors_;   // poses storeImpl wrote back (12 per node); sensors as sent
    std::vector<uint8_t> sent_fixed_, sent_valid_;
    std::vector<uint64_t> sent_hash_;            // per sent edge: edgeHash of what went across the C ABI
    bool have_graph_ = false, last_append_ = false, sent_xy_ = false, sent_odom_ = false;
};

}  // namespace uzl_adapter
