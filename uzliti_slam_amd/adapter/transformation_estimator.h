// transformation_estimator.h — host-side mirror of the reference's TransformationEstimator plugin family
// (transformation_estimation/include/transformation_estimation/transformation_estimator.h:45-67,
//  src/transformation_estimator.cpp:22-62) with the MI355X back end in the place of
// FeatureTransformationEstimator (feature_transformation_estimator.h:33-60).
// Same contract: estimateEdge() enqueues a node pair and returns; the worker thread delivers exactly one
// callback per enqueued pair, success or not (matching_score_ = 0 on failure, transformation_estimator.cpp:53-56).
// What changes is the worker: instead of one pair per millisecond it drains the whole queue into ONE batched
// uzl_match_estimate call (frames are uploaded to HBM once per FeatureData and referenced by id afterwards).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "slam_types.h"
#include "../../include/uzl_mi355x.h"

namespace uzl_adapter {

class TransformationEstimator {
public:
    explicit TransformationEstimator(uzl_adapter::function<void(SlamEdge)> callback);
    virtual ~TransformationEstimator();
    void estimateEdge(SlamNode& from, SlamNode& to);                               // transformation_estimator.cpp:35-43
    virtual bool estimateEdgeImpl(SlamNode& from, SlamNode& to, SlamEdge& edge) = 0;   // .h:54
    std::map<std::string, Isometry3d> sensor_transforms_;

protected:
    void estimationThread();                                                        // :45-62
    void stopThread();
    // batched hook: default = one estimateEdgeImpl per pair (the reference's behaviour)
    virtual void estimateBatch(std::vector<std::pair<SlamNode, SlamNode>>& pairs, std::vector<SlamEdge>& edges,
                               std::vector<char>& ok);

    std::thread estimation_thread_;
    std::mutex estimation_mutex_;
    std::condition_variable estimation_cv_;
    std::atomic<bool> running_{true};
    std::vector<std::pair<SlamNode, SlamNode>> est_queue_;
    uzl_adapter::function<void(SlamEdge)> callback_;
};

class Mi355xFeatureTransformationEstimator : public TransformationEstimator {
public:
    Mi355xFeatureTransformationEstimator(uzl_adapter::function<void(SlamEdge)> callback, int device = 0, uint64_t seed = 0);
    ~Mi355xFeatureTransformationEstimator() override;
    bool estimateEdgeImpl(SlamNode& from, SlamNode& to, SlamEdge& edge) override;
    void setConfig(FeatureLinkEstimationConfig config);                             // feature_transformation_estimator.cpp:350-353
    // estimateSVD twin (feature_transformation_estimator.h:45) for TransformationFilter
    void estimateSVD(const std::vector<double>& P, const std::vector<double>& Q, Isometry3d& T, int& consensus,
                     double& mse, double maxError, int iterations, double breakPercentage, bool do_prosac = true);
    int lastStatus() const { return status_; }

protected:
    void estimateBatch(std::vector<std::pair<SlamNode, SlamNode>>& pairs, std::vector<SlamEdge>& edges,
                       std::vector<char>& ok) override;

private:
    int32_t frameId(const FeatureDataPtr& fd);
    void uploadNewFrames(const std::vector<std::pair<SlamNode, SlamNode>>& pairs);
    int32_t sensorKey(const std::string& frame);
    uzl_match* h_ = nullptr;
    uzl_match_cfg cfg_{};
    std::mutex cfg_mutex_;
    std::unordered_map<const FeatureData*, int32_t> frame_ids_;       // FeatureData are shared_ptr'd and immutable once created
    std::unordered_map<const FeatureData*, FeatureDataPtr> keep_alive_;
    std::unordered_map<std::string, int32_t> sensor_keys_;
    uint64_t next_job_ = 0;
    int status_ = 0;
};

}  // namespace uzl_adapter
