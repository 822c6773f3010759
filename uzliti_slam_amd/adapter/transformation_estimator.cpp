#include "transformation_estimator.h"

#include <cstring>

namespace uzl_adapter {

TransformationEstimator::TransformationEstimator(uzl_adapter::function<void(SlamEdge)> callback) : callback_(callback)
{
    estimation_thread_ = std::thread(&TransformationEstimator::estimationThread, this);
}

TransformationEstimator::~TransformationEstimator() { stopThread(); }

void TransformationEstimator::stopThread()
{
    {   // under the mutex, so that the worker cannot miss the wakeup between its predicate check and its block
        std::lock_guard<std::mutex> lock(estimation_mutex_);
        running_ = false;
    }
    estimation_cv_.notify_all();
    if (estimation_thread_.joinable()) estimation_thread_.join();
}

void TransformationEstimator::estimateEdge(SlamNode& from, SlamNode& to)
{
    std::lock_guard<std::mutex> lock(estimation_mutex_);
    est_queue_.push_back(std::make_pair(from, to));      // deep copy of both nodes, as the reference does (:39)
    estimation_cv_.notify_all();
}

void TransformationEstimator::estimateBatch(std::vector<std::pair<SlamNode, SlamNode>>& pairs, std::vector<SlamEdge>& edges,
                                            std::vector<char>& ok)
{
    for (size_t i = 0; i < pairs.size(); i++) ok[i] = estimateEdgeImpl(pairs[i].first, pairs[i].second, edges[i]) ? 1 : 0;
}

void TransformationEstimator::estimationThread()
{
    std::unique_lock<std::mutex> lock(estimation_mutex_);
    while (running_) {
        estimation_cv_.wait(lock, [this] { return !est_queue_.empty() || !running_; });
        if (!running_) break;
        std::vector<std::pair<SlamNode, SlamNode>> batch;
        batch.swap(est_queue_);
        lock.unlock();
        // the reference pops from the back (LIFO, :49); keep that delivery order
        std::vector<std::pair<SlamNode, SlamNode>> lifo(batch.rbegin(), batch.rend());
        std::vector<SlamEdge> edges(lifo.size());
        std::vector<char> ok(lifo.size(), 0);
        estimateBatch(lifo, edges, ok);
        for (size_t i = 0; i < lifo.size(); i++) {
            if (!ok[i]) edges[i].matching_score_ = 0.;    // :53-55
            callback_(edges[i]);                           // one callback per pair, on the worker thread (:56)
        }
        lock.lock();
    }
}

Mi355xFeatureTransformationEstimator::Mi355xFeatureTransformationEstimator(uzl_adapter::function<void(SlamEdge)> callback,
                                                                           int device, uint64_t seed)
    : TransformationEstimator(callback)
{
    uzl_match_cfg_default(&cfg_);
    cfg_.device = device;
    cfg_.seed = seed;
    status_ = uzl_match_create(&cfg_, &h_);
}

Mi355xFeatureTransformationEstimator::~Mi355xFeatureTransformationEstimator()
{
    stopThread();
    if (h_) uzl_match_destroy(h_);
}

void Mi355xFeatureTransformationEstimator::setConfig(FeatureLinkEstimationConfig config)
{
    std::lock_guard<std::mutex> lock(cfg_mutex_);
    cfg_.ransac_threshold = config.ransac_threshold;
    cfg_.link_covariance = config.link_covariance;
    cfg_.ransac_iteration = config.ransac_iteration;
    cfg_.ransac_break_percentage = config.ransac_break_percentage;
    cfg_.use_epnp = config.use_epnp ? 1 : 0;
    if (h_) status_ = uzl_match_set_config(h_, &cfg_);
}

int32_t Mi355xFeatureTransformationEstimator::sensorKey(const std::string& frame)
{
    auto it = sensor_keys_.find(frame);
    if (it != sensor_keys_.end()) return it->second;
    const int32_t k = (int32_t)sensor_keys_.size();
    sensor_keys_[frame] = k;
    return k;
}

// one upload per FeatureData object; node copies share the shared_ptr, so later pairs reuse the resident frame
int32_t Mi355xFeatureTransformationEstimator::frameId(const FeatureDataPtr& fd)
{
    auto it = frame_ids_.find(fd.get());
    if (it != frame_ids_.end()) return it->second;
    std::vector<uint8_t> valid(fd->valid_3d_.size());
    for (size_t i = 0; i < valid.size(); i++) valid[i] = fd->valid_3d_[i] ? 1 : 0;
    uzl_frame f;
    std::memset(&f, 0, sizeof(f));
    f.desc = fd->features_.data(); f.n = fd->rows; f.bytes_per_desc = fd->bytes_per_row;
    f.pos_xyz = fd->feature_positions_.data(); f.valid3d = valid.data();
    f.feature_type = fd->feature_type_; f.sensor_frame = sensorKey(fd->sensor_frame_);
    std::memcpy(f.displacement, fd->displacement_.m.data(), sizeof(f.displacement));
    int32_t id = -1;
    if (uzl_match_add_frame(h_, &f, &id) != UZL_OK) return -1;
    frame_ids_[fd.get()] = id;
    keep_alive_[fd.get()] = fd;
    return id;
}

// every FeatureData of the queued pairs that is not resident yet: ONE uzl_match_add_frames call (one arena extent, threaded packing,
// one DMA per staging half) instead of a call and a copy per frame
void Mi355xFeatureTransformationEstimator::uploadNewFrames(const std::vector<std::pair<SlamNode, SlamNode>>& pairs)
{
    std::vector<FeatureDataPtr> fresh;
    std::unordered_map<const FeatureData*, char> seen;
    auto visit = [&](const SlamNode& nd) {
        for (auto& sd : nd.sensor_data_) {
            if (sd->type_ != SENSOR_TYPE_FEATURE) continue;
            FeatureDataPtr fd = std::dynamic_pointer_cast<FeatureData>(sd);
            if (!fd || frame_ids_.count(fd.get()) || seen.count(fd.get())) continue;
            seen[fd.get()] = 1;
            fresh.push_back(fd);
        }
    };
    for (auto& pr : pairs) { visit(pr.first); visit(pr.second); }
    if (fresh.size() < 2) return;                           // a single new frame takes frameId()'s path
    std::vector<std::vector<uint8_t>> valid(fresh.size());
    std::vector<uzl_frame> fr(fresh.size());
    for (size_t k = 0; k < fresh.size(); k++) {
        const FeatureDataPtr& fd = fresh[k];
        valid[k].resize(fd->valid_3d_.size());
        for (size_t i = 0; i < valid[k].size(); i++) valid[k][i] = fd->valid_3d_[i] ? 1 : 0;
        uzl_frame& f = fr[k];
        std::memset(&f, 0, sizeof(f));
        f.desc = fd->features_.data(); f.n = fd->rows; f.bytes_per_desc = fd->bytes_per_row;
        f.pos_xyz = fd->feature_positions_.data(); f.valid3d = valid[k].data();
        f.feature_type = fd->feature_type_; f.sensor_frame = sensorKey(fd->sensor_frame_);
        std::memcpy(f.displacement, fd->displacement_.m.data(), sizeof(f.displacement));
    }
    std::vector<int32_t> ids(fresh.size(), -1);
    if (uzl_match_add_frames(h_, (int32_t)fr.size(), fr.data(), ids.data()) != UZL_OK) return;     // frameId() retries one by one
    for (size_t k = 0; k < fresh.size(); k++) { frame_ids_[fresh[k].get()] = ids[k]; keep_alive_[fresh[k].get()] = fresh[k]; }
}

void Mi355xFeatureTransformationEstimator::estimateBatch(std::vector<std::pair<SlamNode, SlamNode>>& pairs,
                                                         std::vector<SlamEdge>& edges, std::vector<char>& ok)
{
    const int32_t n = (int32_t)pairs.size();
    if (!h_ || n == 0) return;
    uploadNewFrames(pairs);
    std::vector<uzl_pair_job> jobs((size_t)n);
    std::vector<int32_t> ids;
    std::vector<std::vector<FeatureDataPtr>> lookup;       // frame id -> FeatureData for sensor_from_/displacement_
    std::unordered_map<int32_t, FeatureDataPtr> by_id;
    for (int32_t j = 0; j < n; j++) {
        jobs[j].job_id = next_job_++;
        jobs[j].from_begin = (int32_t)ids.size();
        for (auto& sd : pairs[j].first.sensor_data_) {      // only SENSOR_TYPE_FEATURE takes part (:41,:43)
            if (sd->type_ != SENSOR_TYPE_FEATURE) continue;
            FeatureDataPtr fd = std::dynamic_pointer_cast<FeatureData>(sd);
            if (!fd) continue;
            const int32_t id = frameId(fd);
            if (id >= 0) { ids.push_back(id); by_id[id] = fd; }
        }
        jobs[j].from_count = (int32_t)ids.size() - jobs[j].from_begin;
        jobs[j].to_begin = (int32_t)ids.size();
        for (auto& sd : pairs[j].second.sensor_data_) {
            if (sd->type_ != SENSOR_TYPE_FEATURE) continue;
            FeatureDataPtr fd = std::dynamic_pointer_cast<FeatureData>(sd);
            if (!fd) continue;
            const int32_t id = frameId(fd);
            if (id >= 0) { ids.push_back(id); by_id[id] = fd; }
        }
        jobs[j].to_count = (int32_t)ids.size() - jobs[j].to_begin;
    }
    std::vector<uzl_edge_result> res((size_t)n);
    {
        std::lock_guard<std::mutex> lock(cfg_mutex_);
        status_ = uzl_match_estimate(h_, n, jobs.data(), ids.data(), (int32_t)ids.size(), res.data(), 0, nullptr, nullptr,
                                     nullptr, nullptr);
    }
    for (int32_t j = 0; j < n; j++) {
        SlamEdge& e = edges[j];
        e.id_from_ = pairs[j].first.id_;                    // estimateEdgeImpl (:168-169)
        e.id_to_ = pairs[j].second.id_;
        if (status_ != UZL_OK || !res[j].ok) { ok[j] = 0; continue; }
        const uzl_edge_result& r = res[j];
        std::memcpy(e.transform_.m.data(), r.T, sizeof(r.T));               // :147
        std::memcpy(e.information_.data(), r.information, sizeof(r.information));   // :148
        e.type_ = TYPE_3D_FULL;                                              // :149
        const FeatureDataPtr& ff = by_id[r.frame_from];
        const FeatureDataPtr& ft = by_id[r.frame_to];
        e.sensor_from_ = ff->sensor_frame_; e.sensor_to_ = ft->sensor_frame_;   // :150-151
        e.displacement_from_ = ff->displacement_; e.displacement_to_ = ft->displacement_;   // :152-153
        e.matching_score_ = r.consensus;                                     // :155
        ok[j] = 1;
    }
}

bool Mi355xFeatureTransformationEstimator::estimateEdgeImpl(SlamNode& from, SlamNode& to, SlamEdge& edge)
{
    std::vector<std::pair<SlamNode, SlamNode>> one(1, std::make_pair(from, to));
    std::vector<SlamEdge> e(1);
    std::vector<char> ok(1, 0);
    estimateBatch(one, e, ok);
    edge = e[0];
    return ok[0] != 0;
}

void Mi355xFeatureTransformationEstimator::estimateSVD(const std::vector<double>& P, const std::vector<double>& Q,
                                                       Isometry3d& T, int& consensus, double& mse, double maxError,
                                                       int iterations, double breakPercentage, bool do_prosac)
{
    const int32_t m = (int32_t)(P.size() / 3);
    const int32_t offs[2] = {0, m};
    const uint64_t job = next_job_++;
    int32_t cons = 0, it = 0;
    double e = 0., Tm[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    std::vector<uint8_t> mask((size_t)m + 1);
    status_ = uzl_ransac_points(h_, 1, offs, P.data(), Q.data(), maxError, iterations, breakPercentage, do_prosac ? 1 : 0,
                                &job, Tm, &cons, &e, &it, mask.data());
    std::memcpy(T.m.data(), Tm, sizeof(Tm));
    consensus = cons;
    mse = e;
}

}  // namespace uzl_adapter
