// adapter_selftest.cpp — drives the plugin-shaped classes the way GraphSlamNode does
// (graph_slam/src/graph_slam_node.cpp:46-49, :266, :1138-1150, :1248-1282) on inputs read from a flat binary
// file written by tests/test_adapter_gpu.py, and writes the results back for comparison with the oracle.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "graph_optimizer.h"
#include "rosbag_storage.h"
#include "transformation_estimator.h"

using namespace uzl_adapter;

static void rd(FILE* f, void* p, size_t n) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

static std::string node_id(int i) { char b[32]; snprintf(b, sizeof(b), "n%08d", i); return b; }   // lexicographic = numeric order

// ---- "filter" mode: R addGraph/optimize/store rounds with the edge filter in the loop, the way GraphSlamNode drives
// G2oOptimizer while the graph grows (g2o_optimizer.cpp:55-104).  Input: nodes (pose, stamps), sensors, feature and
// odometry edges with the round they appear in / disappear in.  Output per round: SlamEdge::valid_ as the solver saw
// it, poses before and after.
static int filter_mode(const char* in, const char* out)
{
    FILE* f = fopen(in, "rb");
    if (!f) return 2;
    int32_t n, e, ns, rounds, iters; uint64_t seed; double cluster_size;
    rd(f, &n, 4); rd(f, &e, 4); rd(f, &ns, 4); rd(f, &rounds, 4); rd(f, &iters, 4); rd(f, &seed, 8); rd(f, &cluster_size, 8);
    std::vector<SlamNode> nodes(n);
    for (int i = 0; i < n; i++) {
        nodes[i].id_ = node_id(i);
        int32_t fixed, nst; rd(f, nodes[i].pose_.m.data(), 96); rd(f, &fixed, 4); rd(f, &nst, 4);
        nodes[i].fixed_ = fixed != 0; nodes[i].stamps_.resize(nst); rd(f, nodes[i].stamps_.data(), 8 * (size_t)nst);
    }
    std::vector<Isometry3d> sensors(ns);
    for (int i = 0; i < ns; i++) rd(f, sensors[i].m.data(), 96);
    std::vector<SlamEdge> edges(e); std::vector<int32_t> born(e), dies(e);
    for (int k = 0; k < e; k++) {
        SlamEdge& ed = edges[k]; char b[32]; snprintf(b, sizeof(b), "e%08d", k); ed.id_ = b;
        int32_t from, to, type, valid, sf, st; double score;
        rd(f, &from, 4); rd(f, &to, 4); rd(f, &type, 4); rd(f, &valid, 4); rd(f, &sf, 4); rd(f, &st, 4); rd(f, &born[k], 4); rd(f, &dies[k], 4); rd(f, &score, 8);
        ed.id_from_ = node_id(from); ed.id_to_ = node_id(to); ed.type_ = (unsigned char)type; ed.valid_ = valid != 0; ed.matching_score_ = score;
        if (sf >= 0) { char s[32]; snprintf(s, sizeof(s), "sensor%d", sf); ed.sensor_from_ = s; }
        if (st >= 0) { char s[32]; snprintf(s, sizeof(s), "sensor%d", st); ed.sensor_to_ = s; }
        rd(f, ed.transform_.m.data(), 96); rd(f, ed.displacement_from_.m.data(), 96); rd(f, ed.displacement_to_.m.data(), 96);
        rd(f, ed.information_.data(), 288);
    }
    fclose(f);
    FILE* o = fopen(out, "wb");
    SlamGraph graph;
    for (auto& nd : nodes) graph.addNode(nd);
    for (int i = 0; i < ns; i++) { char s[32]; snprintf(s, sizeof(s), "sensor%d", i); graph.addSensor(s, sensors[i]); }
    Mi355xOptimizer opt(0, /*use_edge_filter=*/true, cluster_size, seed);
    GraphOptimizerConfig cfg; cfg.iterations = iters;
    opt.setConfig(cfg);
    for (int r = 0; r < rounds; r++) {
        for (int k = 0; k < e; k++) {
            if (born[k] == r) graph.addEdge(edges[k]);
            if (dies[k] == r) graph.edges().erase(edges[k].id_);
        }
        for (auto& kv : graph.nodes()) fwrite(kv.second.pose_.m.data(), 8, 12, o);              // poses the filter sees
        std::mutex m; std::condition_variable cv; bool done = false;
        if (!opt.optimize(graph, [&] { std::lock_guard<std::mutex> l(m); done = true; cv.notify_all(); })) return 3;
        { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return done; }); }
        opt.storeOptimizationResults(graph);
        int32_t hdr[4] = {opt.lastStatus(), opt.lastStats().iterations_done, opt.edgeFilter()->lastEvaluated(), opt.lastStats().n_edges};
        fwrite(hdr, 4, 4, o);
        for (int k = 0; k < e; k++) {
            int8_t v = graph.existsEdge(edges[k].id_) ? (graph.edge(edges[k].id_).valid_ ? 1 : 0) : -1;
            int8_t u = opt.lastFiltered().count(edges[k].id_) ? 1 : 0;
            fwrite(&v, 1, 1, o); fwrite(&u, 1, 1, o);
        }
        for (auto& kv : graph.nodes()) fwrite(kv.second.pose_.m.data(), 8, 12, o);
    }
    fclose(o);
    return 0;
}

// ---- "storage" mode: RosbagStorage the way GraphSlamNode uses it (graph_slam_node.cpp: storeNode / storeEdge per new object,
// loadGraph at start-up).  A small graph with feature frames from a fixed LCG is stored, two objects are removed, the rest is
// loaded into a fresh SlamGraph and compared field by field; the directory stays behind for the Python-side check.
static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
static Isometry3d lcg_pose(uint32_t& s)
{
    const double a = 0.001 * (lcg(s) % 3000), b = 0.001 * (lcg(s) % 1500);      // yaw, pitch
    const double ca = cos(a), sa = sin(a), cb = cos(b), sb = sin(b);
    Isometry3d T;
    T.m = {ca * cb, -sa, ca * sb, 0.01 * (lcg(s) % 1000), sa * cb, ca, sa * sb, 0.01 * (lcg(s) % 1000), -sb, 0., cb, 0.01 * (lcg(s) % 100)};
    return T;
}
static double max_abs_diff(const Isometry3d& a, const Isometry3d& b)
{
    double m = 0;
    for (int i = 0; i < 12; i++) m = std::max(m, std::fabs(a.m[i] - b.m[i]));
    return m;
}
static int storage_mode(const char* dir, const char* out)
{
    uzl_match_cfg mc; uzl_match_cfg_default(&mc);
    uzl_match* est = nullptr;
    if (uzl_match_create(&mc, &est) != UZL_OK) { fprintf(stderr, "no estimator handle (no HIP device?)\n"); return 3; }
    uint32_t s = 12345;
    const int N = 6, E = 9;
    SlamGraph g;
    for (int i = 0; i < N; i++) {
        SlamNode nd; nd.id_ = node_id(i);
        nd.stamps_ = {1400000000000000000ll + 1000000000ll * i + 5, 1400000000500000000ll + 1000000000ll * i};
        nd.pose_ = lcg_pose(s); nd.sub_pose_ = lcg_pose(s); nd.fixed_ = i == 0; nd.uncertainty_ = 0.125 * i;
        nd.edges_ = {"e" + std::to_string(i), "e" + std::to_string(i + 1)};
        for (int k = 0; k < (i % 3 == 2 ? 2 : 1); k++) {                       // some nodes carry two cameras
            FeatureDataPtr fd(new FeatureData());
            fd->type_ = SENSOR_TYPE_FEATURE; fd->stamp_ = nd.stamps_[0]; fd->sensor_frame_ = k ? "cam_right" : "cam_left";
            fd->displacement_ = lcg_pose(s); fd->feature_type_ = FEATURE_ORB;
            fd->rows = (i == 4 && k == 0) ? 0 : 100 + 37 * i + k; fd->bytes_per_row = 32;
            fd->features_.resize((size_t)fd->rows * 32);
            for (auto& b : fd->features_) b = (uint8_t)lcg(s);
            fd->feature_positions_.resize((size_t)fd->rows * 3);
            for (auto& v : fd->feature_positions_) v = 0.001 * (double)(lcg(s) % 8000) - 4.0;
            fd->feature_positions_2d_.resize((size_t)fd->rows * 2);
            for (auto& v : fd->feature_positions_2d_) v = (int32_t)(lcg(s) % 640);
            fd->valid_3d_.resize((size_t)fd->rows);
            for (size_t q = 0; q < fd->valid_3d_.size(); q++) fd->valid_3d_[q] = lcg(s) % 5 != 0;
            nd.sensor_data_.push_back(fd);
        }
        g.addNode(nd);
    }
    for (int k = 0; k < E; k++) {
        SlamEdge e; char b[32]; snprintf(b, sizeof(b), "e%08d", k); e.id_ = b;
        e.id_from_ = node_id(k % N); e.id_to_ = node_id((k + 1 + k / N) % N);
        e.transform_ = lcg_pose(s); e.displacement_from_ = lcg_pose(s); e.displacement_to_ = lcg_pose(s);
        for (int i = 0; i < 36; i++) e.information_[i] = (i / 6 == i % 6) ? 10.0 + k : 0.01 * (double)((i / 6) * (i % 6));
        e.type_ = k % 2 ? TYPE_3D_FULL : TYPE_2D_WHEEL_ODOMETRY; e.sensor_from_ = k % 2 ? "cam_left" : ""; e.sensor_to_ = e.sensor_from_;
        e.age_ = k; e.error_ = 0.5 * k; e.matching_score_ = 40 + k; e.diff_time_ = (k % 3 == 0) ? -1.25 : 2.5 + k; e.valid_ = k % 3 != 1;
        g.addEdge(e);
    }
    RosbagStorage st(est, dir, /*clear_storage=*/true);
    for (auto& kv : g.nodes()) if (!st.storeNode(kv.second, 1500000000000000000ll)) { fprintf(stderr, "storeNode: %s\n", st.lastError().c_str()); return 4; }
    for (auto& kv : g.edges()) if (!st.storeEdge(kv.second, 1500000000000000000ll)) { fprintf(stderr, "storeEdge: %s\n", st.lastError().c_str()); return 4; }
    g.name_ = "selftest"; g.frame_ = "/map";
    g.addSensor("cam_left", lcg_pose(s)); g.addSensor("cam_right", lcg_pose(s)); g.sensorInitial("cam_left") = lcg_pose(s);
    for (int i = 0; i < 6; i++) g.odom()[(size_t)i] = 0.25 * (i + 1);
    if (!st.storeMetaData(g, lcg_pose(s), 1500000000000000000ll)) { fprintf(stderr, "storeMetaData: %s\n", st.lastError().c_str()); return 4; }
    st.removeNode(node_id(1)); st.removeEdge("e00000003"); st.removeEdge("never-stored");
    SlamGraph back;
    RosbagStorage st2(est, dir, false);
    if (!st2.loadGraph(back)) { fprintf(stderr, "loadGraph: %s\n", st2.lastError().c_str()); return 5; }
    int bad = 0;
    if (back.name_ != g.name_ || back.frame_ != g.frame_ || back.odom() != g.odom() || back.sensors().size() != 2 || back.sensorsInitial().size() != 1 ||
        max_abs_diff(back.sensors().at("cam_right"), g.sensors().at("cam_right")) > 1e-14 ||
        max_abs_diff(back.sensorInitial("cam_left"), g.sensorInitial("cam_left")) > 1e-14) bad |= 128;
    if (back.nodes().size() != (size_t)N - 1 || back.edges().size() != (size_t)E - 1 || back.existsNode(node_id(1)) || back.existsEdge("e00000003")) bad |= 1;
    for (auto& kv : back.nodes()) {
        const SlamNode& a = g.node(kv.first); const SlamNode& b = kv.second;
        if (a.stamps_ != b.stamps_ || a.fixed_ != b.fixed_ || a.uncertainty_ != b.uncertainty_ || a.edges_ != b.edges_) bad |= 2;
        if (max_abs_diff(a.pose_, b.pose_) > 1e-14 || max_abs_diff(a.sub_pose_, b.sub_pose_) > 1e-14) bad |= 4;
        if (a.sensor_data_.size() != b.sensor_data_.size()) { bad |= 8; continue; }
        for (size_t k = 0; k < a.sensor_data_.size(); k++) {
            const FeatureData& x = *std::dynamic_pointer_cast<FeatureData>(a.sensor_data_[k]);
            const FeatureData& y = *std::dynamic_pointer_cast<FeatureData>(b.sensor_data_[k]);
            if (x.rows != y.rows || x.features_ != y.features_ || x.feature_positions_ != y.feature_positions_ || x.valid_3d_ != y.valid_3d_ ||
                x.feature_positions_2d_ != y.feature_positions_2d_ || x.sensor_frame_ != y.sensor_frame_ || x.stamp_ != y.stamp_ ||
                x.feature_type_ != y.feature_type_ || x.type_ != y.type_ || max_abs_diff(x.displacement_, y.displacement_) > 1e-14) bad |= 16;
        }
    }
    for (auto& kv : back.edges()) {
        const SlamEdge& a = g.edge(kv.first); const SlamEdge& b = kv.second;
        if (a.id_from_ != b.id_from_ || a.id_to_ != b.id_to_ || a.type_ != b.type_ || a.sensor_from_ != b.sensor_from_ || a.sensor_to_ != b.sensor_to_ ||
            a.age_ != b.age_ || a.error_ != b.error_ || a.matching_score_ != b.matching_score_ || a.valid_ != b.valid_ || a.information_ != b.information_ ||
            std::fabs(a.diff_time_ - b.diff_time_) > 1e-9) bad |= 32;
        if (max_abs_diff(a.transform_, b.transform_) > 1e-14 || max_abs_diff(a.displacement_from_, b.displacement_from_) > 1e-14) bad |= 64;
    }
    FILE* o = fopen(out, "wb");
    int32_t rep[4] = {bad, (int32_t)back.nodes().size(), (int32_t)back.edges().size(), uzl_match_frame_count(est)};
    fwrite(rep, 4, 4, o);
    // the loaded node n00000002's first frame, for the Python side to compare with what it reads from the same file
    const FeatureData& fd = *std::dynamic_pointer_cast<FeatureData>(back.node(node_id(2)).sensor_data_[0]);
    int32_t rows = fd.rows; fwrite(&rows, 4, 1, o);
    fwrite(fd.features_.data(), 1, fd.features_.size(), o); fwrite(fd.feature_positions_.data(), 8, fd.feature_positions_.size(), o);
    fclose(o);
    uzl_match_destroy(est);
    return bad ? 6 : 0;
}

// ---- "grow" mode: an online session in two steps.  The graph section of in.bin (same layout as the default mode) is split at node n1:
// the optimizer first gets the nodes below n1 and the edges among them (addGraph + optimize + store), then the whole graph - which it must
// recognise as GROWN ONLY and send through uzl_pgo_append_graph - and the result must be the one a second optimizer gets from a full
// rebuild of the same SlamGraph state.  Prints "GROW_OK <max |pose difference|>".
static void solve_sync(Mi355xOptimizer& opt, SlamGraph& g)
{
    std::mutex m; std::condition_variable cv; bool done = false;
    if (!opt.optimize(g, [&] { std::lock_guard<std::mutex> l(m); done = true; cv.notify_all(); })) { fprintf(stderr, "optimize refused\n"); exit(3); }
    { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return done; }); }
    opt.storeOptimizationResults(g);
}
static int grow_mode(const char* in, int n1)
{
    FILE* f = fopen(in, "rb");
    if (!f) return 2;
    int32_t n, e, iters, xy;
    rd(f, &n, 4); rd(f, &e, 4); rd(f, &iters, 4); rd(f, &xy, 4);
    std::vector<SlamNode> nodes(n); std::vector<SlamEdge> edges(e); std::vector<int32_t> efrom(e), eto(e);
    for (int i = 0; i < n; i++) { int32_t fixed; nodes[i].id_ = node_id(i); rd(f, nodes[i].pose_.m.data(), 96); rd(f, &fixed, 4); nodes[i].fixed_ = fixed != 0; }
    for (int k = 0; k < e; k++) {
        SlamEdge& ed = edges[k]; int32_t type, valid;
        rd(f, &efrom[k], 4); rd(f, &eto[k], 4); rd(f, &type, 4); rd(f, &valid, 4);
        ed.id_from_ = node_id(efrom[k]); ed.id_to_ = node_id(eto[k]); ed.type_ = (unsigned char)type; ed.valid_ = valid != 0;
        rd(f, ed.transform_.m.data(), 96); rd(f, ed.information_.data(), 288);
    }
    fclose(f);
    // edge ids in the order they enter the graph: the old edges must sort in front of the new ones (time-ordered ids, graph_slam_node.cpp:294)
    int seq = 0;
    SlamGraph graph;
    for (int i = 0; i < n1; i++) graph.addNode(nodes[i]);
    for (int k = 0; k < e; k++) if (efrom[k] < n1 && eto[k] < n1) { char b[32]; snprintf(b, sizeof(b), "e%08d", seq++); edges[k].id_ = b; graph.addEdge(edges[k]); }
    GraphOptimizerConfig cfg; cfg.iterations = iters; cfg.optimize_xy_only = xy != 0;
    Mi355xOptimizer opt(0, /*use_edge_filter=*/false);
    opt.setConfig(cfg);
    solve_sync(opt, graph);
    if (opt.lastWasAppend() || opt.lastStatus() < 0) { fprintf(stderr, "first solve: append %d status %d\n", (int)opt.lastWasAppend(), opt.lastStatus()); return 3; }
    for (int i = n1; i < n; i++) graph.addNode(nodes[i]);
    int flipped = 0;
    for (int k = 0; k < e; k++) {
        if (efrom[k] < n1 && eto[k] < n1) {        // the filter changes its mind about a few old feature edges
            if (edges[k].type_ != TYPE_2D_WHEEL_ODOMETRY && flipped < 5 && k % 7 == 0) { SlamEdge& ge = graph.edge(edges[k].id_); ge.valid_ = !ge.valid_; flipped++; }
        } else { char b[32]; snprintf(b, sizeof(b), "e%08d", seq++); edges[k].id_ = b; graph.addEdge(edges[k]); }
    }
    SlamGraph copy = graph;                       // the state a full rebuild starts from
    solve_sync(opt, graph);
    if (!opt.lastWasAppend() || opt.lastStatus() < 0) { fprintf(stderr, "second solve: append %d status %d\n", (int)opt.lastWasAppend(), opt.lastStatus()); return 3; }
    Mi355xOptimizer fresh(0, /*use_edge_filter=*/false);
    fresh.setConfig(cfg);
    solve_sync(fresh, copy);
    if (fresh.lastWasAppend() || fresh.lastStatus() < 0) return 3;
    double worst = 0.;
    for (auto& kv : graph.nodes()) {
        const SlamNode& o = copy.node(kv.first);
        for (int q = 0; q < 12; q++) worst = std::max(worst, std::fabs(kv.second.pose_.m[q] - o.pose_.m[q]));
    }
    if (opt.lastStats().n_edges != fresh.lastStats().n_edges || opt.lastStats().iterations_done != fresh.lastStats().iterations_done) return 4;
    // a solve of the unchanged graph grows nothing: still the append path, with nothing to send
    solve_sync(opt, graph);
    if (!opt.lastWasAppend()) return 5;
    // moving an old node from outside makes it a full rebuild again
    graph.node(node_id(1)).pose_.m[3] += 0.01;
    solve_sync(opt, graph);
    if (opt.lastWasAppend()) return 6;
    // an old edge rewritten in place (mergeNodes moves displacement_from_ under the same edge id, graph_slam_node.cpp:947-976) with
    // ids, flags and poses untouched: the tail alone would leave the resident graph behind - full rebuild, same result as a fresh handle
    solve_sync(opt, graph);                       // (poses written back: the graph is "grown only" again)
    if (!opt.lastWasAppend()) return 7;
    for (int k = 0; k < e; k++)
        if (edges[k].type_ != TYPE_2D_WHEEL_ODOMETRY) { graph.edge(edges[k].id_).displacement_from_.m[3] += 0.05; break; }
    SlamGraph copy2 = graph;
    solve_sync(opt, graph);
    if (opt.lastWasAppend() || opt.lastStatus() < 0) { fprintf(stderr, "rewritten edge: append %d status %d\n", (int)opt.lastWasAppend(), opt.lastStatus()); return 8; }
    Mi355xOptimizer fresh2(0, /*use_edge_filter=*/false);
    fresh2.setConfig(cfg);
    solve_sync(fresh2, copy2);
    for (auto& kv : graph.nodes()) {
        const SlamNode& o = copy2.node(kv.first);
        for (int q = 0; q < 12; q++) worst = std::max(worst, std::fabs(kv.second.pose_.m[q] - o.pose_.m[q]));
    }
    printf("GROW_OK %.3e flipped %d\n", worst, flipped);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc == 4 && std::string(argv[1]) == "grow") return grow_mode(argv[2], atoi(argv[3]));
    if (argc == 4 && std::string(argv[1]) == "storage") return storage_mode(argv[2], argv[3]);
    if (argc == 4 && std::string(argv[1]) == "filter") return filter_mode(argv[2], argv[3]);
    if (argc < 3) { fprintf(stderr, "usage: adapter_selftest in.bin out.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    // ---- graph section
    int32_t n, e, iters, xy;
    rd(f, &n, 4); rd(f, &e, 4); rd(f, &iters, 4); rd(f, &xy, 4);
    SlamGraph graph;
    for (int i = 0; i < n; i++) {
        SlamNode nd; nd.id_ = node_id(i);
        int32_t fixed; rd(f, nd.pose_.m.data(), 96); rd(f, &fixed, 4); nd.fixed_ = fixed != 0;
        graph.addNode(nd);
    }
    for (int k = 0; k < e; k++) {
        SlamEdge ed; char b[32]; snprintf(b, sizeof(b), "e%08d", k); ed.id_ = b;
        int32_t from, to, type, valid;
        rd(f, &from, 4); rd(f, &to, 4); rd(f, &type, 4); rd(f, &valid, 4);
        ed.id_from_ = node_id(from); ed.id_to_ = node_id(to); ed.type_ = (unsigned char)type; ed.valid_ = valid != 0;
        rd(f, ed.transform_.m.data(), 96); rd(f, ed.information_.data(), 288);
        graph.addEdge(ed);
    }
    // ---- pair section
    int32_t n_pairs, nkp, bytes;
    rd(f, &n_pairs, 4); rd(f, &nkp, 4); rd(f, &bytes, 4);
    std::vector<SlamNode> from_nodes(n_pairs), to_nodes(n_pairs);
    for (int j = 0; j < n_pairs; j++) {
        for (int side = 0; side < 2; side++) {
            FeatureDataPtr fd(new FeatureData());
            fd->sensor_frame_ = "camera"; fd->feature_type_ = FEATURE_ORB; fd->rows = nkp; fd->bytes_per_row = bytes;
            fd->features_.resize((size_t)nkp * bytes); fd->feature_positions_.resize((size_t)nkp * 3);
            std::vector<uint8_t> v(nkp);
            rd(f, fd->features_.data(), fd->features_.size()); rd(f, fd->feature_positions_.data(), 24 * (size_t)nkp); rd(f, v.data(), nkp);
            fd->valid_3d_.assign(v.begin(), v.end());
            SlamNode& nd = side == 0 ? from_nodes[j] : to_nodes[j];
            char b[32]; snprintf(b, sizeof(b), "%c%06d", side == 0 ? 'f' : 't', j); nd.id_ = b;
            nd.sensor_data_.push_back(fd);
        }
    }
    fclose(f);

    FILE* o = fopen(argv[2], "wb");
    // ---- optimizer plugin: optimize() -> callback on the worker thread -> storeOptimizationResults()
    {
        Mi355xOptimizer opt(0, /*use_edge_filter=*/false);      // this section feeds ready-made verdicts in SlamEdge::valid_
        GraphOptimizerConfig cfg; cfg.iterations = iters; cfg.optimize_xy_only = xy != 0;
        opt.setConfig(cfg);
        std::mutex m; std::condition_variable cv; bool done = false;
        const bool accepted = opt.optimize(graph, [&] { std::lock_guard<std::mutex> l(m); done = true; cv.notify_all(); });
        const bool second = opt.optimize(graph, [] {});         // must be refused while the first is in flight... or accepted after
        { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return done; }); }
        opt.storeOptimizationResults(graph);
        int32_t hdr[4] = {accepted ? 1 : 0, second ? 1 : 0, opt.lastStatus(), opt.lastStats().iterations_done};
        fwrite(hdr, 4, 4, o);
        double chi[2] = {opt.lastStats().chi2_initial, opt.lastStats().chi2_final};
        fwrite(chi, 8, 2, o);
        for (auto& kv : graph.nodes()) { fwrite(kv.second.pose_.m.data(), 8, 12, o); int32_t op = kv.second.optimized_; fwrite(&op, 4, 1, o); }
        for (auto& kv : graph.edges()) { fwrite(&kv.second.error_, 8, 1, o); fwrite(&kv.second.age_, 8, 1, o); }
        if (second) {   // the refused/accepted second solve must finish before the optimizer is destroyed
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
    }
    // ---- estimator plugin: estimateEdge() x n_pairs -> one callback per pair on the worker thread
    {
        std::mutex m; std::condition_variable cv; std::vector<SlamEdge> got;
        Mi355xFeatureTransformationEstimator est([&](SlamEdge e) { std::lock_guard<std::mutex> l(m); got.push_back(e); cv.notify_all(); }, 0, 777);
        FeatureLinkEstimationConfig c; c.ransac_threshold = 0.1; c.ransac_iteration = 100; c.ransac_break_percentage = 0.6;
        est.setConfig(c);
        for (int j = 0; j < n_pairs; j++) est.estimateEdge(from_nodes[j], to_nodes[j]);
        { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return (int)got.size() == n_pairs; }); }
        int32_t cnt = (int32_t)got.size(); fwrite(&cnt, 4, 1, o);
        for (auto& e : got) {
            int32_t j = atoi(e.id_from_.c_str() + 1); fwrite(&j, 4, 1, o);
            fwrite(&e.matching_score_, 8, 1, o); fwrite(e.transform_.m.data(), 8, 12, o); fwrite(e.information_.data(), 8, 36, o);
            int32_t ty = e.type_; fwrite(&ty, 4, 1, o);
        }
    }
    fclose(o);
    return 0;
}
