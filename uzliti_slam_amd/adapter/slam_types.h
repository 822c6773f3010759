// slam_types.h — minimal stand-ins for the reference's data model, so that the plugin-shaped host classes in
// this directory compile without ROS / Eigen / OpenCV / boost (none exist in this image).  Member names follow
//   graph_slam_common/include/graph_slam_common/slam_node.h:89-107   (SlamNode)
//   graph_slam_common/include/graph_slam_common/slam_edge.h:78-92    (SlamEdge)
//   graph_slam_common/include/graph_slam_common/sensor_data.h:40-70  (SensorData, FeatureData)
//   graph_slam_common/include/graph_slam_common/slam_graph.h         (SlamGraph accessors used by the plugins)
// On a ROS machine these types are the real ones and this header is not used.
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

// The reference's plugin bases take their callbacks as boost::function (transformation_estimator.h:48,66; graph_optimizer.h:35,54).
// Where boost exists the mirror takes the same type, so that a caller's boost::function / boost::bind expression binds without a
// conversion; in this image (no boost) std::function stands in - same call syntax, same copy semantics.
#if defined(__has_include)
#if __has_include(<boost/function.hpp>)
#include <boost/function.hpp>
#define UZL_ADAPTER_HAVE_BOOST_FUNCTION 1
#endif
#endif

namespace uzl_adapter {
#ifdef UZL_ADAPTER_HAVE_BOOST_FUNCTION
template <class Sig> using function = boost::function<Sig>;
#else
template <class Sig> using function = std::function<Sig>;
#endif

// Eigen::Isometry3d stand-in: top three rows of the 4x4 matrix, row-major [R|t]
struct Isometry3d {
    std::array<double, 12> m{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}};
    static Isometry3d Identity() { return Isometry3d(); }
};

// graph_slam_msgs/msg/Edge.msg, Features.msg, SensorData.msg constants
enum { TYPE_3D_FULL = 1, TYPE_3D_ROTATION = 2, TYPE_3D_TRANSLATION = 3, TYPE_3D_GPS = 4, TYPE_2D_FULL = 101, TYPE_2D_ROTATION = 102,
       TYPE_2D_TRANSLATION = 103, TYPE_2D_WHEEL_ODOMETRY = 104, TYPE_2D_LASER = 105 };
enum { FEATURE_BRIEF = 1, FEATURE_ORB = 2, FEATURE_BRISK = 3, FEATURE_FREAK = 4 };
enum { SENSOR_TYPE_UNKNOWN = 0, SENSOR_TYPE_FEATURE = 1, SENSOR_TYPE_DEPTH_IMAGE = 2, SENSOR_TYPE_BINARY_GIST = 3, SENSOR_TYPE_LASERSCAN = 4 };

struct SensorData {
    virtual ~SensorData() = default;
    int type_ = SENSOR_TYPE_FEATURE;
    int64_t stamp_ = 0;                      // ros::Time in the reference (sensor_data.h:45): nanoseconds here
    std::string sensor_frame_;
    Isometry3d displacement_;
};

struct FeatureData : SensorData {
    int feature_type_ = FEATURE_ORB;
    // cv::Mat features_ (CV_8U, rows = keypoints): rows x bytes, row-major
    std::vector<uint8_t> features_;
    int rows = 0, bytes_per_row = 0;
    // Eigen::MatrixXd feature_positions_ (3 x N, column-major)
    std::vector<double> feature_positions_;
    // Eigen::MatrixXd feature_positions_2d_ (2 x N): pixel coordinates, int32 here as they travel in Feature.msg (u, v)
    std::vector<int32_t> feature_positions_2d_;
    std::vector<bool> valid_3d_;
};

typedef std::shared_ptr<SensorData> SensorDataPtr;
typedef std::shared_ptr<FeatureData> FeatureDataPtr;

struct SlamNode {
    std::string id_;
    std::vector<int64_t> stamps_;            // std::vector<ros::Time> in the reference (slam_node.h:90): nanoseconds here
    Isometry3d pose_;
    Isometry3d sub_pose_;                    // odometry pose (slam_node.h:91), Node.msg odom_pose
    std::vector<SensorDataPtr> sensor_data_;
    bool fixed_ = false;
    bool optimized_ = false;
    double uncertainty_ = 0;
    std::set<std::string> edges_;            // ids of the edges at this node (slam_node.h:104)
};

struct SlamEdge {
    std::string id_, id_from_, id_to_;
    Isometry3d transform_, displacement_from_, displacement_to_;
    std::array<double, 36> information_{{1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1}};
    unsigned char type_ = 0;
    std::string sensor_from_, sensor_to_;
    double age_ = 0, error_ = 0, matching_score_ = 0;
    double diff_time_ = 0;                   // ros::Duration in the reference (slam_edge.h): seconds here
    bool valid_ = false;
};

// The accessors the two plugin families use (slam_graph.h): ordered by id like the reference's std::map
class SlamGraph {
public:
    void addNode(const SlamNode& n) { nodes_[n.id_] = n; }
    void addEdge(const SlamEdge& e) { edges_[e.id_] = e; }
    void addSensor(const std::string& name, const Isometry3d& T) { sensors_[name] = T; }
    bool existsNode(const std::string& id) const { return nodes_.count(id) != 0; }
    bool existsEdge(const std::string& id) const { return edges_.count(id) != 0; }
    SlamNode& node(const std::string& id) { return nodes_.at(id); }
    SlamEdge& edge(const std::string& id) { return edges_.at(id); }
    std::map<std::string, SlamNode>& nodes() { return nodes_; }
    std::map<std::string, SlamEdge>& edges() { return edges_; }
    std::map<std::string, Isometry3d>& sensors() { return sensors_; }
    // meta data (slam_graph.h:165-182; toMetaData / updateMetaData, slam_graph.cpp:592-633)
    Isometry3d& sensorInitial(const std::string& name) { return sensors_initial_[name]; }
    std::map<std::string, Isometry3d>& sensorsInitial() { return sensors_initial_; }
    std::array<double, 6>& odom() { return odometry_parameters_; }
    std::string frame_ = "/map", name_;
    Isometry3d sub_transform_;              // meta.map_transform on load (updateMetaData :625)

private:
    std::map<std::string, SlamNode> nodes_;
    std::map<std::string, SlamEdge> edges_;
    std::map<std::string, Isometry3d> sensors_, sensors_initial_;
    std::array<double, 6> odometry_parameters_{{0, 0, 0, 0, 0, 0}};
};

// dynamic_reconfigure-generated config structs (cfg/GraphOptimizer.cfg:10-12, cfg/FeatureLinkEstimation.cfg:9-13)
struct GraphOptimizerConfig {
    int iterations = 20;
    bool use_odometry_parameters = false;
    bool optimize_xy_only = false;
};
struct FeatureLinkEstimationConfig {
    double ransac_threshold = 0.2;
    double link_covariance = 0.01;
    int ransac_iteration = 100;
    double ransac_break_percentage = 0.6;
    bool use_epnp = true;
};

}  // namespace uzl_adapter
