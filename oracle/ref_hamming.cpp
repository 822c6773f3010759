// ref_hamming.cpp — C-ABI shim around the ONE source file of the reference's hot path that compiles without ROS / OpenCV / PCL / g2o:
// graph_slam_common/thirdparty/include/graph_slam_tools/hammingsse.hpp (cv::HammingSse, the SSSE3 popcount-of-XOR functor: :60-139 the
// kernel, :140-160 the functor).  The header is compiled WHERE IT LIES under the reference tree (-I from oracle/Makefile, target `ref`);
// nothing of it is copied here.  Output: oracle/_ref/libref_hamming.so - a known-answer pin for the Hamming primitive of M1
// (feature_transformation_estimator.cpp:38,58 go through OpenCV's own NORM_HAMMING, which this functor replaces elsewhere in the
// reference: lsh.cpp:550).  It pins that primitive only; parity of the path as a whole stays unpinned.
// Test infrastructure: loaded by tests/test_oracle_match.py alone.
#include <graph_slam_tools/hammingsse.hpp>

extern "C" int ref_hamming(const unsigned char* a, const unsigned char* b, int size_bytes)
{
    // (the functor reads 16 bytes at a time through aligned loads: callers pass 16-byte aligned buffers, sizes in multiples of 16)
    return cv::HammingSse()(a, b, size_bytes);
}
