// gate_types.hpp — POD shared by gate_kernels.hip and uzl_gate.hip
#pragma once
#include <cstdint>
#include "../../include/uzl_mi355x.h"

namespace uzl {

constexpr int kGateBlock = 64;

struct GateHeapEnt { double w; int32_t v; int32_t pad; };

struct GateArgs {
    int32_t n;                      // nodes
    int32_t n_query;                // candidates in this launch
    const double* poses;            // [n][12]
    const int32_t* adj_ptr;         // [n+1]  neighbours over VALID non-laser edges (getNeighbors(v, true))
    const int32_t* adj_nbr;         // [adj_ptr[n]]
    const uzl_gate_edge* cand;      // [n_query]
    const uint8_t* run;             // [n_query] 1 = candidate passed the index checks, evaluate it
    // per-query scratch
    double* gs;                     // [n_query][n]
    uint8_t* st;                    // [n_query][n]  0 none, 1 open, 2 closed (zeroed by the host before the launch)
    GateHeapEnt* heap;              // [n_query][heap_cap]
    int32_t heap_cap;
    // thresholds
    double min_score, max_T, max_R, ssf;
    // results
    uint8_t* pre_ok;                // score and transform thresholds passed (graph_slam_node.cpp:798-803)
    uint8_t* heur_ok;               // checkEdgeHeuristic (:1064-1085)
    double* dist;                   // astar path length, DBL_MAX = not reachable, -1 = not searched
    int32_t* overflow;              // set when a heap ran out of space
};

}  // namespace uzl
