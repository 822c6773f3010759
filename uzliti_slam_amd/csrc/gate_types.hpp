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
    int32_t keep_unrun;             // 1: leave the outputs of candidates with run == 0 alone (they hold gate_wave_kernel's verdicts)
    int32_t skip_decided;           // 1: no search for a candidate whose verdict the straight-line distance already decides (dist = -2)
};

// ---- wave-per-candidate search (gate_wave_kernel) ----
// one 64-byte record per node: what an expansion needs from a node in ONE load
struct GateNodeRec {
    double  px, py, pz;             // translation of the node's pose
    int32_t deg;                    // number of neighbours (valid, non-laser edges)
    int32_t adj;                    // start of the node's neighbour list in adj_nbr (for deg > kGateRecNbr)
    int32_t nbr[8];                 // the first neighbours, in adjacency order
};
constexpr int kGateRecNbr = 8;
constexpr int kGateRecMulti = 1 << 30;   // GateNodeRec::deg flag: the neighbour list names some node twice (multi-edge) - the searches dedupe only then
constexpr int kGateOpenCap = 2048;  // open-list entries held in LDS per candidate; a search that needs more is redone by gate_kernel
struct GateState { double g; int32_t st; int32_t pad; };     // per (candidate, node): g-score and 0 none / 1 open / 2 closed

// ---- the same search with the open list in registers and the per-node state in LDS (gate_reg_kernel) ----
constexpr int kGateCacheBlocks = 32;   // direct-mapped record cache: 32 blocks of kGateBlockNodes consecutive nodes (chain order = index order)
constexpr int kGateBlockShift = 5, kGateBlockNodes = 1 << kGateBlockShift;                     // 32 records = 2 KB per block: one miss per ~32 steps along the chain
constexpr int kGateLdsFixed = kGateCacheBlocks * kGateBlockNodes * 64 + kGateCacheBlocks * 4;  // cache + tags
constexpr int kGateLdsMax = 160 * 1024 - 2048;
inline int gate_lds_bytes(int n) { return kGateLdsFixed + 2 * 4 * ((n + 31) / 32); }        // + closed / open bitmaps (n <= ~500k nodes)

struct GateWaveArgs {
    int32_t n, n_query;
    const double* poses;            // [n][12] (checkEdgeHeuristic needs the rotations)
    const GateNodeRec* rec;         // [n]
    const int32_t* adj_nbr;
    const uzl_gate_edge* cand;
    const uint8_t* run;
    GateState* gst;                 // [n_query][n], zeroed by the host (gate_wave_kernel)
    double* gclosed;                // [n_query][n] g-score of closed nodes (gate_reg_kernel; written before it is read: not zeroed)
    double min_score, max_T, max_R, ssf;
    uint8_t* pre_ok; uint8_t* heur_ok; double* dist;
    uint8_t* redo;                  // [n_query] 1 = open list overflowed: search this candidate with gate_kernel
    int32_t keep_unrun;             // 1: leave the outputs of candidates with run == 0 alone (they hold an earlier launch's verdicts)
    int32_t skip_decided;           // 1: no search for a candidate whose verdict the straight-line distance already decides (dist = -2)
    long long* dbg;                 // diagnostic build: [n_query][4] = expansions, shader clocks, 100 MHz ticks, largest open list (else null)
};

}  // namespace uzl
