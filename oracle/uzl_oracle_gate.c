/*
 * uzl_oracle_gate.c — CPU ORACLE (test infrastructure, NOT product code), see uzl_oracle.h.
 *
 * Restatement of the caller's edge acceptance gate:
 *   GraphSlamNode::newEdgeCallback      graph_slam/src/graph_slam_node.cpp:779-829
 *   GraphSlamNode::checkEdgeHeuristic   graph_slam/src/graph_slam_node.cpp:1064-1085
 *   SlamGraph::astar / heuristic_cost   graph_slam_common/src/slam_graph.cpp:838-890
 *   SlamGraph::getNeighbors             graph_slam_common/src/slam_graph.cpp:558-578
 *   SlamGraph::existsEdge(from,to,type) graph_slam_common/src/slam_graph.cpp:407-422
 *   SlamGraph::isMerged                 graph_slam_common/src/slam_graph.cpp:206-209
 *
 * PARITY UNPINNED (no reference tests; not buildable here).  Notes:
 *   - astar() pushes nodes with priority heuristic_cost(u, target) only - not g + h - so it is a greedy best-first
 *     search and nodes_[target].distance_ is the length of the path it happens to find (restated as written);
 *   - the reference's fibonacci heap orders equal priorities arbitrarily; equal priorities of DIFFERENT nodes need
 *     equidistant positions (measure zero) -> broken here by node index; duplicates of one node are interchangeable;
 *   - Eigen::AngleAxisd(R).angle() = 2 acos(clamp(w)) of Quaterniond(R) [EXT, Eigen 3.2] (may exceed pi when w < 0);
 *   - node / edge ids are indices here; an edge's position in node.edges_ (a std::set of id strings) only decides the
 *     order neighbours are visited in, which cannot change the result (every neighbour's update is independent).
 */
#include "uzl_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct gedge { int32_t from, to, type, valid; } gedge;

struct uzlo_gate {
    uzlo_gate_cfg cfg;
    int32_t n;
    double* poses;            /* n x 12 */
    uint8_t* merged;          /* n */
    gedge* edges; int32_t ne, cape;
    /* adjacency over all edges (CSR rebuilt lazily) */
    int32_t* adj_ptr; int32_t* adj_edge; int adj_ok;
    int64_t last_expansions;
};

void uzlo_gate_cfg_default(uzlo_gate_cfg* c)
{
    c->min_matching_score = 20.0;      /* GraphSlam.cfg:18 */
    c->max_edge_distance_T = 1.0;      /* :19 */
    c->max_edge_distance_R = 20.0;     /* :20 (degrees) */
    c->scope_size_factor = 0.1;        /* :34 */
    c->min_accept_valid = DBL_MAX;     /* graph_slam_node.cpp:139 */
    c->device = 0; c->pad = 0;
}

uzlo_gate* uzlo_gate_create(const uzlo_gate_cfg* cfg)
{
    uzlo_gate* g = (uzlo_gate*)calloc(1, sizeof(*g));
    if (cfg) g->cfg = *cfg; else uzlo_gate_cfg_default(&g->cfg);
    return g;
}

void uzlo_gate_destroy(uzlo_gate* g)
{
    if (!g) return;
    free(g->poses); free(g->merged); free(g->edges); free(g->adj_ptr); free(g->adj_edge); free(g);
}

static void push_edge(uzlo_gate* g, int32_t from, int32_t to, int32_t type, int32_t valid)
{
    if (g->ne == g->cape) { g->cape = g->cape ? 2 * g->cape : 256; g->edges = (gedge*)realloc(g->edges, sizeof(gedge) * (size_t)g->cape); }
    g->edges[g->ne].from = from; g->edges[g->ne].to = to; g->edges[g->ne].type = type; g->edges[g->ne].valid = valid;
    g->ne++;
    g->adj_ok = 0;
}

void uzlo_gate_set_graph(uzlo_gate* g, int32_t n, const double* poses, const uint8_t* merged, int32_t ne, const uzlo_gate_edge* edges)
{
    g->n = n;
    g->poses = (double*)realloc(g->poses, sizeof(double) * 12 * (size_t)(n > 0 ? n : 1));
    g->merged = (uint8_t*)realloc(g->merged, (size_t)(n > 0 ? n : 1));
    if (n > 0) memcpy(g->poses, poses, sizeof(double) * 12 * (size_t)n);
    for (int32_t i = 0; i < n; i++) g->merged[i] = merged ? merged[i] : 0;
    g->ne = 0;
    for (int32_t k = 0; k < ne; k++)
        if (edges[k].from >= 0 && edges[k].from < n && edges[k].to >= 0 && edges[k].to < n)
            push_edge(g, edges[k].from, edges[k].to, edges[k].type, edges[k].valid ? 1 : 0);
    g->adj_ok = 0;
}

static void build_adj(uzlo_gate* g)
{
    if (g->adj_ok) return;
    g->adj_ptr = (int32_t*)realloc(g->adj_ptr, sizeof(int32_t) * (size_t)(g->n + 2));
    g->adj_edge = (int32_t*)realloc(g->adj_edge, sizeof(int32_t) * (size_t)(2 * g->ne + 1));
    memset(g->adj_ptr, 0, sizeof(int32_t) * (size_t)(g->n + 2));
    for (int32_t k = 0; k < g->ne; k++) { g->adj_ptr[g->edges[k].from + 1]++; if (g->edges[k].to != g->edges[k].from) g->adj_ptr[g->edges[k].to + 1]++; }
    for (int32_t i = 0; i < g->n; i++) g->adj_ptr[i + 1] += g->adj_ptr[i];
    int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * (size_t)(g->n + 1));
    memcpy(fill, g->adj_ptr, sizeof(int32_t) * (size_t)(g->n + 1));
    for (int32_t k = 0; k < g->ne; k++) {
        g->adj_edge[fill[g->edges[k].from]++] = k;
        if (g->edges[k].to != g->edges[k].from) g->adj_edge[fill[g->edges[k].to]++] = k;
    }
    free(fill);
    g->adj_ok = 1;
}

/* heuristic_cost (:838-841): epsilon * ||t_a - t_b||, epsilon = 1 */
static double node_dist(const uzlo_gate* g, int32_t a, int32_t b)
{
    const double* A = g->poses + 12 * (size_t)a; const double* B = g->poses + 12 * (size_t)b;
    const double dx = A[3] - B[3], dy = A[7] - B[7], dz = A[11] - B[11];
    return 1. * sqrt((dx * dx + dy * dy) + dz * dz);
}

typedef struct hent { double w; int32_t v; } hent;
static int hless(const hent* a, const hent* b) { return a->w < b->w || (a->w == b->w && a->v < b->v); }

/* SlamGraph::astar (:843-890); returns DBL_MAX when the target is not reached */
double uzlo_gate_astar(uzlo_gate* g, int32_t source, int32_t target)
{
    if (source < 0 || target < 0 || source >= g->n || target >= g->n) return DBL_MAX;
    build_adj(g);
    double* gs = (double*)malloc(sizeof(double) * (size_t)g->n);
    uint8_t* st = (uint8_t*)calloc((size_t)g->n, 1);          /* 1 = open, 2 = closed */
    int32_t hcap = 64, hn = 0;
    hent* heap = (hent*)malloc(sizeof(hent) * (size_t)hcap);
    int64_t n_open = 0, expansions = 0;
    gs[source] = 0.; st[source] = 1; n_open = 1;
    heap[hn].w = node_dist(g, source, target); heap[hn].v = source; hn++;
    int success = 0;
    while (n_open > 0) {
        const int32_t v = heap[0].v;                          /* f_score.top() */
        if (v == target) { success = 1; break; }
        /* pop */
        heap[0] = heap[--hn];
        for (int32_t i = 0;;) {
            int32_t l = 2 * i + 1, r = l + 1, m = i;
            if (l < hn && hless(&heap[l], &heap[m])) m = l;
            if (r < hn && hless(&heap[r], &heap[m])) m = r;
            if (m == i) break;
            hent t = heap[i]; heap[i] = heap[m]; heap[m] = t; i = m;
        }
        if (st[v] == 1) n_open--;
        st[v] = 2;
        expansions++;
        for (int32_t q = g->adj_ptr[v]; q < g->adj_ptr[v + 1]; q++) {                   /* getNeighbors(v, true) */
            const gedge* e = &g->edges[g->adj_edge[q]];
            if (!e->valid || e->type == 105 /* TYPE_2D_LASER, Edge.msg:9 */) continue;
            const int32_t u = (e->from == v) ? e->to : e->from;
            if (st[u] == 2) continue;
            const double tent = gs[v] + node_dist(g, v, u);
            if (st[u] != 1 || tent < gs[u]) {
                gs[u] = tent;
                if (hn == hcap) { hcap *= 2; heap = (hent*)realloc(heap, sizeof(hent) * (size_t)hcap); }
                int32_t i = hn++;
                heap[i].w = node_dist(g, u, target); heap[i].v = u;
                while (i > 0) { int32_t p = (i - 1) / 2; if (!hless(&heap[i], &heap[p])) break; hent t = heap[i]; heap[i] = heap[p]; heap[p] = t; i = p; }
                if (st[u] != 1) { st[u] = 1; n_open++; }
            }
        }
    }
    const double res = success ? gs[target] : DBL_MAX;
    g->last_expansions = expansions;
    free(gs); free(st); free(heap);
    return res;
}

/* Eigen::AngleAxisd(R).angle() [EXT]: Quaterniond(R), then 2 acos(clamp(w)) unless the vector part vanishes */
static double angle_of(const double R[9])
{
    double q[4];
    uzlo_quat_from_R(R, q);
    const double n2 = (q[1] * q[1] + q[2] * q[2]) + q[3] * q[3];
    if (n2 < 1e-12 * 1e-12) return 0.;                      /* NumTraits<double>::dummy_precision()^2 */
    double w = q[0];
    if (w < -1.) w = -1.;
    if (w > 1.) w = 1.;
    return 2. * acos(w);
}

static int exists_edge(uzlo_gate* g, int32_t from, int32_t to, int32_t type)      /* :407-422 */
{
    build_adj(g);
    for (int32_t q = g->adj_ptr[from]; q < g->adj_ptr[from + 1]; q++) {
        const gedge* e = &g->edges[g->adj_edge[q]];
        if (((e->from == from && e->to == to) || (e->from == to && e->to == from)) && e->type == type) return 1;
    }
    return 0;
}

/* newEdgeCallback for each candidate in order; accepted edges join the graph (:812) */
void uzlo_gate_check(uzlo_gate* g, int32_t nc, const uzlo_gate_edge* cand, uint8_t* accept, uint8_t* valid, double* astar_dist)
{
    const double ssf = g->cfg.scope_size_factor;
    for (int32_t k = 0; k < nc; k++) {
        const uzlo_gate_edge* c = &cand[k];
        accept[k] = 0; if (valid) valid[k] = 0; if (astar_dist) astar_dist[k] = -1.;
        if (c->from < 0 || c->to < 0 || c->from >= g->n || c->to >= g->n) continue;
        if (g->merged[c->from] || g->merged[c->to]) continue;                       /* :784-787 */
        if (exists_edge(g, c->from, c->to, c->type)) continue;                      /* :789-791 */
        if (!(c->matching_score >= g->cfg.min_matching_score)) continue;            /* :798 */
        const double* T = c->transform;
        const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
        const double diff_rot = fabs(angle_of(R)) * 180 / M_PI;                     /* :800-801 */
        const double tn = sqrt((T[3] * T[3] + T[7] * T[7]) + T[11] * T[11]);
        if (!(tn <= g->cfg.max_edge_distance_T && diff_rot <= g->cfg.max_edge_distance_R)) continue;   /* :803 */
        /* checkEdgeHeuristic (:1064-1085) */
        const double dist = uzlo_gate_astar(g, c->from, c->to);
        if (astar_dist) astar_dist[k] = dist;
        int ok = 1;
        if (dist != DBL_MAX) {
            const double* A = g->poses + 12 * (size_t)c->from; const double* B = g->poses + 12 * (size_t)c->to;
            /* diff_pose = pose_from^-1 * pose_to: R = Ra^T Rb, t = Ra^T tb + (-(Ra^T ta)) */
            double Rd[9], ti[3], td[3];
            for (int r = 0; r < 3; r++) {
                for (int cc = 0; cc < 3; cc++) Rd[r * 3 + cc] = (A[0 * 4 + r] * B[0 * 4 + cc] + A[1 * 4 + r] * B[1 * 4 + cc]) + A[2 * 4 + r] * B[2 * 4 + cc];
                ti[r] = -((A[0 * 4 + r] * A[3] + A[1 * 4 + r] * A[7]) + A[2 * 4 + r] * A[11]);
            }
            for (int r = 0; r < 3; r++) td[r] = ((A[0 * 4 + r] * B[3] + A[1 * 4 + r] * B[7]) + A[2 * 4 + r] * B[11]) + ti[r];
            const double dn = sqrt((td[0] * td[0] + td[1] * td[1]) + td[2] * td[2]);
            const double drot = 180. * angle_of(Rd) / M_PI;
            ok = (2 * ssf * dist + 1.0 > dn) && (10 * ssf * dist + 30.0 > drot);   /* :1074-1075 */
        }
        if (!ok) continue;
        const int v = c->matching_score >= g->cfg.min_accept_valid;                /* :809-811 */
        push_edge(g, c->from, c->to, c->type, v);                                   /* :812 */
        accept[k] = 1; if (valid) valid[k] = (uint8_t)v;
    }
}

int32_t uzlo_gate_edge_count(const uzlo_gate* g) { return g->ne; }
int64_t uzlo_gate_last_expansions(const uzlo_gate* g) { return g->last_expansions; }
