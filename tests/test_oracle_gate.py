"""CPU tests of the edge-gate oracle (oracle/uzl_oracle_gate.c): GraphSlamNode::newEdgeCallback / checkEdgeHeuristic
(graph_slam_node.cpp:779-829,1064-1085) and SlamGraph::astar (slam_graph.cpp:843-890), against hand-made known answers
and a pure-Python restatement of the search."""
import heapq

import numpy as np

from uzliti_slam_amd import capi, synth

DMAX = np.finfo(np.float64).max


def poses_at(xyz):
    P = np.tile(np.eye(3, 4).reshape(12), (len(xyz), 1))
    P[:, [3, 7, 11]] = np.asarray(xyz, float)
    return P


def chain_edges(n, valid=1):
    return capi.gate_edges(np.arange(n - 1), np.arange(1, n), np.full(n - 1, synth.EDGE_TYPE_ODOM), valid=np.full(n - 1, valid))


def cand(frm, to, score=50.0, t=(0.1, 0.0, 0.0), yaw_deg=0.0, typ=1):
    T = synth.se3(synth.quat_to_R(synth.quat_from_rotvec(np.array([[0.0, 0.0, np.deg2rad(yaw_deg)]])))[0], np.array(t)).reshape(1, 12)
    return capi.gate_edges([frm], [to], [typ], score=[score], transform=T)


def py_astar(pos, adj, s, t):
    """the reference's loop, literally (priority = straight-line distance to the target only)"""
    h = lambda a: float(np.sqrt(((pos[a] - pos[t]) ** 2)[[0, 1]].sum() + (pos[a][2] - pos[t][2]) ** 2))
    d = lambda a, b: float(np.sqrt(((pos[a] - pos[b]) ** 2)[[0, 1]].sum() + (pos[a][2] - pos[b][2]) ** 2))
    g = {s: 0.0}; open_ = {s}; closed = set(); heap = [(h(s), s)]
    while open_:
        w, v = heap[0]
        if v == t:
            return g[t]
        heapq.heappop(heap); open_.discard(v); closed.add(v)
        for u in adj[v]:
            if u in closed:
                continue
            tent = g[v] + d(v, u)
            if u not in open_ or tent < g[u]:
                g[u] = tent; heapq.heappush(heap, (h(u), u)); open_.add(u)
    return DMAX


def test_astar_is_greedy_best_first(oracle):
    # 0 -> target 3.  Short route 0-1-3 starts by moving AWAY from the target; the long route 0-2-4-3 moves towards it first.
    pos = np.array([[0, 0, 0], [-1, 0.5, 0], [1, 2, 0], [2, 0, 0], [2.5, 2.5, 0]], float)
    E = capi.gate_edges([0, 1, 0, 2, 4], [1, 3, 2, 4, 3], [synth.EDGE_TYPE_ODOM] * 5, valid=[1] * 5)
    g = oracle.Gate()
    g.set_graph(poses_at(pos), E)
    d = g.astar(0, 3)
    short = np.linalg.norm(pos[1] - pos[0]) + np.linalg.norm(pos[3] - pos[1])
    long_ = np.linalg.norm(pos[2] - pos[0]) + np.linalg.norm(pos[4] - pos[2]) + np.linalg.norm(pos[3] - pos[4])
    # priority = distance to the target only: 2 (2.0 away) is expanded before 1 (3.04 away), then 4, which reaches the target
    assert abs(d - long_) < 1e-12 and d > short + 2.0
    adj = {0: [1, 2], 1: [0, 3], 2: [0, 4], 3: [1, 4], 4: [2, 3]}
    assert d == py_astar(pos, adj, 0, 3)


def test_astar_path_need_not_be_shortest(oracle):
    # the search commits to the neighbour closest to the target; the target is then reached through it although a shorter path exists
    pos = np.array([[0, 0, 0], [1, 0.1, 0], [0.2, -0.5, 0], [2, 0, 0], [1.8, 1.5, 0]], float)
    # edges: 0-1, 1-4, 4-3 (detour via 4), and 0-2, 2-3 (direct but 2 is farther from the target than 1)
    E = capi.gate_edges([0, 1, 4, 0, 2], [1, 4, 3, 2, 3], [synth.EDGE_TYPE_ODOM] * 5, valid=[1] * 5)
    g = oracle.Gate()
    g.set_graph(poses_at(pos), E)
    adj = {0: [1, 2], 1: [0, 4], 4: [1, 3], 2: [0, 3], 3: [4, 2]}
    d = g.astar(0, 3)
    assert d == py_astar(pos, adj, 0, 3)
    direct = np.linalg.norm(pos[2] - pos[0]) + np.linalg.norm(pos[3] - pos[2])
    assert d >= direct - 1e-12


def test_astar_ignores_invalid_and_laser_edges_and_unreachable(oracle):
    pos = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0]], float)
    g = oracle.Gate()
    g.set_graph(poses_at(pos), capi.gate_edges([0, 1], [1, 2], [synth.EDGE_TYPE_ODOM, synth.EDGE_TYPE_3D_FULL], valid=[1, 0]))       # 1-2 not valid
    assert g.astar(0, 1) == 1.0 and g.astar(0, 2) == DMAX
    g.set_graph(poses_at(pos), capi.gate_edges([0, 1], [1, 2], [synth.EDGE_TYPE_ODOM, synth.EDGE_TYPE_2D_LASER], valid=[1, 1]))       # 1-2 is TYPE_2D_LASER
    assert g.astar(0, 2) == DMAX
    assert g.astar(1, 1) == 0.0


def test_astar_equals_python_restatement_on_random_graphs(oracle):
    for seed in range(5):
        gph = synth.make_pose_graph(120, 420, seed=seed)
        e = gph["edges"]
        rng = np.random.default_rng(seed)
        valid = np.where(e["type"] == synth.EDGE_TYPE_ODOM, 1, (rng.random(len(e["type"])) < 0.5).astype(int))
        P = gph["nodes_pose"]
        pos = P[:, [3, 7, 11]]
        adj = {i: [] for i in range(120)}
        for a, b, v in zip(e["from"], e["to"], valid):
            if v:
                adj[int(a)].append(int(b)); adj[int(b)].append(int(a))
        g = oracle.Gate()
        g.set_graph(P, capi.gate_edges(e["from"], e["to"], e["type"], valid=valid))
        for _ in range(40):
            s, t = map(int, rng.integers(0, 120, 2))
            assert g.astar(s, t) == py_astar(pos, adj, s, t), (seed, s, t)


def test_gate_thresholds_and_duplicates(oracle):
    n = 30
    xyz = np.stack([0.3 * np.arange(n), np.zeros(n), np.zeros(n)], 1)
    g = oracle.Gate()
    g.set_graph(poses_at(xyz), chain_edges(n))
    c = np.concatenate([
        cand(0, 2, score=19.9),                      # below min_matching_score
        cand(0, 2, score=20.0),                      # accepted (>=)
        cand(2, 0, score=80.0),                      # same pair, same type, other direction: exists already (:789)
        cand(0, 2, score=80.0, typ=3),               # other type: allowed
        cand(3, 5, t=(1.0001, 0, 0)),                # translation above max_edge_distance_T
        cand(3, 5, t=(1.0, 0, 0)),                   # exactly 1.0: accepted (<=)
        cand(6, 8, yaw_deg=20.5),                    # rotation above max_edge_distance_R
        cand(6, 8, yaw_deg=19.5),
        cand(0, 1, typ=synth.EDGE_TYPE_ODOM),        # an odometry edge 0-1 of that type exists
        cand(40, 2), cand(-1, 2),                    # unknown nodes
    ])
    acc, val, dist = g.check(c)
    assert list(acc) == [0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0]
    assert list(val) == [0] * 11                     # min_accept_valid defaults to "never"
    assert dist[1] == 0.6 and dist[0] == -1 and dist[2] == -1 and dist[4] == -1
    assert g.edge_count() == n - 1 + 4


def test_gate_plausibility_and_merged(oracle):
    n = 40
    xyz = np.stack([0.3 * np.arange(n), np.zeros(n), np.zeros(n)], 1)
    # the graph path 0..10 is 3.0 m long: 2*0.1*3.0 + 1.0 = 1.6 m must exceed the pose distance (3.0 m) -> implausible
    g = oracle.Gate()
    g.set_graph(poses_at(xyz), chain_edges(n))
    acc, _, dist = g.check(cand(0, 10))
    assert list(acc) == [0] and abs(dist[0] - 3.0) < 1e-12
    # nodes 0 and 3: path 0.9 m, pose distance 0.9 < 1.18 -> plausible
    acc, _, dist = g.check(cand(0, 3))
    assert list(acc) == [1] and abs(dist[0] - 0.9) < 1e-12
    # not connected through valid edges at all -> accepted (:1077-1079)
    g2 = oracle.Gate()
    g2.set_graph(poses_at(xyz), chain_edges(n, valid=0))
    acc, _, dist = g2.check(cand(0, 10))
    assert list(acc) == [1] and dist[0] == DMAX
    # merged nodes are refused before anything else (:784-787)
    g3 = oracle.Gate()
    merged = np.zeros(n, np.uint8); merged[3] = 1
    g3.set_graph(poses_at(xyz), chain_edges(n), merged=merged)
    assert list(g3.check(np.concatenate([cand(0, 3), cand(3, 5), cand(4, 6)]))[0]) == [0, 0, 1]
    # rotation part of the plausibility test: poses 120 deg apart, path 0.3 m: 10*0.1*0.3 + 30 = 30.3 deg < 120 deg
    P = poses_at(xyz)
    P[1] = synth.se3(synth.quat_to_R(synth.quat_from_rotvec(np.array([[0, 0, np.deg2rad(120.0)]])))[0], xyz[1]).reshape(12)
    g4 = oracle.Gate()
    g4.set_graph(P, chain_edges(n))
    assert list(g4.check(cand(0, 1, typ=1))[0]) == [0]


def test_gate_valid_edges_change_reachability_within_a_batch(oracle):
    """min_accept_valid reachable: an accepted edge is valid at once and shortens later searches (:809-812)."""
    n = 40
    xyz = np.stack([0.3 * np.arange(n), 0.02 * np.arange(n) ** 1.5 % 0.3, np.zeros(n)], 1)
    xyz[20:] = xyz[19] + np.stack([-0.25 * np.arange(1, 21), 0.25 + np.zeros(20), np.zeros(20)], 1)     # the path folds back
    g = oracle.Gate(min_accept_valid=60.0)
    g.set_graph(poses_at(xyz), chain_edges(n))
    c = np.concatenate([cand(18, 21, score=70.0), cand(17, 22, score=30.0), cand(16, 23, score=30.0)])
    acc, val, dist = g.check(c)
    assert list(acc) == [1, 1, 1] and list(val) == [1, 0, 0]
    assert dist[1] < 0.3 * 5 - 1e-9          # 17 -> 18 -> 21 -> 22 through the new valid edge instead of 17 .. 22 along the chain
    g_plain = oracle.Gate()
    g_plain.set_graph(poses_at(xyz), chain_edges(n))
    _, _, dist_plain = g_plain.check(c)
    assert dist_plain[1] > dist[1]
