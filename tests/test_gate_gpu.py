"""GPU parity of the edge acceptance gate (uzl_gate_*: newEdgeCallback / checkEdgeHeuristic / astar,
graph_slam_node.cpp:779-829,1064-1085, slam_graph.cpp:843-890) against the CPU oracle: identical accept / valid
verdicts and bit-identical search distances, candidate by candidate."""
import numpy as np
import pytest

from uzliti_slam_amd import synth
from test_oracle_gate import cand, chain_edges, poses_at

pytestmark = pytest.mark.gpu
DMAX = np.finfo(np.float64).max


def scenario(capi, n, e, seed, n_cand, valid_frac=0.5):
    g = synth.make_pose_graph(n, e, seed=seed)
    ed = g["edges"]
    rng = np.random.default_rng(seed + 100)
    valid = np.where(ed["type"] == synth.EDGE_TYPE_ODOM, 1, (rng.random(len(ed["type"])) < valid_frac).astype(int))
    graph_edges = capi.gate_edges(ed["from"], ed["to"], ed["type"], valid=valid)
    gt = g["gt_pose"].reshape(n, 3, 4)
    # candidates: pairs of nodes within 2 m (some far apart in the graph), transform = noisy relative pose, mixed scores / types
    a = rng.integers(0, n, 4 * n_cand); b = rng.integers(0, n, 4 * n_cand)
    close = np.linalg.norm(gt[a][:, :, 3] - gt[b][:, :, 3], axis=1) < 2.0
    a, b = a[close & (a != b)][:n_cand], b[close & (a != b)][:n_cand]
    rel = synth.se3_mul(synth.se3_inv(gt[a]), gt[b])
    rel = synth.se3_mul(rel, synth.se3_from_noise(rng.normal(0, 0.05, (len(a), 3)), rng.normal(0, 0.02, (len(a), 3))))
    c = capi.gate_edges(a, b, rng.choice([1, 1, 1, 3], len(a)), score=rng.uniform(5, 120, len(a)), transform=rel.reshape(-1, 12))
    # some exact duplicates / reversed duplicates inside the batch
    dup = c[: len(c) // 10].copy()
    dup["from"], dup["to"] = c["to"][: len(dup)].copy(), c["from"][: len(dup)].copy()
    c = np.concatenate([c, dup])
    merged = (rng.random(n) < 0.02).astype(np.uint8)
    return g["nodes_pose"], graph_edges, merged, c


@pytest.mark.parametrize("n,e,seed,n_cand,cfg", [
    (300, 1200, 5, 400, dict()),
    (1000, 5000, 6, 700, dict()),                                              # more candidates than one launch chunk
    (400, 1500, 7, 300, dict(min_accept_valid=60.0)),                          # accepted edges become valid: re-search path
    (200, 700, 8, 200, dict(min_matching_score=40.0, max_edge_distance_T=1.5, max_edge_distance_R=35.0, scope_size_factor=0.3)),
])
def test_gate_equals_oracle(capi, oracle, n, e, seed, n_cand, cfg):
    P, E, merged, c = scenario(capi, n, e, seed, n_cand)
    g = capi.Gate(**cfg); o = oracle.Gate(**cfg)
    g.set_graph(P, E, merged); o.set_graph(P, E, merged)
    ag, vg, dg = g.check(c)
    ao, vo, do = o.check(c)
    assert np.array_equal(ag, ao) and np.array_equal(vg, vo)
    assert dg.tobytes() == do.tobytes()                                       # search distances bit-identical
    assert g.edge_count() == o.edge_count()
    assert 0.05 * len(c) < ag.sum() < 0.95 * len(c)                            # the scenario exercises both verdicts
    assert (dg == DMAX).sum() >= 0 and (dg > 0).sum() > 10
    # a second call sees the edges accepted by the first one (existsEdge)
    ag2, _, _ = g.check(c)
    ao2, _, _ = o.check(c)
    assert np.array_equal(ag2, ao2) and ag2.sum() == 0
    g.close(); o.close()


@pytest.mark.parametrize("n,e,seed,n_cand,cfg", [
    (300, 1200, 5, 400, dict()),
    (1000, 5000, 6, 700, dict()),
    (400, 1500, 7, 300, dict(min_accept_valid=60.0)),
    (200, 700, 8, 200, dict(min_matching_score=40.0, max_edge_distance_T=1.5, max_edge_distance_R=35.0, scope_size_factor=0.3)),
    (300, 1200, 9, 300, dict(scope_size_factor=0.0)),                          # the tests do not depend on the path length at all
])
def test_verdicts_without_path_lengths(capi, oracle, n, e, seed, n_cand, cfg):
    """astar_dist = NULL: uzl_gate_check may skip every search whose verdict the straight-line distance between the two nodes decides
    (checkEdgeHeuristic is monotone in the path length, and no path is shorter than the straight line).  The verdicts - and the graph
    the later candidates of the call see - must be the reference's all the same."""
    P, E, merged, c = scenario(capi, n, e, seed, n_cand)
    g = capi.Gate(**cfg); o = oracle.Gate(**cfg)
    g.set_graph(P, E, merged); o.set_graph(P, E, merged)
    ag, vg, dg = g.check(c, want_dist=False)
    ao, vo, do = o.check(c)
    assert dg is None
    assert np.array_equal(ag, ao) and np.array_equal(vg, vo)
    assert g.edge_count() == o.edge_count()
    ag2, _, _ = g.check(c, want_dist=False)
    assert ag2.sum() == 0
    g.close(); o.close()


def test_gate_known_answers_on_gpu(capi, oracle):
    n = 30
    xyz = np.stack([0.3 * np.arange(n), np.zeros(n), np.zeros(n)], 1)
    c = np.concatenate([cand(0, 2, score=19.9), cand(0, 2, score=20.0), cand(2, 0, score=80.0), cand(0, 2, score=80.0, typ=3),
                        cand(3, 5, t=(1.0001, 0, 0)), cand(3, 5, t=(1.0, 0, 0)), cand(6, 8, yaw_deg=20.5), cand(6, 8, yaw_deg=19.5),
                        cand(0, 1, typ=synth.EDGE_TYPE_ODOM), cand(40, 2), cand(-1, 2), cand(0, 10), cand(10, 13)])
    g = capi.Gate()
    g.set_graph(poses_at(xyz), chain_edges(n))
    acc, val, dist = g.check(c)
    assert list(acc) == [0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 0, 1]
    assert dist[1] == 0.6 and abs(dist[11] - 3.0) < 1e-12 and dist[0] == -1
    g2 = capi.Gate()
    g2.set_graph(poses_at(xyz), chain_edges(n, valid=0))
    acc, _, dist = g2.check(cand(0, 10))
    assert list(acc) == [1] and dist[0] == DMAX
    # empty inputs
    acc, _, _ = g2.check(c[:0])
    assert len(acc) == 0
    with pytest.raises(capi.UzlError):
        capi.Gate(device=99)


def _counts(capi, g):
    import ctypes as C
    w = C.c_int64(); l = C.c_int64()
    capi.lib().uzl_gate_search_counts(g._h, C.byref(w), C.byref(l))
    return w.value, l.value


def _verdicts_only(capi, P, E, c, accept_ref, cfg, oracle=None):
    """uzl_gate_check without astar_dist (straight-line shortcut, deciding search, then the reference's search) on the same scenario."""
    if accept_ref is None:
        O = oracle.Gate(**cfg); O.set_graph(P, E); accept_ref = O.check(c)[0]; O.close()
    G = capi.Gate(**cfg)
    G.set_graph(P, E)
    a, _, d = G.check(c, want_dist=False)
    assert d is None and np.array_equal(a, accept_ref)
    G.close()


def test_wave_search_on_long_chains_multi_edges_and_overflow(capi, oracle):
    """The wave-per-candidate search (open list in LDS, neighbours in parallel lanes) against the CPU checker where it differs from
    the lane kernel structurally: paths of thousands of hops along the odometry chain (an online run's shape), a node with more
    neighbours than a node record holds, multi-edges (the same neighbour listed twice), and a graph dense enough that an open
    list outgrows LDS and the search is redone by the lane kernel."""
    rng = np.random.default_rng(3)
    # (a) chain-like: 6000 nodes, few valid loop closures; candidates between nodes thousands of hops apart
    g = synth.make_pose_graph(6000, 6300, seed=11)
    ed = g["edges"]
    valid = np.where(ed["type"] == synth.EDGE_TYPE_ODOM, 1, (rng.random(len(ed["type"])) < 0.03).astype(int))     # 9 shortcuts
    E = capi.gate_edges(ed["from"], ed["to"], ed["type"], valid=valid)
    gt = g["gt_pose"].reshape(-1, 3, 4)
    from uzliti_slam_amd.synth import _close_pairs
    pr = _close_pairs(gt[:, :, 3], 0.9, 1500)[:300]
    rel = synth.se3_mul(synth.se3_inv(gt[pr[:, 0]]), gt[pr[:, 1]])
    c = capi.gate_edges(pr[:, 0], pr[:, 1], np.ones(len(pr), int), score=np.full(len(pr), 50.0), transform=rel.reshape(-1, 12))
    G = capi.Gate(max_edge_distance_R=360.0); O = oracle.Gate(max_edge_distance_R=360.0)
    G.set_graph(g["nodes_pose"], E); O.set_graph(g["nodes_pose"], E)
    ag, vg, dg = G.check(c); ao, vo, do = O.check(c)
    assert np.array_equal(ag, ao) and dg.tobytes() == do.tobytes()
    assert (dg[dg < DMAX] > 100.0).sum() > 50                                 # paths of hundreds of hops (0.3 m each) were walked
    w, l = _counts(capi, G)
    assert w > 0 and l == 0                                                  # all of them by the wave kernel
    G.close(); O.close()
    _verdicts_only(capi, g["nodes_pose"], E, c, ao, dict(max_edge_distance_R=360.0))
    # (b) a hub with 40 neighbours + every edge of the graph listed twice (two edge types between the same nodes)
    g = synth.make_pose_graph(400, 1600, seed=12)
    ed = g["edges"]
    hub = np.stack([np.zeros(40, int) + 7, rng.choice(np.arange(20, 400), 40, replace=False)], 1)
    fr = np.concatenate([ed["from"], ed["from"], hub[:, 0]]); to = np.concatenate([ed["to"], ed["to"], hub[:, 1]])
    ty = np.concatenate([ed["type"], np.full(len(ed["type"]), 3), np.ones(40, int)])
    E = capi.gate_edges(fr, to, ty, valid=np.ones(len(fr), int))
    a = rng.integers(0, 400, 300); b = rng.integers(0, 400, 300); keep = a != b
    c = capi.gate_edges(a[keep], b[keep], np.full(keep.sum(), 2), score=np.full(keep.sum(), 50.0))
    G = capi.Gate(max_edge_distance_T=100.0, max_edge_distance_R=360.0); O = oracle.Gate(max_edge_distance_T=100.0, max_edge_distance_R=360.0)
    G.set_graph(g["nodes_pose"], E); O.set_graph(g["nodes_pose"], E)
    ag, vg, dg = G.check(c); ao, vo, do = O.check(c)
    assert np.array_equal(ag, ao) and dg.tobytes() == do.tobytes()
    G.close(); O.close()
    _verdicts_only(capi, g["nodes_pose"], E, c, ao, dict(max_edge_distance_T=100.0, max_edge_distance_R=360.0))
    # (c) dense random graph: frontiers of thousands of nodes -> some open lists overflow LDS and fall back to the lane kernel
    n = 6000
    P = np.tile(np.eye(3, 4).reshape(1, 12), (n, 1)); P[:, [3, 7, 11]] = rng.uniform(-30, 30, (n, 3))
    fr = rng.integers(0, n, 60000); to = rng.integers(0, n, 60000); keep = fr != to
    E = capi.gate_edges(fr[keep], to[keep], np.ones(keep.sum(), int), valid=np.ones(keep.sum(), int))
    far = np.argsort(P[:, 3])                                                 # from one side of the cloud to the other, against the heuristic's grain
    c = capi.gate_edges(far[:64], far[::-1][:64], np.full(64, 2), score=np.full(64, 50.0))
    G = capi.Gate(max_edge_distance_T=100.0, max_edge_distance_R=360.0); O = oracle.Gate(max_edge_distance_T=100.0, max_edge_distance_R=360.0)
    G.set_graph(P, E); O.set_graph(P, E)
    ag, vg, dg = G.check(c); ao, vo, do = O.check(c)
    assert np.array_equal(ag, ao) and dg.tobytes() == do.tobytes()
    G.close(); O.close()
    _verdicts_only(capi, P, E, c, ao, dict(max_edge_distance_T=100.0, max_edge_distance_R=360.0))      # (the deciding search's list overflows its registers here)
    _verdicts_only(capi, P, E, c, None, dict(max_edge_distance_T=100.0, max_edge_distance_R=360.0, scope_size_factor=0.004), oracle)   # most need the greedy search


def test_grown_graph_equals_fresh_handle(capi, oracle):
    """uzl_gate_set_graph recognises a graph that only grew (the last call's edges first, flags possibly changed, a tail of new edges, more
    nodes) and enters only the tail; the verdicts and search distances must be those of a handle that is given the whole graph for the
    first time - and the oracle's.  Calls in between accept candidates (which are NOT part of the next graph unless they come back), flip
    `valid` flags of old edges, and once change an old edge's endpoint (not a growth: the full rebuild must take over)."""
    n, e = 600, 2400
    P, E, merged, c = scenario(capi, n, e, 21, 500)
    P = np.asarray(P).reshape(n, 12)
    rng = np.random.default_rng(5)
    order = np.argsort(np.maximum(E["from"], E["to"]), kind="stable")          # edges in the order their later node came into being
    E = E[order]
    g = capi.Gate(min_accept_valid=60.0)
    cuts = [150, 300, 301, 450, 600]
    for step, nn in enumerate(cuts):
        ne = int(np.searchsorted(np.maximum(E["from"], E["to"]), nn, side="left"))
        Ek = E[:ne].copy()
        flip = rng.random(ne) < 0.03                                             # the filter changed its mind about some old edges
        Ek["valid"] = np.where(flip & (Ek["type"] != synth.EDGE_TYPE_ODOM), 1 - Ek["valid"], Ek["valid"])
        if step == 3:
            Ek["to"][5] = (Ek["to"][5] + 1) % nn if (Ek["to"][5] + 1) % nn != Ek["from"][5] else (Ek["to"][5] + 2) % nn      # an old edge rewritten
        Pk = P[:nn] + rng.normal(0, 1e-3, (nn, 12))                              # a re-optimisation moved every pose a little
        ck = c[(c["from"] < nn) & (c["to"] < nn)][:200]
        g.set_graph(Pk, Ek, merged[:nn])
        fresh = capi.Gate(min_accept_valid=60.0); fresh.set_graph(Pk, Ek, merged[:nn])
        o = oracle.Gate(min_accept_valid=60.0); o.set_graph(Pk, Ek, merged[:nn])
        a1, v1, d1 = g.check(ck); a2, v2, d2 = fresh.check(ck); a3, v3, d3 = o.check(ck)
        assert np.array_equal(a1, a2) and np.array_equal(v1, v2) and d1.tobytes() == d2.tobytes(), step
        assert np.array_equal(a1, a3) and np.array_equal(v1, v3) and d1.tobytes() == d3.tobytes(), step
        assert g.edge_count() == fresh.edge_count() == o.edge_count()
        assert a1.sum() > 0
        fresh.close(); o.close()
    g.close()
