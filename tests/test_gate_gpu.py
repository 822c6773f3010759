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
