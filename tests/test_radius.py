"""Distance loop-closure candidates (SlamGraph::getNodesWithinRadius + caller's filters, slam_graph.cpp:266-278,
graph_slam_node.cpp:272-289): oracle known answers (CPU) and GPU parity (ordered, exact)."""
import numpy as np
import pytest

from uzliti_slam_amd import synth

S = 10**9


def line(n, step=0.25):
    P = np.tile(np.eye(3, 4).reshape(12), (n, 1))
    P[:, 3] = step * np.arange(n)
    return P, (S * np.arange(n) * 2).astype(np.int64)            # 2 s between nodes


def test_oracle_known_answers(oracle):
    P, st = line(12)
    f, t, cnt = oracle.radius_candidates(P, st, [6], radius=0.5, new_edge_time=5.0)
    # within 0.5 m of node 6 (strict <): nodes 5, 7 (0.25 m); they are only 2 s away -> none
    assert len(f) == 0 and list(cnt) == [0]
    f, t, cnt = oracle.radius_candidates(P, st, [6], radius=0.8, new_edge_time=5.0)
    # 0.75 m: nodes 3 and 9 (6 s apart) qualify; 4, 5, 7, 8 are too recent
    assert list(f) == [3, 9] and list(t) == [6, 6] and list(cnt) == [2]
    f, t, cnt = oracle.radius_candidates(P, st, [6], radius=0.75, new_edge_time=5.0)
    assert len(f) == 0                                            # the distance test is strict: 0.75 is not < 0.75
    # rotation filter: node 3 turned by 31 degrees
    P2 = P.copy()
    P2[3] = synth.se3(synth.quat_to_R(synth.quat_from_rotvec(np.array([[0, 0, np.deg2rad(31.0)]])))[0], P[3, [3, 7, 11]]).reshape(12)
    f, _, _ = oracle.radius_candidates(P2, st, [6], radius=0.8)
    assert list(f) == [9]
    P2[3] = synth.se3(synth.quat_to_R(synth.quat_from_rotvec(np.array([[0, 0, np.deg2rad(29.0)]])))[0], P[3, [3, 7, 11]]).reshape(12)
    assert list(oracle.radius_candidates(P2, st, [6], radius=0.8)[0]) == [3, 9]
    # several queries: jobs ordered by query, then by node; unknown query nodes yield nothing
    f, t, cnt = oracle.radius_candidates(P, st, [9, 50, 3], radius=0.8)
    assert list(zip(f, t)) == [(6, 9), (0, 3), (6, 3)] and list(cnt) == [1, 0, 2]


def _numpy_candidates(P, st, queries, radius, new_edge_time, max_rot):
    P = P.reshape(-1, 3, 4); out = []
    for q in queries:
        if not (0 <= q < len(P)):
            continue
        d = np.linalg.norm(P[:, :, 3] - P[q, :, 3], axis=1)
        for c in np.nonzero(d < radius)[0]:
            if c == q or not abs((st[q] - st[c]) * 1e-9) > new_edge_time:
                continue
            if np.rad2deg(synth.rotation_angle(P[c][:, :3].T @ P[q][:, :3])) < max_rot:
                out.append((int(c), int(q)))
    return out


def test_oracle_equals_numpy(oracle):
    g = synth.make_pose_graph(500, 1500, seed=3)
    P = g["gt_pose"]; st = (S * 0.5 * np.arange(500)).astype(np.int64)
    q = np.arange(0, 500, 7)
    f, t, cnt = oracle.radius_candidates(P, st, q, radius=0.8, new_edge_time=5.0, max_rotation_deg=30.0)
    assert list(zip(f, t)) == _numpy_candidates(P, st, q, 0.8, 5.0, 30.0) and cnt.sum() == len(f) > 50


@pytest.mark.gpu
@pytest.mark.parametrize("n,nq,cfg", [(500, 72, dict(radius=0.8)), (20000, 300, dict(radius=0.5)), (257, 257, dict(radius=1.5, new_edge_time=1.0, max_rotation_deg=12.0)),
                                      (3, 3, dict(radius=10.0, new_edge_time=-1.0))])
def test_gpu_equals_oracle(capi, oracle, n, nq, cfg):
    g = synth.make_pose_graph(n, n + n // 2, seed=n)
    P = g["gt_pose"]; st = (S * 0.5 * np.arange(n)).astype(np.int64)
    rng = np.random.default_rng(n)
    q = rng.choice(n, nq, replace=False).astype(np.int32) if nq < n else np.arange(n, dtype=np.int32)
    q[0] = -5 if n > 3 else q[0]                                    # an unknown node id among the queries
    full = dict(radius=0.5, new_edge_time=5.0, max_rotation_deg=30.0); full.update(cfg)
    f, t, cnt = oracle.radius_candidates(P, st, q, **full)
    r = capi.Radius(**cfg)
    r.set_nodes(P, st)
    gf, gt_, gcnt, tot = r.query(q)
    assert tot == len(f) and np.array_equal(gf, f) and np.array_equal(gt_, t) and np.array_equal(gcnt, cnt)
    if tot > 4:                                                      # a too-small output buffer: total still reported, prefix written
        gf2, gt2, _, tot2 = r.query(q, cap=tot // 2)
        assert tot2 == tot and np.array_equal(gf2, f[: tot // 2]) and np.array_equal(gt2, t[: tot // 2])
    assert len(r.query(np.zeros(0, np.int32))[0]) == 0
    r.close()
