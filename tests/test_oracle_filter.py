"""CPU tests of the edge-filter oracle (oracle/uzl_oracle_filter.c) against hand-made known answers and against
the pure-Python restatement in np_reference.FilterRef (TransformationFilter, transformation_filter.cpp:43-350)."""
import numpy as np

from uzliti_slam_amd import synth
from filter_common import assert_same_state, play
from np_reference import FilterRef

S = 10**9


def E(key, tf, tt, valid=0, score=1.0, **kw):
    d = dict(key=key, matching_score=score, valid=valid, sensor_from=-1, sensor_to=-1,
             stamps_from=np.atleast_1d(np.array(tf, np.int64)), stamps_to=np.atleast_1d(np.array(tt, np.int64)),
             transform=np.eye(3, 4).reshape(12), displacement_from=np.eye(3, 4).reshape(12),
             displacement_to=np.eye(3, 4).reshape(12), pose_from=np.eye(3, 4).reshape(12), pose_to=np.eye(3, 4).reshape(12))
    d.update(kw)
    return d


def test_cluster_window_and_bounds(oracle):
    f = oracle.Filter(max_dt=5.0)
    f.add([E(1, 100 * S, 200 * S)])
    f.add([E(2, 104 * S, 203 * S)])              # inside +-5 s of both bounds -> same cluster
    f.add([E(3, 105 * S + 4 * S, 203 * S)])      # 109 - from_end(104) = 5 -> not < 5 -> new cluster
    f.add([E(4, 100 * S, 195 * S)])              # to: 195 - 200 = -5 -> not > -5 -> new cluster
    cl = f.clusters()
    assert [list(c["keys"]) for c in cl] == [[1, 2], [3], [4]]
    assert (cl[0]["from_start_ns"], cl[0]["from_end_ns"], cl[0]["to_start_ns"], cl[0]["to_end_ns"]) == (100 * S, 104 * S, 200 * S, 203 * S)
    assert cl[0]["changed"] == 1 and cl[1]["changed"] == 0
    assert list(f.all_edges()) == [1, 2, 3, 4]


def test_merge_and_max_cluster_size(oracle):
    f = oracle.Filter(max_dt=5.0, max_cluster_size=6)
    f.add([E(1, 100 * S, 200 * S), E(2, 101 * S, 200 * S)])          # cluster A = {1,2}
    f.add([E(3, 108 * S, 200 * S), E(4, 109 * S, 200 * S)])          # cluster B = {3,4}  (108-101 = 7 >= 5)
    assert [list(c["keys"]) for c in f.clusters()] == [[1, 2], [3, 4]]
    f.add([E(5, 104 * S + S // 2, 200 * S)])                          # matches A and B -> added to A, then 3 + 2 < 6 -> merge
    cl = f.clusters()
    assert [list(c["keys"]) for c in cl] == [[1, 2, 5, 3, 4]]
    assert cl[0]["from_end_ns"] == 109 * S
    f.add([E(6, 105 * S, 200 * S)])                                   # size 5 < 6 -> joins, cluster is now full
    f.add([E(7, 105 * S, 200 * S)])                                   # size 6 not < 6 -> own cluster
    assert [c["size"] for c in f.clusters()] == [6, 1]
    # a merge that would reach the cap is refused (strict <)
    g = oracle.Filter(max_dt=5.0, max_cluster_size=5)
    g.add([E(1, 100 * S, 200 * S), E(2, 101 * S, 200 * S), E(3, 108 * S, 200 * S), E(4, 109 * S, 200 * S)])
    g.add([E(5, 104 * S + S // 2, 200 * S)])                          # 3 + 2 = 5 not < 5
    assert [list(c["keys"]) for c in g.clusters()] == [[1, 2, 5], [3, 4]]


def test_consensus_counter_and_remove(oracle):
    f = oracle.Filter()
    f.add([E(1, 100 * S, 200 * S, valid=1), E(2, 101 * S, 200 * S, valid=0), E(3, 102 * S, 201 * S, valid=1)])
    assert f.clusters()[0]["consensus"] == 2
    f.remove([3])
    c = f.clusters()[0]
    assert c["consensus"] == 1 and list(c["keys"]) == [1, 2]
    f.remove([1, 2])
    assert f.clusters() == [] and len(f.all_edges()) == 0
    f.remove([99])                                                    # unknown id: no-op
    # an edge whose nodes carry two stamps each lands in the same cluster four times: counted four times (:74-76)
    f.add([E(7, [100 * S, 101 * S], [200 * S, 201 * S], valid=1)])
    c = f.clusters()[0]
    assert c["size"] == 1 and c["consensus"] == 4
    f.remove([7])
    assert f.clusters() == []


def test_update_keeps_validity_and_stamps(oracle):
    f = oracle.Filter()
    f.add([E(1, 100 * S, 200 * S, valid=1)])
    f.add([E(1, 500 * S, 900 * S, valid=0)])                          # known id: only the stored edge / poses change
    c = f.clusters()[0]
    assert c["size"] == 1 and c["from_start_ns"] == 100 * S and list(c["valid"]) == [1] and c["consensus"] == 1
    assert len(f.clusters()) == 1


def _line_edges(n, dt_s=0.5, bad=()):
    """n edges forming one cluster; from-chain lands exactly on the to position except for `bad` (1 m off)"""
    out = []
    for k in range(n):
        pf = np.eye(3, 4); pf[:, 3] = [0.3 * k, 0.1 * k * k * 0.01, 0.0]
        pt = np.eye(3, 4); pt[:, 3] = [5.0 + 0.3 * k, 1.0 + 0.02 * k, 0.1 * (k % 3)]
        T = np.eye(3, 4); T[:, 3] = pt[:, 3] - pf[:, 3] + ([1.0, -1.0, 0.5] if k in bad else [0.0, 0.0, 0.0])
        out.append(E(10 + k, int((100 + dt_s * k) * S), int((300 + dt_s * k) * S), score=float(50 - k),
                     pose_from=pf.reshape(12), pose_to=pt.reshape(12), transform=T.reshape(12)))
    return out


def test_calc_valid_edges_gates_and_verdict(oracle):
    f = oracle.Filter(min_size=8.0, seed=3)
    f.add(_line_edges(7))
    assert f.calc_valid_edges() == 0                                  # size 7 < min_size
    f = oracle.Filter(min_size=8.0, seed=3)
    f.add(_line_edges(12, dt_s=0.1))
    assert f.calc_valid_edges() == 0                                  # spans 1.1 s < 2 s
    assert f.clusters()[0]["changed"] == 1                            # the gate leaves changed_ set
    f = oracle.Filter(min_size=8.0, seed=3)
    f.add(_line_edges(12, bad=(3, 7)))
    assert f.calc_valid_edges() == 1
    c = f.clusters(with_eval=True)[0]
    assert c["changed"] == 0 and c["evaluations"] == 1
    assert list(c["valid"]) == [0 if k in (3, 7) else 1 for k in range(12)] and c["consensus"] == 10
    assert np.allclose(c["P"][0], c["Q"][0]) and not np.allclose(c["P"][3], c["Q"][3])
    assert f.calc_valid_edges() == 0                                  # unchanged cluster is skipped
    assert list(f.valid_edges()) == [10 + k for k in range(12) if k not in (3, 7)]      # 10 valid: not > 2*5 -> all


def test_valid_edges_thinning(oracle):
    f = oracle.Filter(min_size=8.0, seed=5)
    f.add(_line_edges(16))
    assert f.calc_valid_edges() == 1
    assert f.clusters()[0]["consensus"] == 16
    # 16 valid > 10: best 5 by score (keys 10..14), then floor(3.2 i) for i < 4 -> sorted positions 0, 3, 6, 9, then the last
    assert list(f.valid_edges()) == [10, 11, 12, 13, 14, 16, 19, 25]


def test_failed_ransac_falls_back_to_identity_count(oracle):
    """fewer than 3 consistent pairs: estimateSVD returns T = I, consensus3D(I) still counts near pairs (:275-276)"""
    f = oracle.Filter(min_size=2.0, max_error=0.3, seed=1)
    es = []
    for k in range(4):
        pf = np.eye(3, 4); pf[:, 3] = [k, 0, 0]
        pt = np.eye(3, 4); pt[:, 3] = [k, 0.05 if k < 2 else 3.0 * k * k, 0]     # two pairs agree with the identity, two are far off
        es.append(E(k + 1, (100 + 2 * k) * S, (200 + 2 * k) * S, pose_from=pf.reshape(12), pose_to=pt.reshape(12)))
    f.add(es)
    assert f.calc_valid_edges() == 1
    c = f.clusters(with_eval=True)[0]
    assert c["ransac_consensus"] == 0
    assert np.array_equal(c["T"], np.eye(3, 4).reshape(12)) and list(c["valid"]) == [1, 1, 0, 0]


def test_oracle_equals_python_restatement(oracle):
    scn = synth.make_filter_scenario(160, 420, seed=11)
    cfg = dict(max_dt=5.0, min_size=6.0, max_cluster_size=30, ransac_iterations=60, max_error=0.3, seed=9)
    o = oracle.Filter(**cfg)
    r = FilterRef(**cfg)
    o.set_sensors(scn["sensors"]); r.set_sensors(scn["sensors"])

    def ransac(P, Q, job_id):
        res = oracle.prosac(P.T, Q.T, cfg["max_error"], cfg["ransac_iterations"], 1.0, do_prosac=False, seed=cfg["seed"], job_id=job_id)
        _, s = oracle.consensus3d(P.T, Q.T, res["T"], cfg["max_error"])
        return res["T"], s

    def check(rnd, stage):
        assert_same_state(o.clusters(), r.state(), with_eval=False, tag=(rnd, stage))
        assert np.array_equal(o.all_edges(), r.all_edges())
        if stage == "calc":
            assert np.array_equal(o.valid_edges(), r.valid_edges())
            for co, cr in zip(o.clusters(with_eval=True), r.clusters):
                if co["evaluations"] and hasattr(cr, "lastP") and len(cr.lastP) == len(co["P"]):
                    assert co["P"].tobytes() == cr.lastP.tobytes() and co["Q"].tobytes() == cr.lastQ.tobytes()

    play([o, r], scn, rounds=5, seed=2, check=check, calc=lambda f: f.calc_valid_edges(ransac) if isinstance(f, FilterRef) else f.calc_valid_edges())
    st = o.clusters()
    assert max(c["size"] for c in st) >= 12 and sum(c["evaluations"] for c in st) >= 5      # the scenario exercises the path
    assert len(o.valid_edges()) > 0
