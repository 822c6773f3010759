"""GPU parity of the edge filter (uzl_filter_*, TransformationFilter of transformation_filter.cpp:43-350) against the
CPU oracle: identical cluster bookkeeping, bit-identical point pairs / transforms / votes, identical verdicts."""
import numpy as np
import pytest

from uzliti_slam_amd import synth
from filter_common import assert_same_state, play
from test_oracle_filter import E, S, _line_edges

pytestmark = pytest.mark.gpu


def both(capi, oracle, **cfg):
    return capi.Filter(**cfg), oracle.Filter(**cfg)


@pytest.mark.parametrize("n_nodes,n_loop,cfg", [
    (160, 420, dict(min_size=6.0, max_cluster_size=30, ransac_iterations=60, seed=9)),
    (400, 1400, dict(min_size=8.0, max_cluster_size=100, ransac_iterations=200, seed=1)),       # the reference's settings
    (300, 900, dict(min_size=3.0, max_cluster_size=12, ransac_iterations=33, max_dt=2.5, max_edges=3, seed=77)),
])
def test_filter_rounds_equal_oracle(capi, oracle, n_nodes, n_loop, cfg):
    scn = synth.make_filter_scenario(n_nodes, n_loop, seed=n_nodes + 1)
    g, o = both(capi, oracle, **cfg)
    g.set_sensors(scn["sensors"]); o.set_sensors(scn["sensors"])
    evaluated = []

    def check(rnd, stage):
        assert_same_state(g.clusters(with_eval=(stage == "calc")), o.clusters(with_eval=(stage == "calc")), tag=(rnd, stage))
        assert np.array_equal(g.all_edges(), o.all_edges())
        if stage == "calc":
            assert np.array_equal(g.valid_edges(), o.valid_edges())

    def calc(f):
        evaluated.append(f.calc_valid_edges())

    play([g, o], scn, rounds=6, seed=3, check=check, calc=calc)
    assert evaluated[0::2] == evaluated[1::2] and sum(evaluated) >= 6
    assert len(g.valid_edges()) > 0
    g.close(); o.close()


def test_known_answers_on_gpu(capi, oracle):
    g, o = both(capi, oracle, min_size=8.0, seed=5)
    for f in (g, o):
        f.add(_line_edges(16))
        assert f.calc_valid_edges() == 1
    assert list(g.valid_edges()) == [10, 11, 12, 13, 14, 16, 19, 25]
    assert_same_state(g.clusters(with_eval=True), o.clusters(with_eval=True))
    # failed estimate: identity fallback count
    g, o = both(capi, oracle, min_size=2.0, max_error=0.3, seed=1)
    es = []
    for k in range(4):
        pf = np.eye(3, 4); pf[:, 3] = [k, 0, 0]
        pt = np.eye(3, 4); pt[:, 3] = [k, 0.05 if k < 2 else 3.0 * k * k, 0]
        es.append(E(k + 1, (100 + 2 * k) * S, (200 + 2 * k) * S, pose_from=pf.reshape(12), pose_to=pt.reshape(12)))
    for f in (g, o):
        f.add(es)
        assert f.calc_valid_edges() == 1
    c = g.clusters(with_eval=True)[0]
    assert c["ransac_consensus"] == 0 and list(c["valid"]) == [1, 1, 0, 0]
    assert_same_state(g.clusters(with_eval=True), o.clusters(with_eval=True))


def test_empty_and_bad_arguments(capi):
    f = capi.Filter()
    assert f.calc_valid_edges() == 0 and len(f.valid_edges()) == 0 and len(f.all_edges()) == 0 and f.clusters() == []
    f.remove([1, 2, 3])
    f.add([E(5, [], [100 * S])])                       # a node without stamps: the edge is never clustered (:148-149)
    assert len(f.all_edges()) == 0
    with pytest.raises(capi.UzlError):
        capi.Filter(ransac_iterations=0)
    with pytest.raises(capi.UzlError):
        capi.Filter(max_cluster_size=0)
