"""uzl_pgo_append_graph: the resident graph grown in place must solve like uzl_pgo_add_graph of the grown arrays whose old nodes carry
the poses uzl_pgo_store returned (what addGraphImpl reads back from the SlamGraph, g2o_optimizer.cpp:55-104 after :106-135)."""
import numpy as np
import pytest

from uzliti_slam_amd import synth

pytestmark = pytest.mark.gpu


def _sub(e, idx):
    return {k: np.asarray(v)[idx] for k, v in e.items()}


@pytest.mark.parametrize("numbering", [1, 2])
@pytest.mark.parametrize("n,e,n1", [(600, 2400, 400), (3000, 3300, 2000), (6000, 6600, 5800)])
def test_append_equals_full_rebuild(capi, n, e, n1, numbering):
    # (reduced_numbering fixed: left to the handle it depends on the handle's own history, which the two handles below do not share)
    g = synth.make_pose_graph(n, e, seed=n + 1)
    ed = g["edges"]
    first = np.nonzero((ed["from"] < n1) & (ed["to"] < n1))[0]
    rest = np.nonzero(~((ed["from"] < n1) & (ed["to"] < n1)))[0]
    assert len(first) > 0 and len(rest) > 0
    p = capi.Pgo(reduced_numbering=numbering)
    p.add_graph(g["nodes_pose"][:n1], g["nodes_fixed"][:n1], _sub(ed, first))
    st0 = p.optimize(4)
    P1 = p.store()[0]
    # the filter changed its mind about a few old feature edges
    feat = np.nonzero(np.asarray(ed["type"])[first] != synth.EDGE_TYPE_ODOM)[0]
    flip = feat[:: max(1, len(feat) // 7)][:7]
    valid_old = np.asarray(ed["valid"])[first].copy()
    valid_old[flip] = 1 - valid_old[flip]
    p.append_graph(g["nodes_pose"][n1:], g["nodes_fixed"][n1:], _sub(ed, rest), flip.astype(np.int32), valid_old[flip].astype(np.uint8))
    sa = p.optimize(6)
    Pa = p.store()[0]
    # the same graph rebuilt from scratch
    full = _sub(ed, np.concatenate([first, rest]))
    full["valid"] = np.concatenate([valid_old, np.asarray(ed["valid"])[rest]])
    poses = np.concatenate([P1.reshape(-1, 12), np.asarray(g["nodes_pose"], np.float64).reshape(-1, 12)[n1:]])
    q = capi.Pgo(reduced_numbering=numbering)
    q.add_graph(poses, g["nodes_fixed"], full)
    sb = q.optimize(6)
    Pb = q.store()[0]
    for k in ("n_vertices", "n_edges", "n_gauge_fixed", "n_eliminated", "iterations_done", "lm_trials"):
        assert sa[k] == sb[k], (k, sa[k], sb[k])
    assert abs(sa["chi2_initial"] - sb["chi2_initial"]) <= 1e-9 * sb["chi2_initial"]      # (matrix -> quaternion -> matrix of the old poses)
    assert abs(sa["chi2_final"] - sb["chi2_final"]) <= 1e-6 * sb["chi2_final"]
    dt, dr = synth.pose_errors(Pa.reshape(-1, 3, 4), Pb.reshape(-1, 3, 4))
    assert dt < 1e-5 and dr < 1e-6, (dt, dr)                 # (two runs of the same LM iterations whose start poses differ in the last bit: the PCG's accuracy, pcg_tol = 1e-5 m)
    # reset() goes back to the state the append left, and a second append keeps working (buffers grown once more)
    p.reset(); sc = p.optimize(6)
    assert np.array_equal(p.store()[0], Pa) and sc["chi2_final"] == sa["chi2_final"]
    p.close(); q.close()
    assert st0["status"] == 0 and sa["status"] == 0


def test_append_needs_a_graph_and_valid_indices(capi):
    g = synth.make_pose_graph(50, 120, seed=3)
    p = capi.Pgo()
    with pytest.raises(capi.UzlError):
        p.append_graph(g["nodes_pose"][:1], g["nodes_fixed"][:1], _sub(g["edges"], np.arange(0)))
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    with pytest.raises(capi.UzlError):
        p.append_graph(g["nodes_pose"][:0], g["nodes_fixed"][:0], _sub(g["edges"], np.arange(0)), np.array([120], np.int32), np.array([1], np.uint8))
    # nothing new, nothing flipped: the same graph, the structure is kept
    p.optimize(3)
    p.append_graph(g["nodes_pose"][:0], g["nodes_fixed"][:0], _sub(g["edges"], np.arange(0)))
    st = p.optimize(3)
    assert st["structure_reused"] == 1
    p.close()
