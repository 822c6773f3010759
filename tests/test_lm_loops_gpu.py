"""The device-resident Levenberg-Marquardt loop (uzl_pgo_cfg::lm_loop = 0: decisions on the device, captured passes, one host look per
trial) against the host-driven loop (lm_loop = 1) on the same graphs: same kernel bodies, same order of operations, same scalar
arithmetic (csrc/pgo_lm.hpp) - so poses, chi2, lambda and every counter must be IDENTICAL, on every kernel path a graph class takes
(small / composite operator, large / level-2 operator, Schur-reduced, additive operator, rejected trials, early termination).
The host-driven loop is the one all earlier rounds' oracle comparisons ran on; tests/test_pgo_gpu.py etc. now run the device loop
against the oracle directly.  Reference: OptimizationAlgorithmLevenberg::solve behind graph_optimization/src/g2o_optimizer.cpp:148."""
import numpy as np
import pytest

from uzliti_slam_amd import synth

pytestmark = pytest.mark.gpu

KEYS = ("iterations_done", "lm_trials", "pcg_iterations", "terminated_early", "precond_builds", "n_eliminated",
        "chi2_initial", "chi2_final", "lambda_final")


def _solve(capi, g, its, loop, set_graph=None, **cfg):
    p = capi.Pgo(lm_loop=loop, **cfg)
    if set_graph is not None:
        p.set_graph(*set_graph)
    else:
        p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st = p.optimize(its)
    poses, err, used = p.store()
    st2 = None
    if its > 2:                                   # a second optimize on the same handle continues from the result (no reset)
        st2 = p.optimize(2)
    poses2, _, _ = p.store()
    p.close()
    return st, poses, err, st2, poses2


def _same(capi, g, its=20, **cfg):
    a = _solve(capi, g, its, 0, **cfg)
    b = _solve(capi, g, its, 1, **cfg)
    assert a[0]["lm_passes"] > 0 and b[0]["lm_passes"] == 0, (a[0], b[0])
    for k in KEYS:
        assert a[0][k] == b[0][k], (k, a[0], b[0])
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2], equal_nan=True)
    if a[3] is not None:
        for k in KEYS:
            assert a[3][k] == b[3][k], (k, a[3], b[3])
    assert np.array_equal(a[4], b[4])
    return a[0]


@pytest.mark.parametrize("n,e,its", [(100, 300, 20), (1000, 5000, 20), (300, 1200, 7), (64, 70, 5), (2000, 9000, 10)])
def test_small_graph_class(capi, n, e, its):
    """<= 4096 free vertices: one row per wave, dense level-1 operator, rebuilds ahead on the second stream."""
    st = _same(capi, synth.make_pose_graph(n, e, seed=n + e), its)
    assert st["iterations_done"] >= 1


def test_c4_large_graph_class(capi):
    """10k / 50k (BASELINE config 4's graph on one GPU): four rows per wave, level-2 operator, synchronous rebuilds."""
    st = _same(capi, synth.make_pose_graph(10000, 50000), 8)
    assert st["n_eliminated"] == 0


@pytest.mark.parametrize("n,e", [(1500, 1530), (5000, 5400), (20000, 21800)])
def test_schur_reduced_class(capi, n, e):
    """Chain-like graphs (the shape of an online run, graph_slam_node.cpp:578-663): chain interiors eliminated per lambda."""
    st = _same(capi, synth.make_pose_graph(n, e, seed=3), 10)
    assert st["n_eliminated"] > 0


def test_without_schur_and_with_block_hierarchy_variants(capi):
    g = synth.make_pose_graph(1500, 1530, seed=3)
    _same(capi, g, 8, schur_reduce=-1)
    _same(capi, synth.make_pose_graph(6000, 30000, seed=5), 6)          # AGG = 4 with the six operator rows in registers
    _same(capi, synth.make_pose_graph(14000, 60000, seed=6), 4)         # ... streamed, alpha prepared by ml_alpha_kernel


def test_rejected_trials_and_termination(capi):
    """Scrambled initial guess: LM rejects steps (lambda grows, the same linearisation is solved again, the trial set-up is retaken when
    lambda has grown 32x); and a zero-residual graph, where every late trial is rejected by rounding noise and LM terminates early."""
    g = synth.make_pose_graph(240, 900, seed=77, outlier_frac=0.35)
    rng = np.random.default_rng(5)
    P0 = g["nodes_pose"].reshape(-1, 3, 4).copy()
    P0[1:] = synth.se3_mul(P0[1:], synth.se3_from_noise(rng.normal(0, 1.5, (239, 3)), rng.normal(0, 0.8, (239, 3))))
    g2 = dict(g); g2["nodes_pose"] = P0.reshape(-1, 12)
    st = _same(capi, g2, 8, pcg_tol=1e-12)
    assert st["lm_trials"] > st["iterations_done"] + 3, "the case is meant to contain rejected trials"
    # measurements taken from the start poses themselves: chi2 = 0 from the first iteration on
    g3 = synth.make_pose_graph(300, 1000, seed=21, outlier_frac=0.0)
    P = g3["nodes_pose"].reshape(-1, 3, 4)
    e = dict(g3["edges"])
    e["transform"] = synth.se3_mul(synth.se3_inv(P[np.asarray(e["from"])]), P[np.asarray(e["to"])]).reshape(-1, 12)
    g3["edges"] = e
    st = _same(capi, g3, 12)
    assert st["terminated_early"] == 1 or st["lm_trials"] > st["iterations_done"]


def test_relative_stop_test_switch(capi, oracle):
    """cfg.pcg_stop = 1 restores the plain relative residual test: both loops, and tighter than the default against the oracle."""
    g = synth.make_pose_graph(500, 2200, seed=12)
    st = _same(capi, g, 10, pcg_stop=1, pcg_tol=1e-6)
    st0 = _same(capi, g, 10)
    assert st["pcg_iterations"] > st0["pcg_iterations"]


@pytest.mark.parametrize("n,e", [(1000, 5000), (1500, 1530)])
def test_pass_history_changes_passes_not_results(capi, n, e):
    """uzl_pgo_cfg::pass_history: whether a repeated optimize sizes its passes from the previous one's per-trial counts (0) or from the
    running optimize alone (1) changes how many passes the host enqueues - never a bit of the result."""
    g = synth.make_pose_graph(n, e, seed=77)
    out = []
    for hist in (0, 1):
        p = capi.Pgo(pass_history=hist)
        p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
        runs = []
        for _ in range(3):
            p.reset(); st = p.optimize(12)
            runs.append((st, p.store()[0].copy()))
        p.close()
        out.append(runs)
    for k in range(3):
        (s0, P0), (s1, P1) = out[0][k], out[1][k]
        assert np.array_equal(P0, P1)
        for f in ("iterations_done", "lm_trials", "pcg_iterations", "chi2_final", "lambda_final"):
            assert s0[f] == s1[f], (k, f)
    assert out[0][0][0]["lm_passes"] > 0 and out[1][2][0]["lm_passes"] >= out[0][2][0]["lm_passes"]
    for k in (1, 2):                                          # every repetition of a handle solves the same problem to the same bits
        assert np.array_equal(out[1][0][1], out[1][k][1])
