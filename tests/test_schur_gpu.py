"""GPU tests of the Schur-reduced solve (uzl_pgo_cfg::schur_reduce, csrc/pgo_schur.hpp): chain interiors eliminated exactly per LM
trial, PCG on the Schur complement over the rest, back-substitution.  Exact linear algebra, so the result must sit within the parity
bar of the CPU checker's direct solve (g2o_optimizer.cpp:137-149) and next to the unreduced solve of the same graph."""
import numpy as np
import pytest

from uzliti_slam_amd import synth

pytestmark = pytest.mark.gpu

TOL_T, TOL_R = 1e-3, 1e-4


def _oracle(oracle, g, its):
    fl = oracle.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    fixed, _ = oracle.set_fixed_nodes(fl["fixed"], fl["ij"])
    return oracle.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=its)


def _solve(capi, g, its, **cfg):
    p = capi.Pgo(**cfg)
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st = p.optimize(its)
    poses = p.store()[0].reshape(-1, 3, 4)
    p.close()
    return st, poses


@pytest.mark.parametrize("n,e,its", [(600, 640, 6), (2000, 2040, 8), (3000, 3300, 20), (6000, 6500, 5), (20000, 21800, 4)])
def test_reduced_solve_vs_oracle_and_full_solve(capi, oracle, n, e, its):
    g = synth.make_pose_graph(n, e, seed=n + 1)
    st, poses = _solve(capi, g, its)
    assert st["status"] == 0 and st["n_eliminated"] > n // 2, st
    P, so = _oracle(oracle, g, its)
    dt, dr = synth.pose_errors(poses, P.reshape(-1, 3, 4))
    assert dt < TOL_T and dr < TOL_R, (dt, dr)
    assert st["iterations_done"] == so["iterations_done"] or st["terminated_early"] or so["terminated_early"]
    assert abs(st["chi2_final"] - so["chi2_final"]) <= 1e-6 * abs(so["chi2_final"]) + 1e-9
    st0, poses0 = _solve(capi, g, its, schur_reduce=-1)
    assert st0["n_eliminated"] == 0
    dt, dr = synth.pose_errors(poses, poses0)
    assert dt < TOL_T and dr < TOL_R, (dt, dr)
    assert st["pcg_iterations"] < 2 * st0["pcg_iterations"] + 50           # (iterations of a system a quarter the size)


def test_everything_eliminated(capi, oracle):
    """Chains hanging off the fixed vertex, none longer than a run: no separator is left and the elimination IS the solve."""
    rng = np.random.default_rng(2)
    arms, L = 5, 20
    n = 1 + arms * L
    gt = np.tile(np.eye(3, 4), (n, 1, 1))
    frm, to = [], []
    for a in range(arms):
        prev = 0
        for k in range(L):
            v = 1 + a * L + k
            step = synth.se3_from_noise(np.array([[0.3, 0.02 * a, 0.0]]), np.array([[0.0, 0.0, 0.1 * (a - 2)]]))[0]
            gt[v] = synth.se3_mul(gt[prev], step)
            frm.append(prev); to.append(v); prev = v
    frm = np.array(frm, np.int32); to = np.array(to, np.int32)
    E = len(frm)
    Z = synth.se3_mul(synth.se3_inv(gt[frm]), gt[to])
    Z = synth.se3_mul(Z, synth.se3_from_noise(rng.normal(0, 0.01, (E, 3)), rng.normal(0, 0.002, (E, 3))))
    init = gt.copy(); init[1:] = synth.se3_mul(gt[1:], synth.se3_from_noise(rng.normal(0, 0.05, (n - 1, 3)), rng.normal(0, 0.02, (n - 1, 3))))
    I12 = np.eye(3, 4).reshape(12)
    edges = {"from": frm, "to": to, "type": np.full(E, synth.EDGE_TYPE_3D_FULL, np.int32), "sensor_from": np.full(E, -1, np.int32),
             "sensor_to": np.full(E, -1, np.int32), "valid": np.ones(E, np.int32), "transform": Z.reshape(E, 12),
             "displacement_from": np.tile(I12, (E, 1)), "displacement_to": np.tile(I12, (E, 1)),
             "information": np.tile((np.eye(6) * 400.0).reshape(36), (E, 1)), "diff_time": np.zeros(E)}
    fixed = np.zeros(n, np.uint8); fixed[0] = 1
    g = dict(nodes_pose=init.reshape(n, 12), nodes_fixed=fixed, edges=edges)
    st, poses = _solve(capi, g, 6)
    assert st["status"] == 0 and st["n_eliminated"] == n - 1 and st["pcg_iterations"] == 0
    P, so = _oracle(oracle, g, 6)
    dt, dr = synth.pose_errors(poses, P.reshape(-1, 3, 4))
    assert dt < 1e-6 and dr < 1e-7, (dt, dr)                                # no iterative solve in between: rounding only


def test_structure_reuse_and_repeatability(capi):
    g = synth.make_pose_graph(4000, 4300, seed=11)
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st1 = p.optimize(5); a = p.store()[0].copy()
    p.reset()
    st2 = p.optimize(5); b = p.store()[0].copy()
    assert st2["structure_reused"] == 1 and st1["n_eliminated"] == st2["n_eliminated"] > 2000
    assert np.array_equal(a, b)                                             # deterministic: no atomics anywhere in the reduction
    p.close()
