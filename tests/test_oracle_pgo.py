"""CPU tests (no GPU): pin the C oracle of the pose-graph half against known answers from the in-tree
g2o excerpt (graph_slam_common/thirdparty/src/isometry3d_mappings.cpp) and against the independent
NumPy/SciPy LM (tests/np_reference.py)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import np_reference as NP
from uzliti_slam_amd import synth


def _rand_pose(rng, rs=1.0, ts=1.0):
    return synth.se3(synth.quat_to_R(synth.quat_from_rotvec(rng.normal(size=3) * rs)), rng.normal(size=3) * ts)


def test_quaternion_maps_kat(oracle):
    # identity, 90 deg about z, 180 deg about x (trace <= 0 branch), w<0 flip
    assert np.allclose(oracle.quat_from_R(np.eye(3)), [1, 0, 0, 0])
    Rz = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    assert np.allclose(oracle.quat_from_R(Rz), [np.sqrt(.5), 0, 0, np.sqrt(.5)])
    Rx180 = np.diag([1.0, -1, -1])
    assert np.allclose(np.abs(oracle.quat_from_R(Rx180)), [0, 1, 0, 0])
    rng = np.random.default_rng(0)
    for _ in range(300):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        R = oracle.R_from_quat(q)
        assert np.allclose(R, Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix(), atol=1e-14)
        q2 = oracle.quat_from_R(R)
        assert np.allclose(q2, q, atol=1e-12) or np.allclose(q2, -q, atol=1e-12)


def test_vector_mqt_roundtrip_and_edge_cases(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        T = _rand_pose(rng, 2.0)
        v = oracle.to_vector_mqt(T)
        assert np.allclose(v[:3], T[:, 3])
        q = Rotation.from_matrix(T[:, :3]).as_quat()
        q = q if q[3] >= 0 else -q                       # normalize(): w >= 0 (:38-44)
        assert np.allclose(v[3:], q[:3], atol=1e-12)
        assert np.allclose(oracle.from_vector_mqt(v), T, atol=1e-12)
    # ||v||^2 > 1 -> identity rotation (:84-91)
    T = oracle.from_vector_mqt(np.array([1, 2, 3, 0.8, 0.8, 0.0]))
    assert np.array_equal(T[:, :3], np.eye(3)) and np.array_equal(T[:, 3], [1, 2, 3])
    # exactly 180 degrees: w = 0
    T = oracle.from_vector_mqt(np.array([0, 0, 0, 1.0, 0, 0]))
    assert np.allclose(T[:, :3], np.diag([1.0, -1, -1]))
    assert np.allclose(NP.to_vector_mqt(T[None])[0][:3], 0)


def test_euler_roundtrip(oracle):
    rng = np.random.default_rng(2)
    for _ in range(200):
        rpy = rng.uniform(-1.2, 1.2, 3)
        R = oracle.from_euler(rpy)
        assert np.allclose(R, Rotation.from_euler("ZYX", rpy[::-1]).as_matrix(), atol=1e-12)
        assert np.allclose(oracle.to_euler(R), rpy, atol=1e-10)


def test_edge_error_and_jacobians(oracle):
    rng = np.random.default_rng(3)
    worst = 0.0
    for k in range(200):
        Xi, Xj, Z = _rand_pose(rng), _rand_pose(rng), _rand_pose(rng, 2.0)
        if k % 10 == 0:
            Z = synth.se3_mul(synth.se3_inv(Xi), Xj)     # zero error
        e = oracle.edge_error(Xi, Xj, Z)
        e_np = NP.edge_errors(np.stack([Xi, Xj]).reshape(2, 12), np.array([[0, 1]]), Z.reshape(1, 12))[0]
        assert np.allclose(e, e_np, atol=1e-12)
        Ji, Jj = oracle.edge_jacobians(Xi, Xj, Z)
        Jin, Jjn = NP.numeric_jacobians(np.stack([Xi, Xj]).reshape(2, 12), np.array([[0, 1]]), Z.reshape(1, 12))
        worst = max(worst, np.abs(Ji - Jin[0]).max(), np.abs(Jj - Jjn[0]).max())
    assert worst < 5e-8


def test_huber_kat(oracle):
    """rho = (e2, 1, 0) if e2 <= delta^2 else (2 delta sqrt(e2) - delta^2, delta/sqrt(e2), .) [EXT]."""
    for e2, want in ((0.0, (0, 1)), (0.5, (0.5, 1)), (1.0, (1.0, 1)), (1.0 + 1e-9, (2 * np.sqrt(1 + 1e-9) - 1, 1 / np.sqrt(1 + 1e-9))),
                     (100.0, (19.0, 0.1))):
        r = oracle.huber(e2, 1.0)
        assert np.allclose(r[:2], want, atol=1e-15)
        n0, n1 = NP.huber(np.array([e2]))
        assert np.allclose([n0[0], n1[0]], want)


def test_flatten_rules(oracle):
    g = synth.make_pose_graph(40, 120, seed=7)
    e = g["edges"]
    e["valid"][50:55] = 0; e["from"][60] = -1; e["to"][61] = 999
    g["nodes_fixed"][5] = 1; g["nodes_fixed"][6] = 1
    fl = oracle.flatten_graph(g["nodes_pose"], g["nodes_fixed"], e)
    src = set(fl["src_edge"].tolist())
    assert 0 not in src                                   # odom 0->1: from fixed, to free (:203-206)
    assert 6 not in src and 5 in src and 4 in src         # 6->7 dropped, 5->6 (both fixed) kept, 4->5 kept
    assert not (src & {50, 51, 52, 53, 54, 60, 61})
    # odometry first, then feature edges; robust only on feature edges (:292-294)
    types = e["type"][fl["src_edge"]]
    assert np.all(np.diff((types != synth.EDGE_TYPE_ODOM).astype(int)) >= 0)
    assert np.array_equal(fl["robust"], (types != synth.EDGE_TYPE_ODOM).astype(np.uint8))
    # any type other than TYPE_2D_WHEEL_ODOMETRY (Edge.msg values 1-4, 101-103, 105) goes down the feature-edge branch (:80-103)
    e2 = {k: v.copy() for k, v in e.items()}
    e2["type"][e2["type"] != synth.EDGE_TYPE_ODOM] = synth.EDGE_TYPE_2D_LASER
    fl2 = oracle.flatten_graph(g["nodes_pose"], g["nodes_fixed"], e2)
    assert np.array_equal(fl2["src_edge"], fl["src_edge"]) and np.array_equal(fl2["robust"], fl["robust"])
    # identity displacements/sensors: measurement == transform
    assert np.allclose(fl["meas"], e["transform"][fl["src_edge"]])
    # xy-only: z, roll, pitch zeroed
    flx = oracle.flatten_graph(g["nodes_pose"], g["nodes_fixed"], e, optimize_xy_only=True)
    P = flx["poses"].reshape(-1, 3, 4)
    assert np.allclose(P[:, 2, 3], 0) and np.allclose(P[:, 2, :3], [0, 0, 1]) and np.allclose(P[:, :2, 2], 0)
    yaw = np.arctan2(P[:, 1, 0], P[:, 0, 0])
    yaw0 = Rotation.from_matrix(g["nodes_pose"].reshape(-1, 3, 4)[:, :, :3]).as_euler("ZYX")[:, 0]
    assert np.allclose(np.angle(np.exp(1j * (yaw - yaw0))), 0, atol=1e-9)


def test_set_fixed_nodes(oracle):
    fixed = np.zeros(10, np.uint8)
    ij = np.array([[0, 1], [1, 2], [4, 3], [5, 6], [6, 7], [7, 5]], np.int32)   # comps {0,1,2} {3,4} {5,6,7} {8} {9}
    f, c = oracle.set_fixed_nodes(fixed, ij)
    assert c == 5 and np.array_equal(np.nonzero(f)[0], [0, 3, 5, 8, 9])
    fixed[6] = 1
    f, c = oracle.set_fixed_nodes(fixed, ij)
    assert c == 4 and np.array_equal(np.nonzero(f)[0], [0, 3, 6, 8, 9])


def test_normal_equations_vs_numpy(oracle):
    g = synth.make_pose_graph(30, 80, seed=5)
    fl = oracle.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    fixed, _ = oracle.set_fixed_nodes(fl["fixed"], fl["ij"])
    H, b = oracle.build_dense(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"])
    Hn, bn, chi = NP.build_system(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"])
    scale = np.abs(H).max()
    assert np.abs(H - Hn.toarray()).max() < 1e-6 * scale          # numeric vs analytic Jacobians
    assert np.abs(b - bn).max() < 1e-6 * np.abs(b).max()
    assert abs(chi - oracle.chi2(fl["poses"], fl["ij"], fl["meas"], fl["info"], fl["robust"])) < 1e-9 * chi
    assert np.allclose(H, H.T)


@pytest.mark.parametrize("n,e,seed", [(2, 1, 1), (3, 3, 2), (100, 300, 12345), (250, 900, 3), (1000, 5000, 12345)])
def test_lm_vs_scipy_direct(oracle, n, e, seed):
    """tiny graphs + BASELINE configs 1 and 2: oracle LM (block sparse Cholesky) == NumPy/SciPy LM (spsolve) - the checker the GPU path
    is held to is itself cross-checked, at a BASELINE size, by an independent implementation (parity is otherwise unpinned)."""
    g = synth.make_pose_graph(n, e, seed=seed)
    if n <= 3:
        g["nodes_fixed"][:] = 0       # with node 0 fixed the rule at g2o_optimizer.cpp:203-206 drops edge 0->1
    fl = oracle.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    fixed, _ = oracle.set_fixed_nodes(fl["fixed"], fl["ij"])
    Po, so = oracle.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=20)
    Pn, sn = NP.pgo_lm(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=20)
    dt, dr = synth.pose_errors(Po.reshape(-1, 3, 4), Pn.reshape(-1, 3, 4))
    assert dt < 1e-6 and dr < 1e-7, (dt, dr)
    assert abs(so["chi2_initial"] - sn["chi2_initial"]) <= 1e-10 * max(1.0, sn["chi2_initial"])
    assert abs(so["chi2_final"] - sn["chi2_final"]) <= 1e-8 * max(1.0, sn["chi2_final"])
    assert so["chi2_final"] <= so["chi2_initial"]
    assert np.allclose(Po[fixed == 1], fl["poses"][fixed == 1])


def test_two_node_closed_form(oracle):
    """one edge, node 0 fixed: the optimum puts X1 = X0 * Z exactly."""
    rng = np.random.default_rng(9)
    X0 = _rand_pose(rng); Z = _rand_pose(rng, 0.5); X1 = synth.se3_mul(synth.se3_mul(X0, Z), _rand_pose(rng, 0.05, 0.05))
    poses = np.stack([X0, X1]).reshape(2, 12)
    P, st = oracle.pgo_optimize(poses, [1, 0], [[0, 1]], Z.reshape(1, 12), np.eye(6).reshape(1, 36) * 50, [0], iterations=20)
    assert np.allclose(P[1].reshape(3, 4), synth.se3_mul(X0, Z), atol=1e-9) and st["chi2_final"] < 1e-18


def test_odom_convert_known_answers(oracle):
    """g2o OdomConvert round trip [EXT] (g2o_optimizer.cpp:212-214): identity on exact circular arcs, projection otherwise."""
    for R, th, dt in ((2.0, 0.3, 0.5), (-1.5, 0.2, 1.0), (5.0, -0.7, 2.0), (0.4, 1.2, 0.1)):
        x, y = R * np.sin(th), R * (1 - np.cos(th))                  # motion along a circle with its centre at (0, R)
        out = oracle.odom_convert(x, y, th, dt)
        assert np.allclose(out, [x, y, th], atol=1e-12), (R, th, out)
    # straight motion (|theta| <= 1e-7): the lateral component is folded into the forward distance
    assert np.allclose(oracle.odom_convert(0.3, 0.04, 0.0, 0.5), [np.hypot(0.3, 0.04), 0.0, 0.0], atol=1e-15)
    # lateral slip on a turning motion: projected onto the arc with the same heading change, R = x cot(theta) + y
    x, y, th = 0.3, 0.1, 0.2
    R = x / np.tan(th) + y
    assert np.allclose(oracle.odom_convert(x, y, th, 1.0), [R * np.sin(th), R * (1 - np.cos(th)), th], atol=1e-12)
    # no elapsed time: wheel velocities are zero, so is the motion
    assert np.array_equal(oracle.odom_convert(0.3, 0.1, 0.2, 0.0), [0.0, 0.0, 0.0])
    assert np.array_equal(oracle.odom_convert(0.3, 0.1, 0.0, 0.0), [0.0, 0.0, 0.0])
