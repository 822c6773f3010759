#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference ships no tests or fixtures for this path, and none of its libraries exist in this image,
so the vectors are produced by the CPU oracle (oracle/), after the oracle itself has been cross-checked against the
independent NumPy/SciPy implementation in tests/np_reference.py (asserted below before anything is written).
A fixture is data only: inputs + expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(HERE))

import oracle as O                      # noqa: E402
import np_reference as NP               # noqa: E402
from uzliti_slam_amd import synth       # noqa: E402


def match_fixture(name="match_3pairs.npz", n_kp=160, desc_bytes=32, iterations=150, seed_pairs=99):
    cfg = dict(ransac_threshold=0.1, ransac_iteration=iterations, break_percentage=0.6, seed=4242)
    pairs = synth.make_pairs(3, n_kp=n_kp, desc_bytes=desc_bytes, seed=seed_pairs)
    out = dict(n_pairs=3, **{k: np.array(v) for k, v in cfg.items()})
    for j, (f, t, T) in enumerate(pairs):
        r = O.estimate_edge([f], [t], do_prosac=True, job_id=10 + j, **cfg)
        i0, d0, i1, d1 = O.knn2(t["desc"], f["desc"])
        n0 = NP.knn2(t["desc"], f["desc"])
        assert all(np.array_equal(a, b) for a, b in zip((i0, d0, i1, d1), n0)), "oracle knn2 != numpy"
        q, tr, d, nr = NP.filter_sort(i0, d0, i1, d1, f["valid"], t["valid"])
        assert np.array_equal(q, r["corr_query"]) and np.array_equal(tr, r["corr_train"]) and nr == r["n_matches"]
        for side, fr in (("from", f), ("to", t)):
            out[f"p{j}_{side}_desc"] = fr["desc"]; out[f"p{j}_{side}_pos"] = fr["pos"]; out[f"p{j}_{side}_valid"] = fr["valid"]
        out[f"p{j}_knn"] = np.stack([i0, d0, i1, d1])
        for k in ("corr_query", "corr_train", "corr_dist", "mask", "T", "information"):
            out[f"p{j}_{k}"] = r[k]
        out[f"p{j}_scalars"] = np.array([r["ok"], r["consensus"], r["n_matches"], r["n_corr"], r["iterations_run"], r["best_iteration"]], np.int64)
        out[f"p{j}_mse"] = np.array(r["mse"])
    np.savez_compressed(os.path.join(HERE, name), **out)


def match_deployed_fixture():
    """the operating point the reference deploys: BRISK-512 (64-byte descriptors), 300 keypoints, 100 iterations, early exit at 60 %
    (feature_extraction_service_node.cpp:63-66, iti_slam_launch/yaml/slam.yaml:34-38, cfg/FeatureLinkEstimation.cfg:12)"""
    match_fixture("match_deployed_3pairs.npz", n_kp=300, desc_bytes=64, iterations=100, seed_pairs=4243)


def ransac_fixture():
    rng = np.random.default_rng(17)
    out = {}
    ms = (3, 4, 12, 60)
    for b, m in enumerate(ms):
        P = rng.normal(size=(3, m)) * 2
        R = synth.quat_to_R(synth.quat_from_rotvec(rng.normal(size=3) * 0.4)); t = rng.normal(size=3)
        Q = R @ P + t[:, None] + rng.normal(0, 0.02, (3, m))
        if m >= 12:
            Q[:, ::3] += rng.normal(0, 1.5, Q[:, ::3].shape)
        r = O.prosac(P, Q, 0.3, 200, 0.6, do_prosac=False, seed=7, job_id=b)      # TransformationFilter's call shape
        out[f"b{b}_P"] = P; out[f"b{b}_Q"] = Q; out[f"b{b}_T"] = r["T"]; out[f"b{b}_mask"] = r["mask"]
        out[f"b{b}_scalars"] = np.array([r["consensus"], r["iterations_run"], r["best_iteration"]], np.int64)
        out[f"b{b}_mse"] = np.array(r["mse"])
    out["n"] = np.array(len(ms))
    np.savez_compressed(os.path.join(HERE, "ransac_points.npz"), **out)


def pgo_fixture():
    g = synth.make_pose_graph(60, 180, seed=31)
    fl = O.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    fixed, n_gauge = O.set_fixed_nodes(fl["fixed"], fl["ij"])
    P, st = O.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=20)
    Pn, sn = NP.pgo_lm(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=20)
    dt, dr = synth.pose_errors(P.reshape(-1, 3, 4), Pn.reshape(-1, 3, 4))
    assert dt < 1e-7 and dr < 1e-8, "oracle LM != NumPy/SciPy LM"
    e = g["edges"]
    np.savez_compressed(os.path.join(HERE, "pgo_60n_180e.npz"),
                        nodes_pose=g["nodes_pose"], nodes_fixed=g["nodes_fixed"],
                        **{"e_" + k: v for k, v in e.items()},
                        flat_poses=fl["poses"], flat_ij=fl["ij"], flat_meas=fl["meas"], flat_info=fl["info"],
                        flat_robust=fl["robust"], flat_src=fl["src_edge"], fixed_eff=fixed,
                        poses_out=P, chi2=np.array([st["chi2_initial"], st["chi2_final"]]),
                        edge_err=O.edge_error_norms(P, fl["ij"], fl["meas"]))
    # known answers of the in-tree g2o excerpt (isometry3d_mappings.cpp)
    rng = np.random.default_rng(5)
    T = np.stack([synth.se3(synth.quat_to_R(synth.quat_from_rotvec(rng.normal(size=3) * s)), rng.normal(size=3))
                  for s in (0.1, 1.0, 2.5, 3.1)])
    np.savez_compressed(os.path.join(HERE, "isometry_kat.npz"), T=T,
                        mqt=np.stack([O.to_vector_mqt(x) for x in T]),
                        euler=np.stack([O.to_euler(x[:, :3]) for x in T]))


def filter_fixture():
    """TransformationFilter over a growing graph: inputs per round + expected clusters / verdicts (oracle, after it
    agreed with the pure-Python restatement np_reference.FilterRef on every round)."""
    from filter_common import drifted, replay_filter_fixture
    scn = synth.make_filter_scenario(140, 380, seed=57)
    cfg = dict(max_dt=5.0, min_size=6.0, max_cluster_size=40, ransac_iterations=80, max_error=0.3, seed=13)
    E = scn["edges"]; n = len(E); rounds = 5
    rng = np.random.default_rng(8)
    cut = np.linspace(0, n, rounds + 1).astype(int)
    present = np.zeros((rounds, n), np.uint8); cur = np.zeros(n, bool)
    poses = np.zeros((rounds, 140, 12))
    for r in range(rounds):
        cur[cut[r]:cut[r + 1]] = True
        present[r] = cur                           # edges (re-)added this round; those dropped next round are removed after the add
        drop = cur & (rng.random(n) < 0.04)
        cur = cur & ~drop
        poses[r] = drifted(scn, r, rng).reshape(-1, 12)
    removed = np.zeros((rounds, n), np.uint8)
    for r in range(rounds):
        nxt = present[r + 1] if r + 1 < rounds else present[r]
        removed[r] = present[r] & ~nxt[:n] if r + 1 < rounds else 0
    stamps = np.zeros((140, 2), np.int64); nst = np.zeros(140, np.int32)
    for i, st in enumerate(scn["stamps"]):
        stamps[i, :len(st)] = st; nst[i] = len(st)
    z = dict(cfg_names=np.array(list(cfg)), cfg_values=np.array([float(v) for v in cfg.values()]),
             key=np.array([e["key"] for e in E], np.uint64), score=np.array([e["matching_score"] for e in E]),
             valid=np.array([e["valid"] for e in E], np.int32), sensor_from=np.array([e["sensor_from"] for e in E], np.int32),
             sensor_to=np.array([e["sensor_to"] for e in E], np.int32), node_from=np.array([e["node_from"] for e in E], np.int32),
             node_to=np.array([e["node_to"] for e in E], np.int32), transform=np.stack([e["transform"] for e in E]),
             displacement_from=np.stack([e["displacement_from"] for e in E]), displacement_to=np.stack([e["displacement_to"] for e in E]),
             stamps=stamps, n_stamps=nst, sensors=scn["sensors"], poses=poses, present=present, removed=removed)
    o = O.Filter(**cfg); ref = NP.FilterRef(**cfg)

    def ransac(P, Q, job_id):
        res = O.prosac(P.T, Q.T, cfg["max_error"], cfg["ransac_iterations"], 1.0, do_prosac=False, seed=cfg["seed"], job_id=job_id)
        return res["T"], O.consensus3d(P.T, Q.T, res["T"], cfg["max_error"])[1]
    exp_o = list(replay_filter_fixture(z, o))
    exp_r = list(replay_filter_fixture(z, ref, calc=lambda f: f.calc_valid_edges(ransac), state=lambda f: f.state()))
    for a, b in zip(exp_o, exp_r):
        assert a["evaluated"] == b["evaluated"] and np.array_equal(a["valid_keys"], b["valid_keys"]), "oracle filter != python restatement"
        assert [c["size"] for c in a["clusters"]] == [c["size"] for c in b["clusters"]]
        assert all(np.array_equal(x["keys"], y["keys"]) and np.array_equal(x["valid"], y["valid"]) for x, y in zip(a["clusters"], b["clusters"]))
    assert sum(a["evaluated"] for a in exp_o) >= 4 and len(exp_o[-1]["valid_keys"]) > 5
    for r, a in enumerate(exp_o):
        z[f"r{r}_valid_keys"] = a["valid_keys"]; z[f"r{r}_evaluated"] = np.array(a["evaluated"])
        z[f"r{r}_cluster_info"] = np.array([[c["uid"], c["size"], c["consensus"], c["changed"], c["evaluations"], c["from_start_ns"],
                                              c["from_end_ns"], c["to_start_ns"], c["to_end_ns"]] for c in a["clusters"]], np.int64).reshape(-1, 9)
        z[f"r{r}_cluster_keys"] = np.concatenate([c["keys"] for c in a["clusters"]] + [np.zeros(0, np.uint64)])
        z[f"r{r}_cluster_valid"] = np.concatenate([c["valid"] for c in a["clusters"]] + [np.zeros(0, np.uint8)])
    np.savez_compressed(os.path.join(HERE, "filter_140n_380e.npz"), **z)


def wire_fixture():
    """graph_slam_msgs Edge / Node messages and a single-message bag, serialised by oracle/wire.py.  Before anything is written
    the hand-derivable parts are asserted: field offsets of the Edge message, record stride 41 + 4 D, quaternion conventions."""
    import struct
    from oracle import wire as OW
    rng = np.random.default_rng(2024)
    from scipy.spatial.transform import Rotation

    def pose(big=False):
        v = rng.normal(size=3); v *= (3.0 if big else rng.uniform(0, 1.2)) / np.linalg.norm(v)
        T = np.zeros((3, 4)); T[:, :3] = Rotation.from_rotvec(v).as_matrix(); T[:, 3] = rng.normal(size=3)
        return T.reshape(12)
    A = rng.normal(size=(6, 6))
    edge = dict(id="1400000007.5-1400000001.25", id_from="1400000001.25", id_to="1400000007.5", sensor_from="camera_rgb_optical_frame",
                sensor_to="camera_rgb_optical_frame", type=1, valid=1, transform=pose(big=True), information=(A @ A.T).reshape(36),
                displacement_from=pose(), displacement_to=pose(), error=0.125, age=3.0, matching_score=87.0, diff_time_sec=6, diff_time_nsec=250000000)
    eb = OW.encode_edge(edge)
    o = 4 + len(edge["id"]) + 1 + 4 + len(edge["id_from"]) + 4 + len(edge["id_to"])
    assert struct.unpack_from("<3d", eb, o + 112) == tuple(edge["transform"][[3, 7, 11]])          # transformation.pose.position
    q = np.array(struct.unpack_from("<4d", eb, o + 136))                                             # orientation x y z w
    assert np.allclose(np.abs(Rotation.from_quat(q).as_matrix() - edge["transform"].reshape(3, 4)[:, :3]).max(), 0, atol=1e-14)
    assert np.array_equal(np.frombuffer(eb, "<f8", 36, o + 168), edge["information"])
    n, D = 24, 32
    desc = rng.integers(0, 256, size=(n, D), dtype=np.uint8); pos = rng.normal(size=(3, n)); valid = (rng.random(n) > 0.2).astype(np.uint8)
    uv = rng.integers(0, 640, size=(n, 2)).astype(np.int32)
    rec = OW.features_pack(desc, pos, valid, uv)
    assert len(rec) == n * (41 + 4 * D) and struct.unpack_from("<iiBfI", rec, 0) == (uv[0, 0], uv[0, 1], valid[0], -1.0, D)
    assert struct.unpack_from("<f", rec, 17 + 4 * 5)[0] == float(desc[0, 5]) and struct.unpack_from("<3d", rec, 17 + 4 * D) == tuple(pos[:, 0])
    node = dict(id="1400000001.25", stamps_ns=[1400000001250000000, 1400000002000000000], pose=pose(), odom_pose=pose(), edge_ids=[edge["id"], "odo-1"],
                fixed=0, uncertainty=0.5,
                sensors=[dict(raw=None, sensor_type=1, stamp_sec=1400000001, stamp_nsec=250000000, sensor_frame="camera_rgb_optical_frame", displacement=pose(),
                              descriptor_type=2, n_features=n, desc_len=D, records=rec, camera_info=None)])
    nb = OW.encode_node(node)
    back, used = OW.decode_node(nb)
    assert used == len(nb) and back["sensors"][0]["records"] == rec and back["stamps_ns"] == node["stamps_ns"]
    bag = OW.bag_write_single(b"edge", b"graph_slam_msgs/Edge", b"0" * 32, b"", 1400000010, 1, eb)
    (m,) = OW.bag_read(bag)
    assert m["data"] == eb and bag[:13] == b"#ROSBAG V2.0\n"
    z = dict(edge_bytes=np.frombuffer(eb, np.uint8), node_bytes=np.frombuffer(nb, np.uint8), bag_bytes=np.frombuffer(bag, np.uint8),
             desc=desc, pos=pos, valid=valid, uv=uv, node_pose=node["pose"], node_odom_pose=node["odom_pose"], sensor_displacement=node["sensors"][0]["displacement"])
    for k in ("transform", "information", "displacement_from", "displacement_to"):
        z["edge_" + k] = np.asarray(edge[k])
    np.savez_compressed(os.path.join(HERE, "wire_msgs.npz"), **z)


if __name__ == "__main__":
    match_fixture(); match_deployed_fixture(); ransac_fixture(); pgo_fixture(); filter_fixture(); wire_fixture()
    print("golden fixtures written to", HERE)
