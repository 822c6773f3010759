"""Golden fixtures (tests/golden/*.npz, generator tests/golden/make_golden.py).
CPU part: the oracle reproduces them.  GPU part: the HIP path (through the C ABI) reproduces them —
bit-exact for integer / vote work and the float pose recipe, within 1e-3 m / 1e-4 rad for the LM result."""
import os

import numpy as np
import pytest

from uzliti_slam_amd import synth

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _frames(z, j):
    f = dict(desc=z[f"p{j}_from_desc"], pos=z[f"p{j}_from_pos"], valid=z[f"p{j}_from_valid"], feature_type=2, sensor_frame=0)
    t = dict(desc=z[f"p{j}_to_desc"], pos=z[f"p{j}_to_pos"], valid=z[f"p{j}_to_valid"], feature_type=2, sensor_frame=0)
    return f, t


@pytest.mark.parametrize("fixture", ["match_3pairs.npz", "match_deployed_3pairs.npz"])
def test_oracle_reproduces_match_golden(oracle, fixture):
    z = np.load(os.path.join(G, fixture))
    for j in range(int(z["n_pairs"])):
        f, t = _frames(z, j)
        got = oracle.knn2(t["desc"], f["desc"])
        assert np.array_equal(np.stack(got), z[f"p{j}_knn"])
        r = oracle.estimate_edge([f], [t], ransac_threshold=float(z["ransac_threshold"]), ransac_iteration=int(z["ransac_iteration"]),
                                 break_percentage=float(z["break_percentage"]), seed=int(z["seed"]), job_id=10 + j)
        for k in ("corr_query", "corr_train", "corr_dist", "mask", "T", "information"):
            assert np.array_equal(r[k], z[f"p{j}_{k}"]), k
        assert [r["ok"], r["consensus"], r["n_matches"], r["n_corr"], r["iterations_run"], r["best_iteration"]] == z[f"p{j}_scalars"].tolist()
        assert r["mse"] == float(z[f"p{j}_mse"])


def test_oracle_reproduces_pgo_golden(oracle):
    z = np.load(os.path.join(G, "pgo_60n_180e.npz"))
    edges = {k[2:]: z[k] for k in z.files if k.startswith("e_")}
    fl = oracle.flatten_graph(z["nodes_pose"], z["nodes_fixed"], edges)
    assert np.array_equal(fl["ij"], z["flat_ij"]) and np.array_equal(fl["src_edge"], z["flat_src"])
    assert np.allclose(fl["meas"], z["flat_meas"], atol=1e-14)
    fixed, _ = oracle.set_fixed_nodes(fl["fixed"], fl["ij"])
    assert np.array_equal(fixed, z["fixed_eff"])
    P, st = oracle.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=20)
    assert np.allclose(P, z["poses_out"], atol=1e-9)
    assert np.allclose([st["chi2_initial"], st["chi2_final"]], z["chi2"], rtol=1e-10)
    k = np.load(os.path.join(G, "isometry_kat.npz"))
    for T, v, e in zip(k["T"], k["mqt"], k["euler"]):
        assert np.allclose(oracle.to_vector_mqt(T), v, atol=1e-15) and np.allclose(oracle.to_euler(T[:, :3]), e, atol=1e-15)
        assert np.allclose(oracle.from_vector_mqt(v), T, atol=1e-12)


def test_oracle_reproduces_ransac_golden(oracle):
    z = np.load(os.path.join(G, "ransac_points.npz"))
    for b in range(int(z["n"])):
        r = oracle.prosac(z[f"b{b}_P"], z[f"b{b}_Q"], 0.3, 200, 0.6, do_prosac=False, seed=7, job_id=b)
        assert np.array_equal(r["mask"], z[f"b{b}_mask"]) and np.array_equal(r["T"], z[f"b{b}_T"])
        assert [r["consensus"], r["iterations_run"], r["best_iteration"]] == z[f"b{b}_scalars"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", ["match_3pairs.npz", "match_deployed_3pairs.npz"])
def test_hip_reproduces_match_golden(capi, fixture):
    z = np.load(os.path.join(G, fixture))
    m = capi.Match(ransac_threshold=float(z["ransac_threshold"]), ransac_iteration=int(z["ransac_iteration"]),
                   ransac_break_percentage=float(z["break_percentage"]), seed=int(z["seed"]))
    n = int(z["n_pairs"])
    ids = []
    for j in range(n):
        f, t = _frames(z, j)
        ids.append((m.add_frame(f["desc"], f["pos"], f["valid"]), m.add_frame(t["desc"], t["pos"], t["valid"])))
    for j in range(n):
        got = m.knn2(ids[j][0], ids[j][1], z[f"p{j}_to_desc"].shape[0])
        assert np.array_equal(np.stack(got), z[f"p{j}_knn"])
    res, diag = m.estimate(ids, job_ids=[10 + j for j in range(n)], max_corr=int(z["p0_to_desc"].shape[0]))
    for j in range(n):
        sc = z[f"p{j}_scalars"].tolist()
        r = res[j]
        assert [r["ok"], r["consensus"], r["n_matches"], r["n_corr"], r["iterations_run"], r["best_iteration"]] == sc
        k = sc[3]
        assert np.array_equal(diag["corr_query"][j, :k], z[f"p{j}_corr_query"])
        assert np.array_equal(diag["corr_train"][j, :k], z[f"p{j}_corr_train"])
        assert np.array_equal(diag["corr_dist"][j, :k], z[f"p{j}_corr_dist"])
        assert np.array_equal(diag["mask"][j, :k], z[f"p{j}_mask"])
        assert np.array_equal(r["T"].reshape(3, 4), z[f"p{j}_T"]) and r["mse"] == float(z[f"p{j}_mse"])
        assert np.array_equal(r["information"].reshape(6, 6), z[f"p{j}_information"])
    zr = np.load(os.path.join(G, "ransac_points.npz"))
    m.set_config(seed=7)
    probs = [(zr[f"b{b}_P"], zr[f"b{b}_Q"]) for b in range(int(zr["n"]))]
    out = m.ransac_points(probs, 0.3, 200, 0.6, do_prosac=False, job_ids=list(range(len(probs))))
    for b, r in enumerate(out):
        assert np.array_equal(r["mask"], zr[f"b{b}_mask"]) and np.array_equal(r["T"], zr[f"b{b}_T"])
        assert r["consensus"] == int(zr[f"b{b}_scalars"][0]) and r["mse"] == float(zr[f"b{b}_mse"])
    m.close()


@pytest.mark.gpu
def test_hip_reproduces_pgo_golden(capi):
    z = np.load(os.path.join(G, "pgo_60n_180e.npz"))
    edges = {k[2:]: z[k] for k in z.files if k.startswith("e_")}
    p = capi.Pgo()
    p.add_graph(z["nodes_pose"], z["nodes_fixed"], edges)
    st = p.optimize(20)
    poses, err, used = p.store()
    assert np.array_equal(p.get_fixed(), z["fixed_eff"])
    assert np.array_equal(np.nonzero(used)[0], np.sort(z["flat_src"]))
    dt, dr = synth.pose_errors(poses.reshape(-1, 3, 4), z["poses_out"].reshape(-1, 3, 4))
    assert dt < 1e-3 and dr < 1e-4, (dt, dr)
    assert np.allclose([st["chi2_initial"], st["chi2_final"]], z["chi2"], rtol=1e-7)
    assert np.allclose(err[z["flat_src"]], z["edge_err"], atol=2e-4)
    # flat entry on the same fixture
    p.set_graph(z["flat_poses"], z["fixed_eff"], z["flat_ij"], z["flat_meas"], z["flat_info"], z["flat_robust"])
    p.optimize(20)
    poses2, _, _ = p.store()
    dt, dr = synth.pose_errors(poses2.reshape(-1, 3, 4), z["poses_out"].reshape(-1, 3, 4))
    assert dt < 1e-3 and dr < 1e-4
    p.close()


def test_oracle_reproduces_filter_golden(oracle):
    from filter_common import check_against_filter_fixture, replay_filter_fixture
    z = np.load(os.path.join(G, "filter_140n_380e.npz"))
    cfg = dict(zip(z["cfg_names"].tolist(), z["cfg_values"].tolist()))
    for k in ("max_cluster_size", "ransac_iterations", "seed"):
        cfg[k] = int(cfg[k])
    check_against_filter_fixture(z, list(replay_filter_fixture(z, oracle.Filter(**cfg))))


@pytest.mark.gpu
def test_hip_reproduces_filter_golden(capi):
    from filter_common import check_against_filter_fixture, replay_filter_fixture
    z = np.load(os.path.join(G, "filter_140n_380e.npz"))
    cfg = dict(zip(z["cfg_names"].tolist(), z["cfg_values"].tolist()))
    for k in ("max_cluster_size", "ransac_iterations", "seed"):
        cfg[k] = int(cfg[k])
    check_against_filter_fixture(z, list(replay_filter_fixture(z, capi.Filter(**cfg))))


# ------------------------------------------------------------------------------------------------ wire / disk formats
_WIRE_EDGE = dict(id="1400000007.5-1400000001.25", id_from="1400000001.25", id_to="1400000007.5", sensor_from="camera_rgb_optical_frame",
                  sensor_to="camera_rgb_optical_frame", type=1, valid=1, error=0.125, age=3.0, matching_score=87.0, diff_time_sec=6,
                  diff_time_nsec=250000000)


def _wire_inputs(z):
    edge = dict(_WIRE_EDGE, **{k: z["edge_" + k] for k in ("transform", "information", "displacement_from", "displacement_to")})
    sensor = dict(raw=None, sensor_type=1, stamp_sec=1400000001, stamp_nsec=250000000, sensor_frame="camera_rgb_optical_frame",
                  displacement=z["sensor_displacement"], descriptor_type=2, n_features=len(z["desc"]), desc_len=z["desc"].shape[1], camera_info=None)
    node = dict(id="1400000001.25", stamps_ns=[1400000001250000000, 1400000002000000000], pose=z["node_pose"], odom_pose=z["node_odom_pose"],
                edge_ids=[_WIRE_EDGE["id"], "odo-1"], fixed=0, uncertainty=0.5, sensors=[sensor])
    return edge, node


def test_wire_golden_oracle_and_host_codecs():
    from oracle import wire as OW
    from uzliti_slam_amd import wire as W
    z = np.load(os.path.join(G, "wire_msgs.npz"))
    edge, node = _wire_inputs(z)
    rec = OW.features_pack(z["desc"], z["pos"], z["valid"], z["uv"])
    node["sensors"][0]["records"] = rec
    eb, nb, bag = z["edge_bytes"].tobytes(), z["node_bytes"].tobytes(), z["bag_bytes"].tobytes()
    assert OW.encode_edge(edge) == eb and W.encode_edge(edge) == eb
    assert OW.encode_node(node) == nb and W.encode_node(node) == nb
    args = (b"edge", b"graph_slam_msgs/Edge", b"0" * 32, b"", 1400000010, 1, eb)
    assert OW.bag_write_single(*args) == bag and W.bag_write_single(*args) == bag
    (m,) = W.bag_read(bag)
    d, _ = W.decode_edge(m["data"])
    assert np.array_equal(d["information"], z["edge_information"]) and np.abs(d["transform"] - z["edge_transform"]).max() < 1e-14
    s = W.decode_node(nb).fields["sensors"][0]
    assert s["records"] == rec and s["n_features"] == 24 and s["desc_len"] == 32 and s["uniform"] == 1


@pytest.mark.gpu
def test_wire_golden_on_the_device(capi):
    from uzliti_slam_amd import wire as W
    z = np.load(os.path.join(G, "wire_msgs.npz"))
    d = W.decode_node(z["node_bytes"].tobytes())
    m = capi.Match()
    (fid,), uv = W.add_frames_wire(m, d.sensors_c, 1, want_uv=True)
    desc, pos, valid = W.get_frame(m, fid)
    assert np.array_equal(desc, z["desc"]) and np.array_equal(pos.view(np.uint64), z["pos"].view(np.uint64))
    assert np.array_equal(valid, z["valid"]) and np.array_equal(uv, z["uv"])
    assert W.frame_to_wire(m, fid, z["uv"]) == d.fields["sensors"][0]["records"]
    m.close()
