"""Wire / disk formats.  CPU tests: the product's host-side encoders / decoders (C ABI, no device work)
against the independent struct / numpy restatement in oracle/wire.py, byte for byte; known answers for the ROS 1
serialisation rules and the pose <-> quaternion conventions of Conversions; rosbag round trips through a directory laid
out like RosbagStorage's.  GPU tests: Feature records <-> frame arena (uzl_match_add_frames_wire / frame_to_wire)."""
import struct

import numpy as np
import pytest

from oracle import wire as OW
from uzliti_slam_amd import synth, wire as W


def _rot(rng, big=False):
    from scipy.spatial.transform import Rotation
    v = rng.normal(size=3)
    v *= (rng.uniform(2.6, 3.14159) if big else rng.uniform(0, 1.5)) / np.linalg.norm(v)
    T = np.zeros((3, 4)); T[:, :3] = Rotation.from_rotvec(v).as_matrix(); T[:, 3] = rng.normal(size=3) * 3
    return T.reshape(12)


def _edge(rng, k=0):
    A = rng.normal(size=(6, 6))
    return dict(id=f"16123{k}.5-edge", id_from=f"n{k}", id_to=f"n{k + 7}", sensor_from="camera_rgb_optical_frame", sensor_to="",
                type=int(rng.integers(0, 4)), valid=int(k % 2), transform=_rot(rng, big=k % 3 == 0), information=(A @ A.T).reshape(36),
                displacement_from=_rot(rng), displacement_to=_rot(rng), error=float(rng.normal()), age=float(k), matching_score=float(50 + k),
                diff_time_sec=-3 if k % 2 else 12, diff_time_nsec=int(rng.integers(0, 10**9)))


def _frame(rng, n, D=32):
    desc = rng.integers(0, 256, size=(n, D), dtype=np.uint8)
    pos = rng.normal(size=(3, n)) * 2
    pos[2, rng.random(n) < 0.1] = -1.0                  # invalid depth marker (feature_extraction_core.cpp:286-289)
    valid = (pos[2] > 0).astype(np.uint8)
    uv = rng.integers(0, 640, size=(n, 2)).astype(np.int32)
    return desc, pos, valid, uv


def _node(rng, k, frames):
    sensors = []
    for j, (desc, pos, valid, uv) in enumerate(frames):
        sensors.append(dict(raw=None, sensor_type=1, stamp_sec=1400000000 + k, stamp_nsec=123456789 + j, sensor_frame=f"cam{j}_optical",
                            displacement=_rot(rng), descriptor_type=2, n_features=len(desc), desc_len=desc.shape[1] if len(desc) else 0,
                            records=OW.features_pack(desc, pos, valid, uv) if len(desc) else b"", camera_info=None))
    return dict(id=f"1400000{k:03d}.25", stamps_ns=[(1400000000 + k) * 10**9 + 5, (1400000001 + k) * 10**9], pose=_rot(rng), odom_pose=_rot(rng),
                sensors=sensors, edge_ids=[f"e{k}", f"e{k + 1}", ""], fixed=int(k == 0), uncertainty=0.25 * k)


def test_known_answers_of_the_serialisation_rules():
    # identity pose -> position 0, orientation (0,0,0,1); 180 deg about z -> Eigen's largest-diagonal branch gives (0,0,1,0)
    e = _edge(np.random.default_rng(0))
    e.update(id="ab", id_from="", id_to="c", sensor_from="", sensor_to="", type=1, valid=1, error=1.5, age=2.0, matching_score=3.0,
             diff_time_sec=-1, diff_time_nsec=7)
    e["displacement_from"] = np.eye(3, 4).reshape(12)
    e["displacement_to"] = np.array([-1, 0, 0, 1.0, 0, -1, 0, 2.0, 0, 0, 1, 3.0])
    b = W.encode_edge(e)
    assert b[:6] == struct.pack("<I", 2) + b"ab" and b[6] == 1
    o = 7 + 4 + 4 + 1
    assert struct.unpack_from("<7d", b, o) == (0, 0, 0, 0, 0, 0, 1)
    assert struct.unpack_from("<7d", b, o + 56) == (1, 2, 3, 0, 0, 1, 0)
    tail = struct.unpack_from("<3d", b, o + 3 * 56 + 288)
    assert tail == (1.5, 2.0, 3.0)
    assert b[-17:] == struct.pack("<II", 0, 0) + b"\x01" + struct.pack("<ii", -1, 7)
    assert len(b) == 2 + 0 + 1 + 5 * 4 + 1 + 3 * 56 + 288 + 24 + 1 + 8


def test_edge_encode_decode_match_the_oracle_bytewise():
    rng = np.random.default_rng(5)
    for k in range(40):
        e = _edge(rng, k)
        b = W.encode_edge(e)
        assert b == OW.encode_edge(e)
        d, used = W.decode_edge(b + b"trailing")
        o, used_o = OW.decode_edge(b)
        assert used == used_o == len(b)
        for f in ("transform", "information", "displacement_from", "displacement_to"):
            assert np.array_equal(d[f], o[f])
        for f in ("id", "id_from", "id_to", "sensor_from", "sensor_to"):
            assert d[f] == o[f] == e[f].encode()
        for f in ("type", "valid", "error", "age", "matching_score", "diff_time_sec", "diff_time_nsec"):
            assert d[f] == o[f] == e[f]
        # toMsg writes Quaterniond(R) without a sign convention, fromMsg is toRotationMatrix: R survives to rounding
        assert np.abs(d["transform"] - e["transform"]).max() < 1e-14
        for cut in (0, 3, 60, len(b) - 1):
            with pytest.raises(Exception):
                W.decode_edge(b[:cut])


def test_unnormalised_quaternion_is_not_normalised_on_decode():
    """fromVectorQT builds the matrix from the quaternion as it is (isometry3d_mappings.cpp:131-136)."""
    e = _edge(np.random.default_rng(1))
    b = bytearray(OW.encode_edge(e))
    o = 4 + len(e["id"]) + 1 + 4 + len(e["id_from"]) + 4 + len(e["id_to"]) + 2 * 56
    struct.pack_into("<7d", b, o, 1.0, 2.0, 3.0, 0.0, 0.0, 0.0, 2.0)       # q = (0,0,0,2): toRotationMatrix gives the identity
    struct.pack_into("<7d", b, o - 56, 0.0, 0.0, 0.0, 0.5, 0.0, 0.0, 0.5)  # |q|^2 = 0.5: not a rotation, and kept that way
    d, _ = W.decode_edge(bytes(b))
    assert np.array_equal(d["transform"], [1, 0, 0, 1, 0, 1, 0, 2, 0, 0, 1, 3])
    assert np.array_equal(d["displacement_to"], OW.pose_from_wire([0, 0, 0], [0.5, 0, 0, 0.5]))
    assert np.array_equal(d["displacement_to"].reshape(3, 4)[:, :3], [[1, 0, 0], [0, 0.5, -0.5], [0, 0.5, 0.5]])


def test_node_encode_decode_match_the_oracle_bytewise():
    rng = np.random.default_rng(9)
    for k, shapes in enumerate([[(50, 32)], [(7, 32), (0, 32), (33, 64)], []]):
        n = _node(rng, k, [_frame(rng, a, D) for a, D in shapes])
        b = W.encode_node(n)
        assert b == OW.encode_node(n)
        d = W.decode_node(b)
        o, used = OW.decode_node(b)
        assert d.used == used == len(b)
        f = d.fields
        assert f["id"] == o["id"] and f["stamps_ns"] == o["stamps_ns"] == n["stamps_ns"] and f["edge_ids"] == o["edge_ids"]
        assert np.array_equal(f["pose"], o["pose"]) and np.array_equal(f["odom_pose"], o["odom_pose"])
        assert f["fixed"] == o["fixed"] == n["fixed"] and f["uncertainty"] == o["uncertainty"] == n["uncertainty"]
        assert len(f["sensors"]) == len(o["sensors"]) == len(shapes)
        for s, so, (a, D) in zip(f["sensors"], o["sensors"], shapes):
            for key in ("raw", "sensor_type", "stamp_sec", "stamp_nsec", "sensor_frame", "descriptor_type", "n_features", "uniform", "records",
                        "camera_info"):
                assert s[key] == so[key], key
            assert s["desc_len"] == so["desc_len"] == (D if a else 0)
            assert np.array_equal(s["displacement"], so["displacement"])
            assert len(s["records"]) == W.features_size(a, D) == a * (41 + 4 * D)
        # re-encoding from the raw sub-messages reproduces the message (copy-through of sensor types we do not touch)
        n2 = dict(n, sensors=[dict(raw=s["raw"]) for s in f["sensors"]])
        assert W.encode_node(n2) == b
        for cut in (0, 11, len(b) // 2, len(b) - 1):
            with pytest.raises(Exception):
                W.decode_node(b[:cut])


def test_ragged_descriptor_lengths_are_reported():
    rng = np.random.default_rng(2)
    d1, p1, v1, u1 = _frame(rng, 3, 32)
    d2, p2, v2, u2 = _frame(rng, 2, 16)
    n = _node(rng, 0, [])
    rec = OW.features_pack(d1, p1, v1, u1) + OW.features_pack(d2, p2, v2, u2)
    n["sensors"] = [dict(raw=None, sensor_type=1, stamp_sec=1, stamp_nsec=2, sensor_frame="c", displacement=np.eye(3, 4).reshape(12), descriptor_type=2,
                         n_features=5, desc_len=32, records=rec, camera_info=None)]
    s = W.decode_node(OW.encode_node(n)).fields["sensors"][0]
    assert s["n_features"] == 5 and s["desc_len"] == 32 and s["uniform"] == 0 and s["records"] == rec


def test_feature_records_oracle_round_trip_and_byte_cast():
    rng = np.random.default_rng(3)
    desc, pos, valid, uv = _frame(rng, 100, 32)
    rec = OW.features_pack(desc, pos, valid, uv)
    assert len(rec) == 100 * 169
    d, p, v, u = OW.features_unpack(rec, 100, 32)
    assert np.array_equal(d, desc) and np.array_equal(p, pos) and np.array_equal(v, valid) and np.array_equal(u, uv)
    assert OW.float_to_byte(np.array([0.0, 255.0, 255.9, 256.0, 257.5, -1.0, -0.5, 3e9, np.nan, -3e9], np.float32)).tolist() == \
        [0, 255, 255, 0, 1, 255, 0, 0, 0, 0]


def test_bag_single_message_matches_the_oracle_and_reads_back():
    rng = np.random.default_rng(4)
    data = W.encode_edge(_edge(rng))
    args = (b"edge", b"graph_slam_msgs/Edge", b"0123456789abcdef0123456789abcdef", b"string id\nuint8 type\n", 1400000000, 1, data)
    img = W.bag_write_single(*args)
    assert img == OW.bag_write_single(*args)
    # rosbag::Bag::writeFileHeaderRecord: data_len = FILE_HEADER_LENGTH (4096) - header_len, so header + padding = 4096 bytes, the
    # record 4104 and the first chunk record starts at byte 13 + 4104 = 4117; index_pos points at the connection record of the index section
    assert img[:13] == b"#ROSBAG V2.0\n" and img[4117 - 4:4117] == b"    " and img[4117:4117 + 4] != b"    "
    (hl,) = struct.unpack_from("<I", img, 13)
    (dl,) = struct.unpack_from("<I", img, 13 + 4 + hl)
    assert hl + dl == 4096
    (cl,) = struct.unpack_from("<I", img, 4117)
    assert OW._parse_fields(img[4121:4121 + cl])[b"op"] == b"\x05"                      # the chunk record
    hdr = OW._parse_fields(img[17:17 + hl])
    (index_pos,) = struct.unpack("<Q", hdr[b"index_pos"])
    (l2,) = struct.unpack_from("<I", img, index_pos)
    assert OW._parse_fields(img[index_pos + 4:index_pos + 4 + l2])[b"op"] == b"\x07"
    for reader in (W.bag_read, OW.bag_read):
        (m,) = reader(img)
        assert (m["topic"], m["datatype"], m["md5sum"], m["definition"], m["sec"], m["nsec"], m["data"]) == args
    with pytest.raises(Exception):
        W.bag_read(img[:5000 if len(img) > 5000 else len(img) - 3])
    with pytest.raises(Exception):
        W.bag_read(b"#ROSBAG V1.2\n" + img[13:])
    # a compressed chunk is refused, not misread
    bz = img.replace(b"compression=none", b"compression=bz2\x00", 1)
    with pytest.raises(Exception) as ei:
        W.bag_read(bz)
    assert getattr(ei.value, "status", None) == W.UZL_ERR_UNSUPPORTED


def test_bag_with_two_chunks_and_two_topics_reads_in_file_order():
    """A bag as a recorder would write it (several chunks, connections repeated in the index section): built with the
    oracle's record writer, read by the product."""
    t = struct.pack("<II", 5, 6)
    def conn(i, topic):
        return OW._record({b"op": b"\x07", b"conn": struct.pack("<I", i), b"topic": topic},
                          OW._fields({b"type": b"T%d" % i, b"md5sum": b"m%d" % i, b"message_definition": b"", b"topic": topic}))
    def msg(i, d):
        return OW._record({b"op": b"\x02", b"conn": struct.pack("<I", i), b"time": t}, d)
    def chunk(body):
        return OW._record({b"op": b"\x05", b"compression": b"none", b"size": struct.pack("<I", len(body))}, body)
    body1 = conn(0, b"node") + msg(0, b"AAAA") + conn(1, b"edge") + msg(1, b"BB")
    body2 = msg(1, b"CCCCCC") + msg(0, b"")
    head = OW._record({b"op": b"\x03", b"index_pos": struct.pack("<Q", 0), b"conn_count": struct.pack("<I", 2), b"chunk_count": struct.pack("<I", 2)}, b" " * 100)
    img = OW.BAG_MAGIC + head + chunk(body1) + chunk(body2) + conn(0, b"node") + conn(1, b"edge")
    got = W.bag_read(img)
    assert [(m["topic"], m["datatype"], m["data"]) for m in got] == [(b"node", b"T0", b"AAAA"), (b"edge", b"T1", b"BB"), (b"edge", b"T1", b"CCCCCC"), (b"node", b"T0", b"")]
    assert [(m["topic"], m["data"]) for m in OW.bag_read(img)] == [(m["topic"], m["data"]) for m in got]


def _meta(rng, k=0, n0=3, n1=2):
    return dict(stamp_sec=1400000000 + k, stamp_nsec=int(rng.integers(0, 10**9)), frame_id="/map", name=f"graph{k}",
                map_transform=_rot(rng, big=k % 2 == 1),
                sensor_transforms=[(f"/camera{j}_rgb_optical_frame" if j else "", _rot(rng, big=j == 1)) for j in range(n0)],
                sensor_transforms_initial=[(f"/camera{j}_rgb_optical_frame", _rot(rng)) for j in range(n1)],
                odometry_parameters=rng.normal(size=6))


def test_graph_meta_encode_decode_match_the_oracle_bytewise():
    """graph_slam_msgs/GraphMeta (SlamGraph::toMetaData / updateMetaData, slam_graph.cpp:592-633; RosbagStorage::storeMetaData,
    rosbag_storage.cpp:94-107): the product's host-side codec against oracle/wire.py byte for byte, incl. empty arrays and names."""
    rng = np.random.default_rng(21)
    for k, (n0, n1) in enumerate([(3, 2), (0, 0), (1, 0), (0, 4), (7, 7)]):
        m = _meta(rng, k, n0, n1)
        b = W.encode_meta(m)
        assert b == OW.encode_meta(m)
        d, used = W.decode_meta(b + b"trailing")
        o, used_o = OW.decode_meta(b)
        assert used == used_o == len(b)
        assert (d["stamp_sec"], d["stamp_nsec"], d["frame_id"], d["name"]) == (o["stamp_sec"], o["stamp_nsec"], o["frame_id"], o["name"])
        assert np.array_equal(d["map_transform"], o["map_transform"]) and np.array_equal(d["odometry_parameters"], o["odometry_parameters"])
        assert np.array_equal(d["odometry_parameters"], np.asarray(m["odometry_parameters"]))
        for key in ("sensor_transforms", "sensor_transforms_initial"):
            assert [x[0] for x in d[key]] == [x[0] for x in o[key]] == [x[0].encode() for x in m[key]]
            for (_, Td), (_, To), (_, Tm) in zip(d[key], o[key], m[key]):
                assert np.array_equal(Td, To) and np.abs(Td - Tm).max() < 1e-14
        assert W.encode_meta(d) == OW.encode_meta(o)                     # decode -> encode is stable on both sides
        for cut in sorted(set(rng.integers(0, len(b), size=12).tolist() + [0, len(b) - 1])):
            with pytest.raises(Exception) as ei:
                W.decode_meta(b[:cut])
            assert getattr(ei.value, "code", W.UZL_ERR_TRUNCATED) == W.UZL_ERR_TRUNCATED
    # an array count the message cannot hold is a truncation, not an allocation
    m = _meta(rng, 0, 1, 0)
    b = bytearray(OW.encode_meta(m))
    off = 12 + 4 + len(b"/map") + 4 + len(b"graph0") + 56
    assert struct.unpack_from("<I", b, off)[0] == 1
    struct.pack_into("<I", b, off, 0xFFFFFFF0)
    with pytest.raises(Exception):
        W.decode_meta(bytes(b))


def test_graph_meta_known_answer():
    m = dict(stamp_sec=5, stamp_nsec=6, frame_id="/map", name="g", map_transform=np.eye(3, 4).reshape(12),
             sensor_transforms=[("cam", np.array([-1, 0, 0, 1.0, 0, -1, 0, 2.0, 0, 0, 1, 3.0]))], sensor_transforms_initial=[],
             odometry_parameters=np.arange(6.0))
    b = W.encode_meta(m)
    want = (struct.pack("<III", 0, 5, 6) + struct.pack("<I", 4) + b"/map" + struct.pack("<I", 1) + b"g" + struct.pack("<7d", 0, 0, 0, 0, 0, 0, 1) +
            struct.pack("<I", 1) + struct.pack("<I", 3) + b"cam" + struct.pack("<7d", 1, 2, 3, 0, 0, 1, 0) + struct.pack("<I", 0) +
            struct.pack("<6d", 0, 1, 2, 3, 4, 5))
    assert b == want


def test_storage_meta_round_trip(tmp_path):
    """storeMetaData writes <path>/meta/meta with topic "meta"; loadGraph's meta pass reads it back (rosbag_storage.cpp:94-107,187-207)."""
    rng = np.random.default_rng(22)
    st = W.RosbagStorage(str(tmp_path / "graph"), clear_storage=True)
    assert st.load_meta() is None
    m = _meta(rng, 3)
    st.store_meta(_meta(rng, 1), now_ns=5)
    st.store_meta(m, now_ns=10**9)                                       # overwrites the file, as bag.open(Write) does
    assert [p.name for p in (tmp_path / "graph" / "meta").iterdir()] == ["meta"]
    img = (tmp_path / "graph" / "meta" / "meta").read_bytes()
    got = OW.bag_read(img)
    assert len(got) == 1 and got[0]["topic"] == b"meta" and got[0]["datatype"] == b"graph_slam_msgs/GraphMeta" and got[0]["data"] == OW.encode_meta(m)
    d = W.RosbagStorage(str(tmp_path / "graph")).load_meta()
    assert d["name"] == b"graph3" and np.array_equal(d["odometry_parameters"], m["odometry_parameters"])
    assert [x[0] for x in d["sensor_transforms"]] == [x[0].encode() for x in m["sensor_transforms"]]


def test_storage_directory_round_trip_on_the_host(tmp_path):
    """RosbagStorage layout (rosbag_storage.cpp:62-136, 211-235) without a device: nodes keep their Feature records as bytes."""
    rng = np.random.default_rng(6)
    st = W.RosbagStorage(str(tmp_path / "graph"), clear_storage=True)
    nodes = [_node(rng, k, [_frame(rng, 20 + k, 32)]) for k in range(4)]
    edges = [_edge(rng, k) for k in range(6)]
    for n in nodes:
        st.store_node(n, now_ns=10**9 * (k := 1))
    for e in edges:
        st.store_edge(e, now_ns=10**9)
    st.remove_node(nodes[1]["id"]); st.remove_edge(edges[0]["id"]); st.remove_edge("never-there")
    assert sorted(p.name for p in (tmp_path / "graph" / "nodes").iterdir()) == sorted(n["id"] for i, n in enumerate(nodes) if i != 1)
    N, E = W.RosbagStorage(str(tmp_path / "graph")).load_graph()
    assert sorted(N) == sorted(n["id"].encode() for i, n in enumerate(nodes) if i != 1)
    assert sorted(E) == sorted(e["id"].encode() for e in edges[1:])
    for n in nodes[2:]:
        got = N[n["id"].encode()]
        assert got["sensors"][0]["records"] == n["sensors"][0]["records"] and got["stamps_ns"] == n["stamps_ns"]
        assert np.abs(got["pose"] - n["pose"]).max() < 1e-14
    st.clear()
    assert list((tmp_path / "graph" / "nodes").iterdir()) == []


# ------------------------------------------------------------------------------------------------ device
@pytest.mark.gpu
def test_feature_records_unpack_on_the_device_bit_exact(capi):
    rng = np.random.default_rng(11)
    m = capi.Match()
    shapes = [(1000, 32), (7, 32), (0, 32), (333, 64), (1, 4), (2049, 32)]
    frames = [_frame(rng, a, D) for a, D in shapes]
    node = _node(rng, 1, frames)
    d = W.decode_node(OW.encode_node(node))
    ids, uv = W.add_frames_wire(m, d.sensors_c, len(shapes), sensor_frame_keys=list(range(len(shapes))), want_uv=True)
    assert len(set(ids)) == len(shapes) and m.frame_count() == len(shapes)
    row = 0
    for fid, (desc, pos, valid, u), (a, D) in zip(ids, frames, shapes):
        gd, gp, gv = W.get_frame(m, fid)
        od, op, ov, ou = OW.features_unpack(OW.features_pack(desc, pos, valid, u), a, D) if a else (desc, pos, valid, u)
        assert gd.shape == (a, D if a else 32)
        if a:
            assert np.array_equal(gd, od) and np.array_equal(gd, desc)
            assert np.array_equal(gp.view(np.uint64), op.view(np.uint64)) and np.array_equal(gv, ov)
            assert np.array_equal(uv[row:row + a], ou)
        row += a
        # and back: FeatureData::toMsg on the device reproduces the records byte for byte
        assert W.frame_to_wire(m, fid, u if a else None) == node["sensors"][ids.index(fid)]["records"]
        if a:
            z = W.frame_to_wire(m, fid, None)
            assert z == OW.features_pack(desc, pos, valid, None)
    # frames that arrived over the wire and frames added as arrays are the same thing to the estimator
    f2 = m.add_frame(frames[0][0], frames[0][1], frames[0][2])
    a, b = W.get_frame(m, ids[0]), W.get_frame(m, f2)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    m.close()


@pytest.mark.gpu
def test_float_descriptor_values_off_the_byte_grid_and_bad_records(capi):
    rng = np.random.default_rng(12)
    m = capi.Match()
    desc, pos, valid, uv = _frame(rng, 64, 32)
    rec = np.frombuffer(OW.features_pack(desc, pos, valid, uv), OW.feature_dtype(32)).copy()
    weird = np.array([255.9, 256.0, 257.5, -1.0, -0.5, 3e9, np.nan, -3e9, 0.99, 1e-30, 128.5, 65535.0], np.float32)
    rec["descriptor"][:12, 5] = weird
    rec["descriptor"][20:32, 31] = weird
    node = _node(rng, 0, [])
    node["sensors"] = [dict(raw=None, sensor_type=1, stamp_sec=1, stamp_nsec=2, sensor_frame="c", displacement=np.eye(3, 4).reshape(12), descriptor_type=2,
                            n_features=64, desc_len=32, records=rec.tobytes(), camera_info=None)]
    d = W.decode_node(OW.encode_node(node))
    (fid,), _ = W.add_frames_wire(m, d.sensors_c, 1)
    gd, gp, gv = W.get_frame(m, fid)
    od, op, ov, _ = OW.features_unpack(rec.tobytes(), 64, 32)
    assert np.array_equal(gd, od) and np.array_equal(gp.view(np.uint64), op.view(np.uint64)) and np.array_equal(gv, ov)
    # a record whose descriptor count disagrees with the declared length (same stride, corrupted count) is refused
    bad = rec.copy(); bad["count"][40] = 31
    node["sensors"][0]["records"] = bad.tobytes()
    sens = (W.WireSensor * 1)(W._sensor_in(k := W._Keep(), node["sensors"][0]))
    with pytest.raises(capi.UzlError):
        W.add_frames_wire(m, sens, 1)
    assert m.frame_count() == 1
    # float descriptors (SURF / SIFT) are not binary descriptors: refused
    node["sensors"][0].update(records=rec.tobytes(), descriptor_type=5)
    sens = (W.WireSensor * 1)(W._sensor_in(k, node["sensors"][0]))
    with pytest.raises(capi.UzlError) as e:
        W.add_frames_wire(m, sens, 1)
    assert e.value.status == W.UZL_ERR_UNSUPPORTED
    m.close()


@pytest.mark.gpu
def test_stored_graph_loads_into_the_estimator_and_estimates_like_array_frames(capi, tmp_path):
    """nodes written by storeNode, read back by loadGraph with every feature frame unpacked on the device in one launch;
    edge estimation on those frames equals estimation on frames added as arrays."""
    pairs = synth.make_pairs(6, n_kp=400, seed=21)
    st = W.RosbagStorage(str(tmp_path / "g"), clear_storage=True)
    rng = np.random.default_rng(13)
    names = []
    for j, (f, t, _) in enumerate(pairs):
        for side, fr in (("a", f), ("b", t)):
            n = _node(rng, j, [])
            n["id"] = f"{1400000000 + j}.{side}"
            uv = np.zeros((len(fr["desc"]), 2), np.int32)
            n["sensors"] = [dict(raw=None, sensor_type=1, stamp_sec=j, stamp_nsec=0, sensor_frame="cam", displacement=np.eye(3, 4).reshape(12),
                                 descriptor_type=2, n_features=len(fr["desc"]), desc_len=32,
                                 records=OW.features_pack(fr["desc"], fr["pos"], fr["valid"], uv), camera_info=None)]
            st.store_node(n)
            names.append(n["id"].encode())
    m = capi.Match(ransac_threshold=0.1, ransac_iteration=200, ransac_break_percentage=0.6, seed=5)
    N, _ = st.load_graph(match=m, sensor_frame_key=lambda s: 3)
    assert m.frame_count() == 12
    wire_ids = [(N[f"{1400000000 + j}.a".encode()]["frame_ids"][0], N[f"{1400000000 + j}.b".encode()]["frame_ids"][0]) for j in range(6)]
    arr_ids = [(m.add_frame(f["desc"], f["pos"], f["valid"], sensor_frame=3), m.add_frame(t["desc"], t["pos"], t["valid"], sensor_frame=3)) for f, t, _ in pairs]
    ra, _ = m.estimate(wire_ids, job_ids=list(range(6)))
    rb, _ = m.estimate(arr_ids, job_ids=list(range(6)))
    for a, b in zip(ra, rb):
        assert a["ok"] == b["ok"] == 1 and a["consensus"] == b["consensus"] and np.array_equal(a["T"], b["T"]) and np.array_equal(a["information"], b["information"])
    m.close()


def test_parsers_under_address_and_ub_sanitizers(tmp_path):
    """The decoders take bytes from disk: 100k mutated / truncated Edge, Node, GraphMeta and bag images through an ASan + UBSan build of
    the host-side codec (sanitizers run on the CPU build only)."""
    import os
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z = np.load(os.path.join(root, "tests", "golden", "wire_msgs.npz"))
    seeds = []
    for k in ("edge_bytes", "node_bytes", "bag_bytes"):
        p = tmp_path / (k + ".bin"); p.write_bytes(z[k].tobytes()); seeds.append(str(p))
    p = tmp_path / "meta_bytes.bin"; p.write_bytes(OW.encode_meta(_meta(np.random.default_rng(5), 2, 3, 2))); seeds.append(str(p))
    exe = str(tmp_path / "fuzz_wire")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-x", "c++",
                           os.path.join(root, "uzliti_slam_amd", "csrc", "uzl_wire.hip"), os.path.join(root, "tests", "fuzz_wire.cpp"), "-o", exe])
    out = subprocess.run([exe] + seeds, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "fuzz: 100000 inputs" in out.stdout
