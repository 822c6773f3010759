"""CPU ORACLE for the wire / disk formats (test infrastructure, NOT product code; PARITY UNPINNED: the reference ships no
recorded messages or bags, and ROS is not installed here, so these functions restate the published ROS 1 serialisation
rules and the rosbag 2.0 record layout [EXT] around the reference's own conversion code).

An independent struct / numpy restatement of
  Conversions::toMsg / fromMsg               graph_slam_common/src/conversions.cpp:43-70, 217-322
  SensorData / FeatureData::toMsg / fromMsg  graph_slam_common/src/sensor_data.cpp:40-167
  RosbagStorage::storeNode / storeEdge / loadGraph   graph_slam_common/src/rosbag_storage.cpp:62-209
for the message layouts of graph_slam_msgs/msg/{Edge,Node,SensorData,SensorDataArray,Features,Feature,GraphMeta,SensorTransform}.msg.
ROS 1 serialisation: little-endian scalars, string = u32 length + bytes, T[] = u32 count + elements, T[N] = elements,
time / duration = two 32-bit words, bool = one byte.
"""
import struct

import numpy as np


# ------------------------------------------------------------------------------------------------ poses
def quat_from_R(T):
    """Eigen::Quaterniond(Matrix3d) [EXT Eigen 3.2], used un-normalised by Conversions::toMsg (conversions.cpp:57-70).
    T: 12 doubles row-major [R|t] -> (x, y, z, w)."""
    T = np.asarray(T, np.float64).reshape(3, 4)
    m = T[:, :3]
    t = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)          # x y z w
    if t > 0.0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def pose_from_wire(p, q):
    """g2o::internal::fromVectorQT (thirdparty/src/isometry3d_mappings.cpp:131-136): Quaterniond(w,x,y,z).toRotationMatrix()
    [EXT Eigen], no normalisation (Conversions::fromMsg, conversions.cpp:229-240)."""
    x, y, z, w = (np.float64(v) for v in q)
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([1 - (tyy + tzz), txy - twz, txz + twy, p[0],
                     txy + twz, 1 - (txx + tzz), tyz - twx, p[1],
                     txz - twy, tyz + twx, 1 - (txx + tyy), p[2]], np.float64)


def _pose_bytes(T):
    T = np.asarray(T, np.float64).reshape(12)
    q = quat_from_R(T)
    return struct.pack("<7d", T[3], T[7], T[11], q[0], q[1], q[2], q[3])


class _R:
    def __init__(self, b, o=0):
        self.b = memoryview(b)
        self.o = o

    def get(self, fmt):
        n = struct.calcsize(fmt)
        if self.o + n > len(self.b):
            raise ValueError("truncated")
        v = struct.unpack_from(fmt, self.b, self.o)
        self.o += n
        return v if len(v) > 1 else v[0]

    def bytes(self, n):
        if self.o + n > len(self.b):
            raise ValueError("truncated")
        v = bytes(self.b[self.o:self.o + n])
        self.o += n
        return v

    def str(self):
        return self.bytes(self.get("<I"))

    def pose(self):
        v = self.get("<7d")
        return pose_from_wire(v[:3], v[3:])

    def header(self):
        seq, sec, nsec = self.get("<III")
        return dict(seq=seq, sec=sec, nsec=nsec, frame_id=self.str())


def _str(s):
    s = s if isinstance(s, (bytes, bytearray)) else str(s).encode()
    return struct.pack("<I", len(s)) + bytes(s)


def _header(sec=0, nsec=0, frame=b""):
    return struct.pack("<III", 0, sec, nsec) + _str(frame)


# ------------------------------------------------------------------------------------------------ Edge
def encode_edge(e):
    """Conversions::toMsg(SlamEdge) (conversions.cpp:255-274) serialised in Edge.msg field order."""
    out = [_str(e["id"]), struct.pack("<B", int(e["type"]) & 0xFF), _str(e["id_from"]), _str(e["id_to"]),
           _pose_bytes(e["displacement_from"]), _pose_bytes(e["displacement_to"]), _pose_bytes(e["transform"]),
           np.asarray(e["information"], "<f8").reshape(36).tobytes(),
           struct.pack("<3d", e["error"], e["age"], e["matching_score"]), _str(e["sensor_from"]), _str(e["sensor_to"]),
           struct.pack("<B", 1 if e["valid"] else 0), struct.pack("<ii", int(e["diff_time_sec"]), int(e["diff_time_nsec"]))]
    return b"".join(out)


def decode_edge(b):
    """Conversions::fromMsg(Edge) (conversions.cpp:242-253)."""
    r = _R(b)
    e = dict(id=r.str(), type=r.get("<B"), id_from=r.str(), id_to=r.str())
    e["displacement_from"] = r.pose(); e["displacement_to"] = r.pose(); e["transform"] = r.pose()
    e["information"] = np.array(r.get("<36d"))
    e["error"], e["age"], e["matching_score"] = r.get("<3d")
    e["sensor_from"] = r.str(); e["sensor_to"] = r.str()
    e["valid"] = int(r.get("<B") != 0)
    e["diff_time_sec"], e["diff_time_nsec"] = r.get("<ii")
    return e, r.o


# ------------------------------------------------------------------------------------------------ GraphMeta
def encode_meta(m):
    """SlamGraph::toMetaData (graph_slam_common/src/slam_graph.cpp:592-619) serialised in GraphMeta.msg field order, as
    RosbagStorage::storeMetaData writes it (rosbag_storage.cpp:94-107).  m: stamp_sec, stamp_nsec, frame_id, name,
    map_transform (12), sensor_transforms / sensor_transforms_initial = [(name, 12 doubles)], odometry_parameters (6)."""
    out = [_header(int(m["stamp_sec"]), int(m["stamp_nsec"]), m["frame_id"]), _str(m["name"]), _pose_bytes(m["map_transform"])]
    for key in ("sensor_transforms", "sensor_transforms_initial"):
        out.append(struct.pack("<I", len(m[key])))
        for name, T in m[key]:
            out += [_str(name), _pose_bytes(T)]
    out.append(np.asarray(m["odometry_parameters"], "<f8").reshape(6).tobytes())      # conversions.cpp:338-343
    return b"".join(out)


def decode_meta(b):
    """SlamGraph::updateMetaData (slam_graph.cpp:621-633) / loadGraph (rosbag_storage.cpp:187-207)."""
    r = _R(b)
    h = r.header()
    m = dict(stamp_sec=h["sec"], stamp_nsec=h["nsec"], frame_id=h["frame_id"], name=r.str())
    m["map_transform"] = r.pose()
    for key in ("sensor_transforms", "sensor_transforms_initial"):
        m[key] = []
        for _ in range(r.get("<I")):
            name = r.str()
            m[key].append((name, r.pose()))
    m["odometry_parameters"] = np.array(r.get("<6d"))
    return m, r.o


# ------------------------------------------------------------------------------------------------ Feature records
def feature_dtype(D):
    """One graph_slam_msgs/Feature on the wire (Feature.msg): 41 + 4 D bytes, nothing aligned."""
    return np.dtype([("u", "<i4"), ("v", "<i4"), ("is_3d", "u1"), ("keypoint_strength", "<f4"), ("count", "<u4"),
                     ("descriptor", "<f4", (D,)), ("keypoint_position", "<f8", (3,))], align=False)


def features_pack(desc, pos, valid, uv=None):
    """FeatureData::toMsg (sensor_data.cpp:86-118): descriptor bytes become floats, keypoint_strength = -1.
    desc (n, D) u8; pos (3, n) f64; valid (n); uv (n, 2) i32 or None."""
    desc = np.asarray(desc, np.uint8)
    n, D = desc.shape
    rec = np.zeros(n, feature_dtype(D))
    if uv is not None:
        rec["u"] = np.asarray(uv, np.int32)[:, 0]; rec["v"] = np.asarray(uv, np.int32)[:, 1]
    rec["is_3d"] = np.asarray(valid).astype(bool)
    rec["keypoint_strength"] = -1.0
    rec["count"] = D
    rec["descriptor"] = desc.astype(np.float32)
    rec["keypoint_position"] = np.asarray(pos, np.float64).T
    return rec.tobytes()


def float_to_byte(f):
    """`(unsigned char) val` (sensor_data.cpp:137) as x86 evaluates it: truncation to int32, low eight bits; values
    outside the int32 range and NaN convert to 0x80000000 -> 0."""
    f = np.asarray(f, np.float32)
    ok = np.abs(f) < np.float32(2147483648.0)
    t = np.where(ok, np.trunc(np.where(ok, f, 0)), 0).astype(np.int64)
    return (t & 0xFF).astype(np.uint8)


def features_unpack(b, n, D):
    """FeatureData::fromMsg (sensor_data.cpp:123-167) -> desc (n, D) u8, pos (3, n) f64, valid (n) u8, uv (n, 2) i32."""
    rec = np.frombuffer(b, feature_dtype(D), count=n)
    if n and not np.all(rec["count"] == D):
        raise ValueError("ragged descriptor lengths")
    return (float_to_byte(rec["descriptor"]).reshape(n, D), np.ascontiguousarray(rec["keypoint_position"].T),
            (rec["is_3d"] != 0).astype(np.uint8), np.stack([rec["u"], rec["v"]], 1).astype(np.int32))


# ------------------------------------------------------------------------------------------------ SensorData / Node
_CAMERA_INFO_DEFAULT = bytes(16 + 8 + 4 + 4 + (9 + 9 + 12) * 8 + 8 + 17)      # default-constructed sensor_msgs/CameraInfo
_IMAGE_DEFAULT = bytes(16 + 8 + 4 + 1 + 4 + 4)                                # sensor_msgs/Image
_LASER_SCAN_DEFAULT = bytes(16 + 7 * 4 + 4 + 4)                               # sensor_msgs/LaserScan


def encode_sensor(s):
    """SensorData::toMsg (sensor_data.cpp:40-49) + FeatureData::toMsg (:78-121); SensorData.msg field order."""
    if s.get("raw") is not None:
        return bytes(s["raw"])
    frame = s["sensor_frame"]
    rec = s.get("records", b"")
    out = [_header(s["stamp_sec"], s["stamp_nsec"], frame), struct.pack("<i", s["sensor_type"]), _pose_bytes(s["displacement"]),
           _str(frame),
           _header(s["stamp_sec"], s["stamp_nsec"], frame), struct.pack("<iI", s["descriptor_type"], s["n_features"]), bytes(rec),
           bytes(s["camera_info"]) if s.get("camera_info") is not None else _CAMERA_INFO_DEFAULT,
           _IMAGE_DEFAULT, _IMAGE_DEFAULT, struct.pack("<I", 0), _LASER_SCAN_DEFAULT, bytes(24)]
    return b"".join(out)


def _skip_camera_info(r):
    r.header(); r.get("<II"); r.str(); r.bytes(8 * r.get("<I")); r.bytes(240); r.bytes(8); r.bytes(17)


def _skip_image(r):
    r.header(); r.get("<II"); r.str(); r.bytes(5); r.bytes(r.get("<I"))


def decode_sensor(r):
    start = r.o
    h = r.header()
    s = dict(stamp_sec=h["sec"], stamp_nsec=h["nsec"], sensor_frame=h["frame_id"])       # sensor_data.cpp:52-58
    s["sensor_type"] = r.get("<i")
    s["displacement"] = r.pose()
    r.str()
    r.header()
    s["descriptor_type"] = r.get("<i")
    n = r.get("<I")
    rec0 = r.o
    D, uniform = 0, 1
    for i in range(n):
        r.bytes(13)
        d = r.get("<I")
        if i == 0:
            D = d
        elif d != D:
            uniform = 0
        r.bytes(4 * d + 24)
    s.update(n_features=n, desc_len=D, uniform=uniform, records=bytes(r.b[rec0:r.o]))
    c0 = r.o
    _skip_camera_info(r)
    s["camera_info"] = bytes(r.b[c0:r.o])
    _skip_image(r); _skip_image(r)
    r.bytes(4 * r.get("<I"))
    r.header(); r.bytes(28); r.bytes(4 * r.get("<I")); r.bytes(4 * r.get("<I"))
    r.bytes(24)
    s["raw"] = bytes(r.b[start:r.o])
    return s


def encode_node(n):
    """Conversions::toMsg(SlamNode) (conversions.cpp:299-322); Node.msg field order."""
    out = [struct.pack("<I", len(n["stamps_ns"]))]
    for t in n["stamps_ns"]:
        out.append(struct.pack("<II", int(t) // 10**9, int(t) % 10**9))
    out += [_str(n["id"]), _pose_bytes(n["pose"]), _pose_bytes(n["odom_pose"]), _header(), struct.pack("<I", len(n["sensors"]))]
    out += [encode_sensor(s) for s in n["sensors"]]
    out.append(struct.pack("<I", len(n["edge_ids"])))
    out += [_str(e) for e in n["edge_ids"]]
    out.append(struct.pack("<Bd", 1 if n["fixed"] else 0, n["uncertainty"]))
    return b"".join(out)


def decode_node(b):
    """Conversions::fromMsg(Node) (conversions.cpp:276-297)."""
    r = _R(b)
    ns = r.get("<I")
    stamps = []
    for _ in range(ns):
        sec, nsec = r.get("<II")
        stamps.append(sec * 10**9 + nsec)
    n = dict(stamps_ns=stamps, id=r.str(), pose=r.pose(), odom_pose=r.pose())
    r.header()
    n["sensors"] = [decode_sensor(r) for _ in range(r.get("<I"))]
    n["edge_ids"] = [r.str() for _ in range(r.get("<I"))]
    n["fixed"] = int(r.get("<B") != 0)
    n["uncertainty"] = r.get("<d")
    return n, r.o


# ------------------------------------------------------------------------------------------------ rosbag 2.0 [EXT]
BAG_MAGIC = b"#ROSBAG V2.0\n"


def _fields(d):
    """record header: `name=value` fields, each with a u32 length; rosbag's C++ writer emits them in name order (std::map)"""
    return b"".join(struct.pack("<I", len(k) + 1 + len(v)) + k + b"=" + v for k, v in sorted(d.items()))


def _record(hdr, data):
    h = _fields(hdr)
    return struct.pack("<I", len(h)) + h + struct.pack("<I", len(data)) + data


def bag_write_single(topic, datatype, md5sum, definition, sec, nsec, data):
    """rosbag::Bag::open(Write) / write(topic, time, msg) / close() for one message (rosbag_storage.cpp:62-76): file header
    record (header + padding = 4096 bytes: data_len = 4096 - header_len as rosbag::Bag::writeFileHeaderRecord sets it), one uncompressed chunk (connection + message), its index record, then the index section
    (connection, chunk info)."""
    t = struct.pack("<II", sec, nsec)
    conn = _record({b"op": b"\x07", b"conn": struct.pack("<I", 0), b"topic": topic},
                   _fields({b"type": datatype, b"md5sum": md5sum, b"message_definition": definition}))
    msg = _record({b"op": b"\x02", b"conn": struct.pack("<I", 0), b"time": t}, data)
    chunk = _record({b"op": b"\x05", b"compression": b"none", b"size": struct.pack("<I", len(conn) + len(msg))}, conn + msg)
    index = _record({b"op": b"\x04", b"ver": struct.pack("<I", 1), b"conn": struct.pack("<I", 0), b"count": struct.pack("<I", 1)},
                    t + struct.pack("<I", len(conn)))
    chunk_pos = len(BAG_MAGIC) + 4 + 4 + 4096          # 4117
    info = _record({b"op": b"\x06", b"ver": struct.pack("<I", 1), b"chunk_pos": struct.pack("<Q", chunk_pos), b"start_time": t,
                    b"end_time": t, b"count": struct.pack("<I", 1)}, struct.pack("<II", 0, 1))
    index_pos = chunk_pos + len(chunk) + len(index)
    h = _fields({b"op": b"\x03", b"index_pos": struct.pack("<Q", index_pos), b"conn_count": struct.pack("<I", 1),
                 b"chunk_count": struct.pack("<I", 1)})
    pad = 4096 - len(h)
    head = struct.pack("<I", len(h)) + h + struct.pack("<I", pad) + b" " * pad
    return BAG_MAGIC + head + chunk + index + conn + info


def _parse_fields(b):
    d, o = {}, 0
    while o < len(b):
        (l,) = struct.unpack_from("<I", b, o)
        f = bytes(b[o + 4:o + 4 + l])
        k, v = f.split(b"=", 1)
        d[k] = v
        o += 4 + l
    return d


def _records(b):
    o = 0
    while o < len(b):
        (hl,) = struct.unpack_from("<I", b, o)
        hdr = _parse_fields(b[o + 4:o + 4 + hl])
        o += 4 + hl
        (dl,) = struct.unpack_from("<I", b, o)
        if o + 4 + dl > len(b):
            raise ValueError("truncated")
        yield hdr, b[o + 4:o + 4 + dl]
        o += 4 + dl


def bag_read(b):
    """Every message of an uncompressed bag image in file order: dicts topic, datatype, md5sum, definition, sec, nsec, data."""
    if bytes(b[:13]) != BAG_MAGIC:
        raise ValueError("not a rosbag 2.0 file")
    b = bytes(b)
    conns, msgs = {}, []

    def walk(data, inside, collect):
        for hdr, d in _records(data):
            op = hdr[b"op"][0]
            if op == 5 and not inside:
                if hdr[b"compression"] != b"none":
                    raise NotImplementedError("compressed chunk")
                walk(d, True, collect)
            elif op == 7 and not collect:
                c = struct.unpack("<I", hdr[b"conn"])[0]
                info = {k.decode(): v for k, v in _parse_fields(d).items()}
                info["topic"] = hdr.get(b"topic", b"")
                conns.setdefault(c, info)
            elif op == 2 and collect:
                c = conns[struct.unpack("<I", hdr[b"conn"])[0]]
                sec, nsec = struct.unpack("<II", hdr[b"time"])
                msgs.append(dict(topic=c["topic"], datatype=c.get("type", b""), md5sum=c.get("md5sum", b""),
                                 definition=c.get("message_definition", b""), sec=sec, nsec=nsec, data=bytes(d)))

    walk(b[13:], False, False)
    walk(b[13:], False, True)
    return msgs
