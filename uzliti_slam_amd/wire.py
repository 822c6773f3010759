"""Wire / disk formats either side of the path: ctypes binding of uzl_wire_* / uzl_bag_* and a host-side
mirror of RosbagStorage (graph_slam_common/src/rosbag_storage.cpp:36-209: one rosbag per node / edge under
<path>/nodes/<id> and <path>/edges/<id>, topics "node" / "edge").

All conversion work is done by libuzl_mi355x.so: message headers and strings on the host, the Feature[] payload of every
node on the device (Match.add_frames_wire / frame_to_wire).  This module only moves bytes between files and the C ABI.
"""
import ctypes as C
import os

import numpy as np

from . import capi

UZL_ERR_TRUNCATED = -9
UZL_ERR_UNSUPPORTED = -10
SENSOR_TYPE_FEATURE = 1


class Span(C.Structure):
    _fields_ = [("p", C.c_void_p), ("n", C.c_uint64)]


class WireEdge(C.Structure):
    _fields_ = [("id", Span), ("id_from", Span), ("id_to", Span), ("sensor_from", Span), ("sensor_to", Span),
                ("type", C.c_int32), ("valid", C.c_int32), ("transform", C.c_double * 12), ("information", C.c_double * 36),
                ("displacement_from", C.c_double * 12), ("displacement_to", C.c_double * 12),
                ("error", C.c_double), ("age", C.c_double), ("matching_score", C.c_double),
                ("diff_time_sec", C.c_int32), ("diff_time_nsec", C.c_int32)]


class WireSensor(C.Structure):
    _fields_ = [("raw", Span), ("sensor_type", C.c_int32), ("stamp_sec", C.c_uint32), ("stamp_nsec", C.c_uint32),
                ("sensor_frame", Span), ("displacement", C.c_double * 12), ("descriptor_type", C.c_int32),
                ("n_features", C.c_int32), ("desc_len", C.c_int32), ("uniform", C.c_int32), ("records", Span),
                ("camera_info", Span)]


class WireNode(C.Structure):
    _fields_ = [("id", Span), ("pose", C.c_double * 12), ("odom_pose", C.c_double * 12), ("fixed", C.c_int32),
                ("n_stamps", C.c_int32), ("n_edge_ids", C.c_int32), ("n_sensors", C.c_int32), ("uncertainty", C.c_double)]


class WireSensorTransform(C.Structure):
    _fields_ = [("sensor_name", Span), ("transform", C.c_double * 12)]


class WireMeta(C.Structure):
    _fields_ = [("stamp_sec", C.c_uint32), ("stamp_nsec", C.c_uint32), ("frame_id", Span), ("name", Span),
                ("map_transform", C.c_double * 12), ("n_sensor_transforms", C.c_int32), ("n_sensor_transforms_initial", C.c_int32),
                ("odometry_parameters", C.c_double * 6)]


class BagMsg(C.Structure):
    _fields_ = [("topic", Span), ("datatype", Span), ("md5sum", Span), ("definition", Span), ("data", Span),
                ("time_sec", C.c_uint32), ("time_nsec", C.c_uint32)]


_proto_done = False


def _lib():
    global _proto_done
    L = capi.lib()
    if not _proto_done:
        for f in ("uzl_wire_edge_size", "uzl_wire_node_size", "uzl_wire_meta_size", "uzl_wire_features_size", "uzl_bag_single_size"):
            getattr(L, f).restype = C.c_uint64
        _proto_done = True
    return L


def _check(rc, what):
    if rc != capi.UZL_OK:
        raise capi.UzlError(rc, what + ": " + capi.lib().uzl_status_string(rc).decode())


class _Keep:
    """keeps the Python buffers alive that spans point into"""

    def __init__(self):
        self.refs = []

    def span(self, b):
        if b is None:
            return Span(None, 0)
        if isinstance(b, str):
            b = b.encode()
        buf = C.create_string_buffer(bytes(b), max(len(b), 1))
        self.refs.append(buf)
        return Span(C.cast(buf, C.c_void_p).value, len(b))


def _bytes(s):
    return C.string_at(s.p, s.n) if s.p and s.n else b""


def _arr(a, n):
    return np.asarray(a, np.float64).reshape(n).tolist()


# ---------------------------------------------------------------------------------------------------- Edge
_EDGE_STR = ("id", "id_from", "id_to", "sensor_from", "sensor_to")
_EDGE_ARR = (("transform", 12), ("information", 36), ("displacement_from", 12), ("displacement_to", 12))
_EDGE_NUM = ("type", "valid", "error", "age", "matching_score", "diff_time_sec", "diff_time_nsec")


def encode_edge(e):
    """SlamEdge fields (dict) -> serialised graph_slam_msgs/Edge."""
    L = _lib()
    k = _Keep()
    w = WireEdge()
    for f in _EDGE_STR:
        setattr(w, f, k.span(e[f]))
    for f, n in _EDGE_ARR:
        getattr(w, f)[:] = _arr(e[f], n)
    for f in _EDGE_NUM:
        setattr(w, f, e[f])
    size = L.uzl_wire_edge_size(C.byref(w))
    buf = (C.c_uint8 * size)()
    wr = C.c_uint64(0)
    _check(L.uzl_wire_edge_encode(C.byref(w), buf, C.c_uint64(size), C.byref(wr)), "edge_encode")
    assert wr.value == size
    return bytes(buf)


def decode_edge(b):
    L = _lib()
    src = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) or b"\0")
    w = WireEdge()
    used = C.c_uint64(0)
    _check(L.uzl_wire_edge_decode(src, C.c_uint64(len(b)), C.byref(w), C.byref(used)), "edge_decode")
    e = {f: _bytes(getattr(w, f)) for f in _EDGE_STR}
    for f, n in _EDGE_ARR:
        e[f] = np.array(getattr(w, f)[:])
    for f in _EDGE_NUM:
        e[f] = getattr(w, f)
    return e, used.value


# ---------------------------------------------------------------------------------------------------- GraphMeta
def _transforms_in(k, items):
    arr = (WireSensorTransform * max(len(items), 1))()
    for i, (name, T) in enumerate(items):
        arr[i].sensor_name = k.span(name)
        arr[i].transform[:] = _arr(T, 12)
    return arr


def encode_meta(m):
    """The graph's meta data (dict: stamp_sec, stamp_nsec, frame_id, name, map_transform, sensor_transforms /
    sensor_transforms_initial = [(name, 12 doubles)], odometry_parameters) -> serialised graph_slam_msgs/GraphMeta."""
    L = _lib()
    k = _Keep()
    w = WireMeta()
    w.stamp_sec, w.stamp_nsec = int(m["stamp_sec"]), int(m["stamp_nsec"])
    w.frame_id, w.name = k.span(m["frame_id"]), k.span(m["name"])
    w.map_transform[:] = _arr(m["map_transform"], 12)
    w.n_sensor_transforms, w.n_sensor_transforms_initial = len(m["sensor_transforms"]), len(m["sensor_transforms_initial"])
    w.odometry_parameters[:] = _arr(m["odometry_parameters"], 6)
    st, sti = _transforms_in(k, m["sensor_transforms"]), _transforms_in(k, m["sensor_transforms_initial"])
    size = L.uzl_wire_meta_size(C.byref(w), st, sti)
    buf = (C.c_uint8 * size)()
    wr = C.c_uint64(0)
    _check(L.uzl_wire_meta_encode(C.byref(w), st, sti, buf, C.c_uint64(size), C.byref(wr)), "meta_encode")
    assert wr.value == size
    return bytes(buf)


def decode_meta(b):
    L = _lib()
    src = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) or b"\0")
    w = WireMeta()
    used = C.c_uint64(0)
    _check(L.uzl_wire_meta_decode(src, C.c_uint64(len(b)), C.byref(w), 0, None, 0, None, C.byref(used)), "meta_decode")      # counts
    st = (WireSensorTransform * max(w.n_sensor_transforms, 1))()
    sti = (WireSensorTransform * max(w.n_sensor_transforms_initial, 1))()
    _check(L.uzl_wire_meta_decode(src, C.c_uint64(len(b)), C.byref(w), w.n_sensor_transforms, st, w.n_sensor_transforms_initial, sti,
                                  C.byref(used)), "meta_decode")
    m = dict(stamp_sec=w.stamp_sec, stamp_nsec=w.stamp_nsec, frame_id=_bytes(w.frame_id), name=_bytes(w.name),
             map_transform=np.array(w.map_transform[:]), odometry_parameters=np.array(w.odometry_parameters[:]))
    m["sensor_transforms"] = [(_bytes(st[i].sensor_name), np.array(st[i].transform[:])) for i in range(w.n_sensor_transforms)]
    m["sensor_transforms_initial"] = [(_bytes(sti[i].sensor_name), np.array(sti[i].transform[:])) for i in range(w.n_sensor_transforms_initial)]
    return m, used.value


# ---------------------------------------------------------------------------------------------------- Node
def _sensor_in(k, s):
    w = WireSensor()
    if s.get("raw") is not None:
        w.raw = k.span(s["raw"])
        return w
    w.sensor_type = s["sensor_type"]; w.stamp_sec = s["stamp_sec"]; w.stamp_nsec = s["stamp_nsec"]
    w.sensor_frame = k.span(s["sensor_frame"])
    w.displacement[:] = _arr(s["displacement"], 12)
    w.descriptor_type = s["descriptor_type"]; w.n_features = s["n_features"]; w.desc_len = s["desc_len"]; w.uniform = 1
    w.records = k.span(s.get("records", b""))
    w.camera_info = k.span(s.get("camera_info"))
    return w


def _sensor_out(w):
    return dict(raw=_bytes(w.raw), sensor_type=w.sensor_type, stamp_sec=w.stamp_sec, stamp_nsec=w.stamp_nsec,
                sensor_frame=_bytes(w.sensor_frame), displacement=np.array(w.displacement[:]), descriptor_type=w.descriptor_type,
                n_features=w.n_features, desc_len=w.desc_len, uniform=w.uniform, records=_bytes(w.records),
                camera_info=_bytes(w.camera_info))


def encode_node(n):
    """SlamNode fields (dict: id, stamps_ns, pose, odom_pose, sensors, edge_ids, fixed, uncertainty) -> graph_slam_msgs/Node."""
    L = _lib()
    k = _Keep()
    w = WireNode()
    w.id = k.span(n["id"])
    w.pose[:] = _arr(n["pose"], 12); w.odom_pose[:] = _arr(n["odom_pose"], 12)
    w.fixed = int(n["fixed"]); w.uncertainty = float(n["uncertainty"])
    w.n_stamps = len(n["stamps_ns"]); w.n_edge_ids = len(n["edge_ids"]); w.n_sensors = len(n["sensors"])
    stamps = (C.c_int64 * max(w.n_stamps, 1))(*[int(t) for t in n["stamps_ns"]])
    eids = (Span * max(w.n_edge_ids, 1))(*[k.span(e) for e in n["edge_ids"]])
    sens = (WireSensor * max(w.n_sensors, 1))(*[_sensor_in(k, s) for s in n["sensors"]])
    size = L.uzl_wire_node_size(C.byref(w), eids, sens)
    buf = (C.c_uint8 * size)()
    wr = C.c_uint64(0)
    _check(L.uzl_wire_node_encode(C.byref(w), stamps, eids, sens, buf, C.c_uint64(size), C.byref(wr)), "node_encode")
    assert wr.value == size
    return bytes(buf)


class DecodedNode:
    """Result of decode_node: `fields` (dict) plus the ctypes sensor array whose spans point into the kept source buffer
    (hand `sensors_c` / `feature_index` to Match.add_frames_wire without copying the Feature records)."""

    def __init__(self, fields, src, sensors_c, used):
        self.fields, self._src, self.sensors_c, self.used = fields, src, sensors_c, used


def decode_node(b):
    L = _lib()
    src = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) or b"\0")
    w = WireNode()
    used = C.c_uint64(0)
    _check(L.uzl_wire_node_decode(src, C.c_uint64(len(b)), C.byref(w), 0, None, 0, None, 0, None, C.byref(used)), "node_decode")
    stamps = (C.c_int64 * max(w.n_stamps, 1))()
    eids = (Span * max(w.n_edge_ids, 1))()
    sens = (WireSensor * max(w.n_sensors, 1))()
    _check(L.uzl_wire_node_decode(src, C.c_uint64(len(b)), C.byref(w), w.n_stamps, stamps, w.n_edge_ids, eids, w.n_sensors, sens,
                                  C.byref(used)), "node_decode")
    f = dict(id=_bytes(w.id), pose=np.array(w.pose[:]), odom_pose=np.array(w.odom_pose[:]), fixed=w.fixed, uncertainty=w.uncertainty,
             stamps_ns=[stamps[i] for i in range(w.n_stamps)], edge_ids=[_bytes(eids[i]) for i in range(w.n_edge_ids)],
             sensors=[_sensor_out(sens[i]) for i in range(w.n_sensors)])
    return DecodedNode(f, src, sens, used.value)


def features_size(n, desc_len):
    return _lib().uzl_wire_features_size(C.c_int32(n), C.c_int32(desc_len))


# ---------------------------------------------------------------------------------------------------- rosbag
def bag_write_single(topic, datatype, md5sum, definition, sec, nsec, data):
    L = _lib()
    k = _Keep()
    m = BagMsg(k.span(topic), k.span(datatype), k.span(md5sum), k.span(definition), k.span(data), sec, nsec)
    size = L.uzl_bag_single_size(C.byref(m))
    buf = (C.c_uint8 * size)()
    wr = C.c_uint64(0)
    _check(L.uzl_bag_write_single(C.byref(m), buf, C.c_uint64(size), C.byref(wr)), "bag_write_single")
    assert wr.value == size
    return bytes(buf)


def bag_read(b):
    L = _lib()
    src = (C.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) or b"\0")
    n = C.c_int32(0)
    _check(L.uzl_bag_read(src, C.c_uint64(len(b)), 0, None, C.byref(n)), "bag_read")
    msgs = (BagMsg * max(n.value, 1))()
    _check(L.uzl_bag_read(src, C.c_uint64(len(b)), n.value, msgs, C.byref(n)), "bag_read")
    return [dict(topic=_bytes(m.topic), datatype=_bytes(m.datatype), md5sum=_bytes(m.md5sum), definition=_bytes(m.definition),
                 sec=m.time_sec, nsec=m.time_nsec, data=_bytes(m.data)) for m in msgs[:n.value]]


# ---------------------------------------------------------------------------------------------------- frames on the device
def add_frames_wire(match, sensors_c, count, sensor_frame_keys=None, want_uv=False):
    """capi.Match + a ctypes WireSensor array (from decode_node) -> frame ids (and u,v per keypoint)."""
    ids = (C.c_int32 * max(count, 1))()
    keys = (C.c_int32 * max(count, 1))(*(sensor_frame_keys or [0] * count))
    total = sum(sensors_c[i].n_features for i in range(count))
    uv = np.zeros((max(total, 1), 2), np.int32) if want_uv else None
    match._check(_lib().uzl_match_add_frames_wire(match._h, C.c_int32(count), sensors_c, keys, ids,
                                                  uv.ctypes.data_as(capi.c_i32p) if want_uv else None))
    return [ids[i] for i in range(count)], (uv[:total] if want_uv else None)


def frame_to_wire(match, frame_id, uv=None):
    L = _lib()
    need = C.c_uint64(0)
    n, bpd = C.c_int32(0), C.c_int32(0)
    match._check(L.uzl_match_get_frame(match._h, C.c_int32(frame_id), None, None, None, C.byref(n), C.byref(bpd)))
    size = features_size(n.value, bpd.value)
    buf = (C.c_uint8 * max(size, 1))()
    u = np.ascontiguousarray(uv, np.int32) if uv is not None else None
    match._check(L.uzl_match_frame_to_wire(match._h, C.c_int32(frame_id), u.ctypes.data_as(capi.c_i32p) if u is not None else None,
                                           buf, C.c_uint64(size), C.byref(need)))
    return bytes(buf[:need.value])


def get_frame(match, frame_id):
    L = _lib()
    n, bpd = C.c_int32(0), C.c_int32(0)
    match._check(L.uzl_match_get_frame(match._h, C.c_int32(frame_id), None, None, None, C.byref(n), C.byref(bpd)))
    desc = np.zeros((n.value, bpd.value), np.uint8); pos = np.zeros((n.value, 3), np.float64); valid = np.zeros(n.value, np.uint8)
    match._check(L.uzl_match_get_frame(match._h, C.c_int32(frame_id), desc.ctypes.data_as(capi.c_u8p), pos.ctypes.data_as(capi.c_f64p),
                                       valid.ctypes.data_as(capi.c_u8p), C.byref(n), C.byref(bpd)))
    return desc, np.ascontiguousarray(pos.T), valid


# ---------------------------------------------------------------------------------------------------- RosbagStorage mirror
class RosbagStorage:
    """RosbagStorage (rosbag_storage.cpp): storeNode / storeEdge write one bag per object, removeNode / removeEdge delete it,
    loadGraph reads every file of nodes/ and edges/.  md5sum / definition are the message traits of the caller's ROS build
    (ros::message_traits::MD5Sum<M> / Definition<M>); they are stored, never interpreted."""

    NODE_TYPE, EDGE_TYPE, META_TYPE = b"graph_slam_msgs/Node", b"graph_slam_msgs/Edge", b"graph_slam_msgs/GraphMeta"

    def __init__(self, storage_path, clear_storage=False, traits=None):
        self.path = storage_path
        self.traits = traits or {}
        self.initialize(storage_path, clear_storage)

    def initialize(self, storage_path, clear_storage):                   # rosbag_storage.cpp:211-235
        import shutil
        if clear_storage and os.path.isdir(storage_path):
            shutil.rmtree(storage_path)
        for d in ("nodes", "edges", "meta"):
            os.makedirs(os.path.join(storage_path, d), exist_ok=True)

    def clear(self):                                                     # :54-60
        self.initialize(self.path, True)

    def _write(self, sub, name, topic, datatype, data, now_ns):
        md5, definition = self.traits.get(datatype, (b"*", b""))
        t = now_ns + 1                                                   # ros::Time::now() + ros::Duration(0, 1) (:73)
        img = bag_write_single(topic, datatype, md5, definition, t // 10**9, t % 10**9, data)
        with open(os.path.join(self.path, sub, name if isinstance(name, str) else name.decode()), "wb") as f:
            f.write(img)

    def store_node(self, node, now_ns=0):                                # storeNode (:62-76)
        self._write("nodes", node["id"], b"node", self.NODE_TYPE, encode_node(node), now_ns)

    def store_edge(self, edge, now_ns=0):                                # storeEdge (:78-92)
        self._write("edges", edge["id"], b"edge", self.EDGE_TYPE, encode_edge(edge), now_ns)

    def store_meta(self, meta, now_ns=0):                                # storeMetaData (:94-107): <path>/meta/meta, topic "meta"
        self._write("meta", "meta", b"meta", self.META_TYPE, encode_meta(meta), now_ns)

    def load_meta(self):
        """loadGraph's meta pass (:187-207): every file under <path>/meta, every "meta" message, the last one read wins
        (updateMetaData overwrites); None when there is none."""
        meta = None
        d = os.path.join(self.path, "meta")
        for name in sorted(os.listdir(d)):
            with open(os.path.join(d, name), "rb") as f:
                for m in bag_read(f.read()):
                    if m["topic"] == b"meta":
                        meta, _ = decode_meta(m["data"])
        return meta

    def _remove(self, sub, name):
        p = os.path.join(self.path, sub, name if isinstance(name, str) else name.decode())
        if os.path.exists(p):
            os.remove(p)

    def remove_node(self, name):                                         # removeNode (:110-122)
        self._remove("nodes", name)

    def remove_edge(self, name):                                         # removeEdge (:124-136)
        self._remove("edges", name)

    def load_graph(self, match=None, sensor_frame_key=None):
        """loadGraph (:138-209): nodes (first "node" message of each file, :149-156) and edges (every "edge" message).
        With a capi.Match the feature frames of ALL nodes are unpacked on the device by one launch; every node dict then
        carries `frame_ids` (one per FEATURE sensor, in order)."""
        nodes, keep = {}, []
        for name in sorted(os.listdir(os.path.join(self.path, "nodes"))):
            with open(os.path.join(self.path, "nodes", name), "rb") as f:
                for m in bag_read(f.read()):
                    if m["topic"] == b"node":
                        d = decode_node(m["data"])
                        nodes[d.fields["id"]] = d.fields
                        keep.append(d)
                        break
        edges = {}
        for name in sorted(os.listdir(os.path.join(self.path, "edges"))):
            with open(os.path.join(self.path, "edges", name), "rb") as f:
                for m in bag_read(f.read()):
                    if m["topic"] == b"edge":
                        e, _ = decode_edge(m["data"])
                        edges[e["id"]] = e
        if match is not None:
            batch, owner = [], []
            for d in keep:
                d.fields["frame_ids"] = []
                for i in range(len(d.fields["sensors"])):
                    if d.sensors_c[i].sensor_type == SENSOR_TYPE_FEATURE and d.sensors_c[i].n_features > 0:
                        batch.append(d.sensors_c[i]); owner.append(d)
            if batch:
                arr = (WireSensor * len(batch))(*batch)
                keys = [sensor_frame_key(_bytes(s.sensor_frame)) if sensor_frame_key else 0 for s in batch]
                ids, _ = add_frames_wire(match, arr, len(batch), keys)
                for d, fid in zip(owner, ids):
                    d.fields["frame_ids"].append(fid)
            self._keep = keep
        return nodes, edges
