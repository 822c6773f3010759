"""ctypes binding of libuzl_mi355x.so (include/uzl_mi355x.h) — the only way Python reaches the HIP path.

There is no fallback: if the shared library has not been built, or no HIP device is visible when a
handle is created, the calls raise.  Tests and bench.py drive the product exclusively through this C ABI.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UZL_LIB", os.path.join(_HERE, "libuzl_mi355x.so"))   # UZL_LIB: diagnostic builds only
CSRC = os.path.join(_HERE, "csrc")

c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
c_u64p = C.POINTER(C.c_uint64)

UZL_OK = 0
UZL_ERR_BAD_ARG = -1
UZL_ERR_NO_DEVICE = -2
UZL_ERR_HIP = -3
UZL_ERR_NOT_CONVERGED = -4
UZL_ERR_BUSY = -5
UZL_ERR_OOM = -6
UZL_ERR_NOT_FOUND = -7
UZL_ERR_STATE = -8


class UzlError(RuntimeError):
    def __init__(self, status, msg=""):
        super().__init__(f"uzl status {status}: {msg}")
        self.status = status


class MatchCfg(C.Structure):
    _fields_ = [("ransac_threshold", C.c_double), ("link_covariance", C.c_double),
                ("ransac_iteration", C.c_int32), ("ransac_break_percentage", C.c_double),
                ("use_epnp", C.c_int32), ("do_prosac", C.c_int32), ("device", C.c_int32),
                ("seed", C.c_uint64)]


class Frame(C.Structure):
    _fields_ = [("desc", c_u8p), ("n", C.c_int32), ("bytes_per_desc", C.c_int32),
                ("pos_xyz", c_f64p), ("valid3d", c_u8p), ("feature_type", C.c_int32),
                ("sensor_frame", C.c_int32), ("displacement", C.c_double * 12)]


class PairJob(C.Structure):
    _fields_ = [("job_id", C.c_uint64), ("from_begin", C.c_int32), ("from_count", C.c_int32),
                ("to_begin", C.c_int32), ("to_count", C.c_int32)]


class EdgeResult(C.Structure):
    _fields_ = [("job_id", C.c_uint64), ("ok", C.c_int32), ("consensus", C.c_int32),
                ("n_matches", C.c_int32), ("n_corr", C.c_int32), ("frame_from", C.c_int32),
                ("frame_to", C.c_int32), ("iterations_run", C.c_int32), ("best_iteration", C.c_int32),
                ("mse", C.c_double), ("T", C.c_double * 12), ("information", C.c_double * 36)]


EDGE_RESULT_DTYPE = np.dtype([("job_id", "<u8"), ("ok", "<i4"), ("consensus", "<i4"), ("n_matches", "<i4"),
                              ("n_corr", "<i4"), ("frame_from", "<i4"), ("frame_to", "<i4"),
                              ("iterations_run", "<i4"), ("best_iteration", "<i4"), ("mse", "<f8"),
                              ("T", "<f8", (12,)), ("information", "<f8", (36,))], align=True)
FRAME_DTYPE = np.dtype([("desc", "<u8"), ("n", "<i4"), ("bytes_per_desc", "<i4"), ("pos_xyz", "<u8"), ("valid3d", "<u8"),
                        ("feature_type", "<i4"), ("sensor_frame", "<i4"), ("displacement", "<f8", (12,))], align=True)      # = Frame / uzl_frame
PAIR_JOB_DTYPE = np.dtype([("job_id", "<u8"), ("from_begin", "<i4"), ("from_count", "<i4"),
                           ("to_begin", "<i4"), ("to_count", "<i4")], align=True)


class PgoCfg(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("use_odometry_parameters", C.c_int32),
                ("optimize_xy_only", C.c_int32), ("device", C.c_int32), ("pcg_tol", C.c_double),
                ("pcg_max_iter", C.c_int32), ("schur_reduce", C.c_int32), ("huber_delta", C.c_double), ("verbose", C.c_int32),
                ("preconditioner", C.c_int32), ("pcg_stop", C.c_int32), ("lm_loop", C.c_int32), ("reduced_numbering", C.c_int32), ("pass_history", C.c_int32)]


class PgoStats(C.Structure):
    _fields_ = [("iterations_done", C.c_int32), ("lm_trials", C.c_int32), ("pcg_iterations", C.c_int32),
                ("terminated_early", C.c_int32), ("n_vertices", C.c_int32), ("n_edges", C.c_int32),
                ("n_gauge_fixed", C.c_int32), ("pcg_not_converged", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
                ("solve_ms", C.c_double), ("precond_builds", C.c_int32), ("exchange_calls", C.c_int32),
                ("structure_ms", C.c_double), ("exchange_ms", C.c_double), ("structure_reused", C.c_int32), ("n_eliminated", C.c_int32),
                ("lm_passes", C.c_int32), ("reduced_strong", C.c_int32)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


NODE_DTYPE = np.dtype([("pose", "<f8", (12,)), ("fixed", "<i4")], align=True)
EDGE_DTYPE = np.dtype([("from", "<i4"), ("to", "<i4"), ("type", "<i4"), ("sensor_from", "<i4"),
                       ("sensor_to", "<i4"), ("valid", "<i4"), ("transform", "<f8", (12,)),
                       ("displacement_from", "<f8", (12,)), ("displacement_to", "<f8", (12,)),
                       ("information", "<f8", (36,)), ("diff_time", "<f8")], align=True)

class FilterCfg(C.Structure):
    _fields_ = [("max_dt", C.c_double), ("min_size", C.c_double), ("max_cluster_size", C.c_int32),
                ("ransac_iterations", C.c_int32), ("max_error", C.c_double), ("min_time_span", C.c_double),
                ("max_edges", C.c_int32), ("device", C.c_int32), ("seed", C.c_uint64)]


class FilterEdge(C.Structure):
    _fields_ = [("key", C.c_uint64), ("matching_score", C.c_double), ("valid", C.c_int32),
                ("sensor_from", C.c_int32), ("sensor_to", C.c_int32), ("n_stamps_from", C.c_int32),
                ("n_stamps_to", C.c_int32), ("_pad", C.c_int32),
                ("stamps_from_ns", C.POINTER(C.c_int64)), ("stamps_to_ns", C.POINTER(C.c_int64)),
                ("transform", C.c_double * 12), ("displacement_from", C.c_double * 12),
                ("displacement_to", C.c_double * 12), ("pose_from", C.c_double * 12), ("pose_to", C.c_double * 12)]


# numpy mirror of uzl_filter_edge (pointers as addresses): batches of thousands of edges are packed without a Python loop
FILTER_EDGE_DTYPE = np.dtype([("key", "<u8"), ("matching_score", "<f8"), ("valid", "<i4"), ("sensor_from", "<i4"), ("sensor_to", "<i4"),
                              ("n_stamps_from", "<i4"), ("n_stamps_to", "<i4"), ("_pad", "<i4"), ("stamps_from_ns", "<u8"),
                              ("stamps_to_ns", "<u8"), ("transform", "<f8", (12,)), ("displacement_from", "<f8", (12,)),
                              ("displacement_to", "<f8", (12,)), ("pose_from", "<f8", (12,)), ("pose_to", "<f8", (12,))], align=True)
assert FILTER_EDGE_DTYPE.itemsize == C.sizeof(FilterEdge)


class ClusterInfo(C.Structure):
    _fields_ = [("uid", C.c_uint64), ("from_start_ns", C.c_int64), ("from_end_ns", C.c_int64),
                ("to_start_ns", C.c_int64), ("to_end_ns", C.c_int64), ("size", C.c_int32),
                ("consensus", C.c_int32), ("changed", C.c_int32), ("evaluations", C.c_int32)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


_IDENT12 = (1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0)


def pack_filter_edges(edges, struct=None):
    """list of dicts (key, matching_score, valid, sensor_from, sensor_to, stamps_from, stamps_to, transform,
    displacement_from, displacement_to, pose_from, pose_to) -> (ctypes array, keep-alive list).  3x4 row-major
    transforms (12 doubles); stamps are int64 nanoseconds."""
    struct = struct or FilterEdge
    arr = (struct * max(len(edges), 1))()
    keep = []
    for i, e in enumerate(edges):
        a = arr[i]
        a.key = int(e["key"]); a.matching_score = float(e.get("matching_score", 0.0)); a.valid = int(e.get("valid", 0))
        a.sensor_from = int(e.get("sensor_from", -1)); a.sensor_to = int(e.get("sensor_to", -1))
        sf = np.ascontiguousarray(e.get("stamps_from", ()), np.int64); st = np.ascontiguousarray(e.get("stamps_to", ()), np.int64)
        keep += [sf, st]
        a.n_stamps_from = len(sf); a.n_stamps_to = len(st)
        a.stamps_from_ns = sf.ctypes.data_as(C.POINTER(C.c_int64)); a.stamps_to_ns = st.ctypes.data_as(C.POINTER(C.c_int64))
        for f in ("transform", "displacement_from", "displacement_to", "pose_from", "pose_to"):
            v = e.get(f)
            getattr(a, f)[:] = _IDENT12 if v is None else tuple(np.asarray(v, np.float64).reshape(-1)[:12])
    return arr, keep


class GateCfg(C.Structure):
    _fields_ = [("min_matching_score", C.c_double), ("max_edge_distance_T", C.c_double), ("max_edge_distance_R", C.c_double),
                ("scope_size_factor", C.c_double), ("min_accept_valid", C.c_double), ("device", C.c_int32), ("_pad", C.c_int32)]


GATE_EDGE_DTYPE = np.dtype([("from", "<i4"), ("to", "<i4"), ("type", "<i4"), ("valid", "<i4"), ("matching_score", "<f8"),
                            ("transform", "<f8", (12,))], align=True)


def gate_edges(frm, to, typ, valid=None, score=None, transform=None):
    n = len(frm)
    a = np.zeros(max(n, 1), GATE_EDGE_DTYPE)
    a["from"][:n] = frm; a["to"][:n] = to; a["type"][:n] = typ
    if valid is not None:
        a["valid"][:n] = valid
    if score is not None:
        a["matching_score"][:n] = score
    a["transform"][:n] = np.eye(3, 4).reshape(12) if transform is None else np.asarray(transform, np.float64).reshape(n, 12)
    return a[:n] if n else a[:0]


class RadiusCfg(C.Structure):
    _fields_ = [("radius", C.c_double), ("new_edge_time", C.c_double), ("max_rotation_deg", C.c_double),
                ("device", C.c_int32), ("_pad", C.c_int32)]


class PlacesCfg(C.Structure):
    _fields_ = [("key_width", C.c_int32), ("min_rows_to_add", C.c_int32), ("T", C.c_double), ("k_nearest_neighbors", C.c_int32),
                ("device", C.c_int32), ("min_time_gap", C.c_double)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)

_lib = None


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the in-tree shared library (csrc/Makefile)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC])
    return LIB_PATH


def lib():
    """Load libuzl_mi355x.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UzlError(UZL_ERR_STATE, f"{LIB_PATH} is missing: run __graft_entry__.build() / make -C {CSRC}; "
                                          "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.uzl_status_string.restype = C.c_char_p
        L.uzl_match_last_error.restype = C.c_char_p
        L.uzl_match_last_error.argtypes = [C.c_void_p]
        L.uzl_match_destroy.restype = None
        L.uzl_match_destroy.argtypes = [C.c_void_p]
        L.uzl_match_cfg_default.restype = None
        if hasattr(L, "uzl_pgo_last_error"):
            L.uzl_pgo_last_error.restype = C.c_char_p
            L.uzl_pgo_last_error.argtypes = [C.c_void_p]
            L.uzl_pgo_destroy.restype = None
            L.uzl_pgo_destroy.argtypes = [C.c_void_p]
            L.uzl_pgo_cfg_default.restype = None
        if hasattr(L, "uzl_places_create"):
            L.uzl_places_last_error.restype = C.c_char_p
            L.uzl_places_last_error.argtypes = [C.c_void_p]
            L.uzl_places_destroy.restype = None
            L.uzl_places_destroy.argtypes = [C.c_void_p]
            L.uzl_places_cfg_default.restype = None
            L.uzl_places_count.argtypes = [C.c_void_p]
        if hasattr(L, "uzl_radius_create"):
            L.uzl_radius_last_error.restype = C.c_char_p
            L.uzl_radius_last_error.argtypes = [C.c_void_p]
            L.uzl_radius_destroy.restype = None
            L.uzl_radius_destroy.argtypes = [C.c_void_p]
            L.uzl_radius_cfg_default.restype = None
        if hasattr(L, "uzl_gate_create"):
            L.uzl_gate_last_error.restype = C.c_char_p
            L.uzl_gate_last_error.argtypes = [C.c_void_p]
            L.uzl_gate_destroy.restype = None
            L.uzl_gate_destroy.argtypes = [C.c_void_p]
            L.uzl_gate_cfg_default.restype = None
            L.uzl_gate_edge_count.argtypes = [C.c_void_p]
        if hasattr(L, "uzl_filter_create"):
            L.uzl_filter_last_error.restype = C.c_char_p
            L.uzl_filter_last_error.argtypes = [C.c_void_p]
            L.uzl_filter_destroy.restype = None
            L.uzl_filter_destroy.argtypes = [C.c_void_p]
            L.uzl_filter_cfg_default.restype = None
            L.uzl_filter_cluster_count.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def device_count():
    return lib().uzl_device_count()


DIAG_LIB_PATH = os.path.join(_HERE, "libuzl_mi355x_diag.so")
_diag = None


def diag_lib():
    """The diagnostic twin (same sources, -DUZL_DIAG): the only place the uzl_debug_* test hooks exist - the product library exports
    include/uzl_mi355x.h and nothing else.  Loaded beside the product library (both keep their symbols to themselves)."""
    global _diag
    if _diag is None:
        if not os.path.exists(DIAG_LIB_PATH):
            raise UzlError(UZL_ERR_STATE, f"{DIAG_LIB_PATH} is missing: make -C {CSRC} diag")
        _diag = C.CDLL(DIAG_LIB_PATH)
    return _diag


def stream_stats(device=0):
    """uzl_stream_stats: the device's stream pool (uzl_streams.hip)"""
    v = [C.c_int32() for _ in range(6)]
    ms = C.c_double()
    rc = lib().uzl_stream_stats(C.c_int32(device), *[C.byref(x) for x in v], C.byref(ms))
    if rc != UZL_OK:
        raise UzlError(rc, "uzl_stream_stats")
    return dict(zip(("pooled", "leased", "registered", "pairs_measured", "pairs_independent", "fallbacks"), [x.value for x in v]), probe_ms=ms.value)


RCCL_UNIQUE_ID_BYTES = 128


def rccl_unique_id():
    """ncclGetUniqueId through the library (rank 0); the bytes go to the other ranks by any channel."""
    buf = (C.c_char * RCCL_UNIQUE_ID_BYTES)()
    rc = lib().uzl_rccl_unique_id(buf, C.c_int32(RCCL_UNIQUE_ID_BYTES))
    if rc != UZL_OK:
        raise UzlError(rc, "uzl_rccl_unique_id")
    return bytes(buf.raw)


# --------------------------------------------------------------------------------------- estimator
class Match:
    """Thin object wrapper over the uzl_match_* C ABI."""

    def __init__(self, **cfg):
        L = lib()
        c = MatchCfg()
        L.uzl_match_cfg_default(C.byref(c))
        for k, v in cfg.items():
            setattr(c, k, v)
        self.cfg = c
        self._h = C.c_void_p()
        rc = L.uzl_match_create(C.byref(c), C.byref(self._h))
        if rc != UZL_OK:
            raise UzlError(rc, L.uzl_status_string(rc).decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().uzl_match_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != UZL_OK:
            raise UzlError(rc, lib().uzl_match_last_error(self._h).decode())

    def set_config(self, **cfg):
        for k, v in cfg.items():
            setattr(self.cfg, k, v)
        self._check(lib().uzl_match_set_config(self._h, C.byref(self.cfg)))

    def add_frame(self, desc, pos, valid, feature_type=2, sensor_frame=0):
        """desc (n,bytes) u8; pos (3,n) f64; valid (n) u8 -> frame id."""
        d = np.ascontiguousarray(desc, np.uint8)
        p = np.ascontiguousarray(np.asarray(pos, np.float64).T)   # (n,3) row-major == 3 x n column-major
        v = np.ascontiguousarray(valid, np.uint8)
        f = Frame()
        f.desc = _p(d, c_u8p); f.n = d.shape[0]; f.bytes_per_desc = d.shape[1] if d.ndim == 2 else 0
        f.pos_xyz = _p(p, c_f64p); f.valid3d = _p(v, c_u8p)
        f.feature_type = int(feature_type); f.sensor_frame = int(sensor_frame)
        f.displacement[:] = np.eye(3, 4).reshape(12).tolist()
        fid = C.c_int32(-1)
        self._check(lib().uzl_match_add_frame(self._h, C.byref(f), C.byref(fid)))
        return fid.value

    @staticmethod
    def pack_frames(frames, feature_type=2, sensor_frame=0):
        """[(desc, pos, valid), ...] -> (array of uzl_frame, keep-alive list): the marshalling a C++ caller does not have.  The structs are
        filled column by column through a numpy view of the same layout (field-by-field ctypes assignment was 18 us per frame)."""
        n = len(frames)
        arr = np.zeros(max(n, 1), FRAME_DTYPE)
        keep = []
        dp = np.empty(n, np.uint64); pp = np.empty(n, np.uint64); vp = np.empty(n, np.uint64); nn = np.empty(n, np.int32); bb = np.empty(n, np.int32)
        for k, (desc, pos, valid) in enumerate(frames):
            d = np.ascontiguousarray(desc, np.uint8)
            p = np.ascontiguousarray(np.asarray(pos, np.float64).T)
            v = np.ascontiguousarray(valid, np.uint8)
            keep.append((d, p, v))
            dp[k] = d.__array_interface__["data"][0]; pp[k] = p.__array_interface__["data"][0]; vp[k] = v.__array_interface__["data"][0]
            nn[k] = d.shape[0]; bb[k] = d.shape[1] if d.ndim == 2 else 0
        a = arr[:n]
        a["desc"] = dp; a["pos_xyz"] = pp; a["valid3d"] = vp; a["n"] = nn; a["bytes_per_desc"] = bb
        a["feature_type"] = int(feature_type); a["sensor_frame"] = int(sensor_frame); a["displacement"] = np.eye(3, 4).reshape(12)
        return a, keep

    def add_frames(self, packed):
        """uzl_match_add_frames over the array pack_frames built -> list of frame ids."""
        arr = packed[0] if isinstance(packed, tuple) else packed
        n = len(arr)
        ids = (C.c_int32 * max(n, 1))()
        self._check(lib().uzl_match_add_frames(self._h, C.c_int32(n), _p(arr, C.c_void_p) if isinstance(arr, np.ndarray) else arr, ids))
        return list(ids[:n])

    def arena_bytes(self):
        live = C.c_uint64(); hw = C.c_uint64(); cap = C.c_uint64()
        self._check(lib().uzl_match_arena_bytes(self._h, C.byref(live), C.byref(hw), C.byref(cap)))
        return dict(live=live.value, high_water=hw.value, capacity=cap.value)

    def remove_frame(self, fid):
        self._check(lib().uzl_match_remove_frame(self._h, C.c_int32(fid)))

    def frame_count(self):
        return lib().uzl_match_frame_count(self._h)

    @staticmethod
    def _jobs(pairs, job_ids):
        """pairs: list of (from_frame_ids, to_frame_ids) (ints or lists)."""
        n = len(pairs)
        jobs = np.zeros(n, PAIR_JOB_DTYPE)
        ids = []
        for j, (fr, to) in enumerate(pairs):
            fr = [fr] if np.isscalar(fr) else list(fr)
            to = [to] if np.isscalar(to) else list(to)
            jobs[j]["job_id"] = job_ids[j] if job_ids is not None else j
            jobs[j]["from_begin"] = len(ids); jobs[j]["from_count"] = len(fr); ids += fr
            jobs[j]["to_begin"] = len(ids); jobs[j]["to_count"] = len(to); ids += to
        return jobs, np.asarray(ids, np.int32)

    def estimate(self, pairs, job_ids=None, max_corr=0):
        """Batched estimateEdgeImpl. Returns (results structured array, diag dict or None)."""
        jobs, ids = self._jobs(pairs, job_ids)
        n = len(pairs)
        res = np.zeros(n, EDGE_RESULT_DTYPE)
        diag = None
        cq = ct = cd = mk = None
        if max_corr > 0:
            cq = np.empty((n, max_corr), np.int32); ct = np.empty((n, max_corr), np.int32)
            cd = np.empty((n, max_corr), np.int32); mk = np.empty((n, max_corr), np.uint8)
            diag = dict(corr_query=cq, corr_train=ct, corr_dist=cd, mask=mk)
        self._check(lib().uzl_match_estimate(self._h, C.c_int32(n), _p(jobs, C.c_void_p), _p(ids, c_i32p),
                                             C.c_int32(len(ids)), _p(res, C.c_void_p), C.c_int32(max_corr),
                                             _p(cq, c_i32p), _p(ct, c_i32p), _p(cd, c_i32p), _p(mk, c_u8p)))
        return res, diag

    def launch(self, pairs, job_ids=None):
        jobs, ids = self._jobs(pairs, job_ids)
        self._n_launched = len(pairs)
        self._check(lib().uzl_match_launch(self._h, C.c_int32(len(pairs)), _p(jobs, C.c_void_p), _p(ids, c_i32p),
                                           C.c_int32(len(ids)), C.c_int32(0)))

    def launch_raw(self, jobs, ids):
        """Pre-built PAIR_JOB_DTYPE array + frame id array (no Python work inside a timed region)."""
        self._n_launched = len(jobs)
        self._check(lib().uzl_match_launch(self._h, C.c_int32(len(jobs)), _p(jobs, C.c_void_p), _p(ids, c_i32p),
                                           C.c_int32(len(ids)), C.c_int32(0)))

    def collect(self, out=None):
        res = out if out is not None else np.zeros(self._n_launched, EDGE_RESULT_DTYPE)
        self._check(lib().uzl_match_collect(self._h, _p(res, C.c_void_p), None, None, None, None))
        return res

    def knn2(self, frame_from, frame_to, nq):
        out = [np.empty(nq, np.int32) for _ in range(4)]
        self._check(lib().uzl_match_knn2(self._h, C.c_int32(frame_from), C.c_int32(frame_to),
                                         *[_p(o, c_i32p) for o in out]))
        return tuple(out)

    def ransac_points(self, problems, max_error, iterations, break_percentage, do_prosac=True, job_ids=None):
        """problems: list of (P (3,M), Q (3,M)). Returns one dict per problem (T, consensus, mse, iterations_run, mask)."""
        nb = len(problems)
        offs = np.zeros(nb + 1, np.int32)
        for b, (P, _) in enumerate(problems):
            offs[b + 1] = offs[b] + np.asarray(P).shape[1]
        tot = int(offs[-1])
        Pc = np.empty((max(tot, 1), 3)); Qc = np.empty((max(tot, 1), 3))
        for b, (P, Q) in enumerate(problems):
            Pc[offs[b]:offs[b + 1]] = np.asarray(P).T; Qc[offs[b]:offs[b + 1]] = np.asarray(Q).T
        T = np.empty((nb, 12)); cons = np.empty(nb, np.int32); mse = np.empty(nb); itr = np.empty(nb, np.int32)
        mask = np.zeros(max(tot, 1), np.uint8)
        jid = np.asarray(job_ids if job_ids is not None else np.arange(nb), np.uint64)
        self._check(lib().uzl_ransac_points(self._h, C.c_int32(nb), _p(offs, c_i32p), _p(Pc, c_f64p), _p(Qc, c_f64p),
                                            C.c_double(max_error), C.c_int32(iterations), C.c_double(break_percentage),
                                            C.c_int32(1 if do_prosac else 0), _p(jid, c_u64p), _p(T, c_f64p),
                                            _p(cons, c_i32p), _p(mse, c_f64p), _p(itr, c_i32p), _p(mask, c_u8p)))
        return [dict(T=T[b].reshape(3, 4), consensus=int(cons[b]), mse=float(mse[b]), iterations_run=int(itr[b]),
                     mask=mask[offs[b]:offs[b + 1]].copy()) for b in range(nb)]

    def set_profiling(self, on):
        self._check(lib().uzl_match_set_profiling(self._h, C.c_int32(1 if on else 0)))

    def kernel_times(self):
        cap = 32
        names = (C.c_char_p * cap)(); ms = (C.c_double * cap)(); ln = (C.c_int32 * cap)()
        n = lib().uzl_match_kernel_times(self._h, C.c_int32(cap), names, ms, ln)
        return {names[i].decode(): dict(ms=ms[i], launches=ln[i]) for i in range(max(n, 0))}


# --------------------------------------------------------------------------------------- optimizer
class Pgo:
    """Thin object wrapper over the uzl_pgo_* C ABI."""

    def __init__(self, **cfg):
        L = lib()
        c = PgoCfg()
        L.uzl_pgo_cfg_default(C.byref(c))
        for k, v in cfg.items():
            setattr(c, k, v)
        self.cfg = c
        self._h = C.c_void_p()
        rc = L.uzl_pgo_create(C.byref(c), C.byref(self._h))
        if rc != UZL_OK:
            raise UzlError(rc, L.uzl_status_string(rc).decode())
        self.n = 0
        self.e_in = 0

    def close(self):
        if getattr(self, "_h", None):
            lib().uzl_pgo_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, allow=()):
        if rc != UZL_OK and rc not in allow:
            raise UzlError(rc, lib().uzl_pgo_last_error(self._h).decode())
        return rc

    def set_config(self, **cfg):
        for k, v in cfg.items():
            setattr(self.cfg, k, v)
        self._check(lib().uzl_pgo_set_config(self._h, C.byref(self.cfg)))

    def add_graph(self, nodes_pose, nodes_fixed, edges, sensors=None):
        """Reference-shaped input (SlamNode / SlamEdge arrays, see synth.make_pose_graph)."""
        n = len(nodes_fixed); ne = len(edges["from"])
        na = np.zeros(max(n, 1), NODE_DTYPE)
        na["pose"][:n] = np.asarray(nodes_pose, np.float64).reshape(n, 12); na["fixed"][:n] = nodes_fixed
        ea = np.zeros(max(ne, 1), EDGE_DTYPE)
        for k in ("from", "to", "type", "sensor_from", "sensor_to", "valid"):
            ea[k][:ne] = edges[k]
        for k, w in (("transform", 12), ("displacement_from", 12), ("displacement_to", 12), ("information", 36)):
            ea[k][:ne] = np.asarray(edges[k], np.float64).reshape(ne, w)
        if "diff_time" in edges:
            ea["diff_time"][:ne] = edges["diff_time"]
        S = np.ascontiguousarray(sensors, np.float64).reshape(-1, 12) if sensors is not None and len(sensors) else None
        self._check(lib().uzl_pgo_add_graph(self._h, C.c_int32(n), _p(na, C.c_void_p), C.c_int32(ne), _p(ea, C.c_void_p),
                                            C.c_int32(0 if S is None else S.shape[0]), _p(S, c_f64p)))
        self.n = n; self.e_in = ne

    @staticmethod
    def pack_edges(edges, lo=0, hi=None):
        """SlamEdge dict-of-arrays (synth.make_pose_graph) -> EDGE_DTYPE array of edges [lo, hi)."""
        hi = len(edges["from"]) if hi is None else hi
        ne = hi - lo
        ea = np.zeros(max(ne, 1), EDGE_DTYPE)
        for k in ("from", "to", "type", "sensor_from", "sensor_to", "valid"):
            ea[k][:ne] = edges[k][lo:hi]
        for k, w in (("transform", 12), ("displacement_from", 12), ("displacement_to", 12), ("information", 36)):
            ea[k][:ne] = np.asarray(edges[k][lo:hi], np.float64).reshape(ne, w)
        if "diff_time" in edges:
            ea["diff_time"][:ne] = edges["diff_time"][lo:hi]
        return ea, ne

    def append_graph(self, new_pose, new_fixed, new_edges, flag_index=None, flag_valid=None):
        """uzl_pgo_append_graph: the resident graph grown by new nodes / edges (`new_edges`: dict like add_graph's, or a packed EDGE_DTYPE
        array), the valid flag of old input edges flag_index set to flag_valid.  Old nodes keep the handle's current estimates."""
        n = len(new_fixed)
        na = np.zeros(max(n, 1), NODE_DTYPE)
        na["pose"][:n] = np.asarray(new_pose, np.float64).reshape(n, 12); na["fixed"][:n] = new_fixed
        if isinstance(new_edges, np.ndarray):
            ea, ne = np.ascontiguousarray(new_edges), len(new_edges)
            if ne == 0:
                ea = np.zeros(1, EDGE_DTYPE)
        else:
            ea, ne = self.pack_edges(new_edges)
        fi = np.ascontiguousarray(flag_index if flag_index is not None else [], np.int32)
        fv = np.ascontiguousarray(flag_valid if flag_valid is not None else [], np.uint8)
        assert fi.shape == fv.shape
        self._check(lib().uzl_pgo_append_graph(self._h, C.c_int32(n), _p(na, C.c_void_p), C.c_int32(ne), _p(ea, C.c_void_p),
                                               C.c_int32(len(fi)), _p(fi, c_i32p), _p(fv, c_u8p)))
        self.n += n; self.e_in += ne

    def set_graph(self, poses, fixed, ij, meas, info, robust):
        P = np.ascontiguousarray(poses, np.float64).reshape(-1, 12); f = np.ascontiguousarray(fixed, np.uint8)
        ijc = np.ascontiguousarray(ij, np.int32).reshape(-1, 2)
        Z = np.ascontiguousarray(meas, np.float64).reshape(-1, 12)
        Om = np.ascontiguousarray(info, np.float64).reshape(-1, 36); rb = np.ascontiguousarray(robust, np.uint8)
        self._check(lib().uzl_pgo_set_graph(self._h, C.c_int32(P.shape[0]), _p(P, c_f64p), _p(f, c_u8p),
                                            C.c_int32(ijc.shape[0]), _p(ijc, c_i32p), _p(Z, c_f64p), _p(Om, c_f64p),
                                            _p(rb, c_u8p)))
        self.n = P.shape[0]; self.e_in = ijc.shape[0]

    def reset(self):
        self._check(lib().uzl_pgo_reset(self._h))

    def set_shard(self, rank, world, allreduce=None):
        """Sharded single-graph solve (BASELINE config 4).  allreduce(dev_ptr:int, count:int, stream:int) -> int must sum
        `count` doubles at dev_ptr in place over all ranks (0 = ok).  See uzliti_slam_amd/sharded.py for the RCCL one."""
        if allreduce is None:
            self._shard_cb = ALLREDUCE_FN()
        else:
            def _cb(ptr, count, stream, user):
                try:
                    return int(allreduce(int(ptr or 0), int(count), int(stream or 0)))
                except Exception:      # never let an exception cross the C ABI
                    import traceback
                    traceback.print_exc()
                    return -1
            self._shard_cb = ALLREDUCE_FN(_cb)
        self._check(lib().uzl_pgo_set_shard(self._h, C.c_int32(rank), C.c_int32(world), self._shard_cb, None))

    def set_shard_rccl(self, rank, world, unique_id):
        """Native exchange: the handle owns the RCCL communicator (collective call: every rank, same id from rccl_unique_id())."""
        buf = (C.c_char * RCCL_UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self._check(lib().uzl_pgo_set_shard_rccl(self._h, C.c_int32(rank), C.c_int32(world), buf, C.c_int32(RCCL_UNIQUE_ID_BYTES)))

    def rccl_ranks(self):
        """ncclCommCount of the handle's communicator (0: none)."""
        return int(lib().uzl_pgo_rccl_ranks(self._h))

    def optimize(self, iterations=0):
        st = PgoStats()
        rc = self._check(lib().uzl_pgo_optimize(self._h, C.c_int32(iterations), C.byref(st)),
                         allow=(UZL_ERR_NOT_CONVERGED,))
        d = st.as_dict(); d["status"] = rc
        return d

    def store(self):
        poses = np.empty((self.n, 12)); err = np.empty(max(self.e_in, 1)); used = np.empty(max(self.e_in, 1), np.uint8)
        self._check(lib().uzl_pgo_store(self._h, _p(poses, c_f64p), _p(err, c_f64p), _p(used, c_u8p)))
        return poses, err[:self.e_in], used[:self.e_in]

    def get_fixed(self):
        f = np.empty(self.n, np.uint8)
        self._check(lib().uzl_pgo_get_fixed(self._h, _p(f, c_u8p)))
        return f

    def set_profiling(self, on):
        self._check(lib().uzl_pgo_set_profiling(self._h, C.c_int32(1 if on else 0)))

    def kernel_times(self):
        cap = 64
        names = (C.c_char_p * cap)(); ms = (C.c_double * cap)(); ln = (C.c_int32 * cap)()
        n = lib().uzl_pgo_kernel_times(self._h, C.c_int32(cap), names, ms, ln)
        return {names[i].decode(): dict(ms=ms[i], launches=ln[i]) for i in range(max(n, 0))}


class _BorrowedPgo(Pgo):
    """A Pgo over a handle the batch owns: neither close() nor the finaliser may destroy it."""

    def close(self):
        self._h = None

    __del__ = close


class PgoBatch:
    """uzl_pgo_batch_*: n independent graphs solved through shared launches.  `graphs[i]` is an ordinary Pgo over handle i
    (add_graph / set_graph / reset / store); optimize() solves them all and returns one stats dict per graph."""

    def __init__(self, n_graphs, **cfg):
        L = lib()
        c = PgoCfg()
        L.uzl_pgo_cfg_default(C.byref(c))
        for k, v in cfg.items():
            setattr(c, k, v)
        self.cfg = c
        self._b = C.c_void_p()
        L.uzl_pgo_batch_graph.restype = C.c_void_p
        L.uzl_pgo_batch_graph.argtypes = [C.c_void_p, C.c_int32]
        L.uzl_pgo_batch_last_error.restype = C.c_char_p
        L.uzl_pgo_batch_last_error.argtypes = [C.c_void_p]
        L.uzl_pgo_batch_destroy.restype = None
        L.uzl_pgo_batch_destroy.argtypes = [C.c_void_p]
        rc = L.uzl_pgo_batch_create(C.byref(c), C.c_int32(n_graphs), C.byref(self._b))
        if rc != UZL_OK:
            raise UzlError(rc, L.uzl_status_string(rc).decode())
        self.graphs = []
        for i in range(n_graphs):
            p = _BorrowedPgo.__new__(_BorrowedPgo)
            p.cfg = c; p._h = C.c_void_p(L.uzl_pgo_batch_graph(self._b, i)); p.n = 0; p.e_in = 0
            self.graphs.append(p)
        self.n_batched = 0

    def optimize(self, iterations=0):
        n = len(self.graphs)
        st = (PgoStats * n)(); nb = C.c_int32()
        rc = lib().uzl_pgo_batch_optimize(self._b, C.c_int32(iterations), st, C.byref(nb))
        if rc not in (UZL_OK, UZL_ERR_NOT_CONVERGED):
            raise UzlError(rc, lib().uzl_pgo_batch_last_error(self._b).decode())
        self.n_batched = nb.value
        out = []
        for i in range(n):
            d = st[i].as_dict(); d["status"] = rc
            out.append(d)
        return out

    def set_resident(self, n):
        """graphs solved at a time (0 = all); the rest of the batch waits in a queue and takes the slots of finished graphs"""
        rc = lib().uzl_pgo_batch_set_resident(self._b, C.c_int32(n))
        if rc != UZL_OK:
            raise UzlError(rc, "uzl_pgo_batch_set_resident")

    def set_profiling(self, on):
        lib().uzl_pgo_batch_set_profiling(self._b, C.c_int32(1 if on else 0))

    def kernel_times(self):
        cap = 16
        names = (C.c_char_p * cap)(); ms = (C.c_double * cap)(); ln = (C.c_int32 * cap)()
        n = lib().uzl_pgo_batch_kernel_times(self._b, C.c_int32(cap), names, ms, ln)
        return {names[i].decode(): dict(ms=ms[i], launches=ln[i]) for i in range(max(n, 0))}

    def close(self):
        if getattr(self, "_b", None):
            for p in self.graphs:
                p._h = None
            lib().uzl_pgo_batch_destroy(self._b)
            self._b = None

    __del__ = close


# --------------------------------------------------------------------------------------- edge filter
class Filter:
    """uzl_filter_* (TransformationFilter / EdgeCluster, transformation_filter.cpp:43-350)."""

    def __init__(self, **cfg):
        L = lib()
        c = FilterCfg()
        L.uzl_filter_cfg_default(C.byref(c))
        for k, v in cfg.items():
            setattr(c, k, v)
        self.cfg = c
        self._h = C.c_void_p()
        rc = L.uzl_filter_create(C.byref(c), C.byref(self._h))
        if rc != UZL_OK:
            raise UzlError(rc, L.uzl_status_string(rc).decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().uzl_filter_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise UzlError(rc, lib().uzl_filter_last_error(self._h).decode())
        return rc

    def set_sensors(self, sensors):
        s = np.ascontiguousarray(sensors, np.float64).reshape(-1, 12)
        self._check(lib().uzl_filter_set_sensors(self._h, C.c_int32(len(s)), _p(s, c_f64p)))

    def add(self, edges):
        arr, keep = pack_filter_edges(edges)
        self._check(lib().uzl_filter_add(self._h, C.c_int32(len(edges)), arr))

    def add_packed(self, arr):
        """arr: FILTER_EDGE_DTYPE array; the stamp arrays its pointer fields address must stay alive for the call."""
        a = np.ascontiguousarray(arr, FILTER_EDGE_DTYPE)
        if len(a):
            self._check(lib().uzl_filter_add(self._h, C.c_int32(len(a)), _p(a, C.c_void_p)))

    def remove(self, keys):
        k = np.ascontiguousarray(keys, np.uint64)
        self._check(lib().uzl_filter_remove(self._h, C.c_int32(len(k)), _p(k, c_u64p)))

    def _keys(self, fn):
        n = C.c_int32()
        self._check(fn(self._h, C.c_int32(0), None, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint64)
        self._check(fn(self._h, C.c_int32(len(out)), _p(out, c_u64p), C.byref(n)))
        return out[:n.value]

    def all_edges(self):
        return self._keys(lib().uzl_filter_all_edges)

    def valid_edges(self):
        return self._keys(lib().uzl_filter_valid_edges)

    def calc_valid_edges(self):
        n = C.c_int32()
        self._check(lib().uzl_filter_calc_valid_edges(self._h, C.byref(n)))
        return n.value

    def clusters(self, with_eval=False):
        """clusters_ in order: list of dicts (info + keys + valid [+ P, Q, T, ransac_consensus of the last evaluation])."""
        out = []
        for i in range(self._check(lib().uzl_filter_cluster_count(self._h))):
            ci = ClusterInfo()
            self._check(lib().uzl_filter_cluster_info(self._h, C.c_int32(i), C.byref(ci)))
            d = ci.as_dict()
            keys = np.zeros(max(ci.size, 1), np.uint64); valid = np.zeros(max(ci.size, 1), np.uint8)
            self._check(lib().uzl_filter_cluster_edges(self._h, C.c_int32(i), C.c_int32(len(keys)), _p(keys, c_u64p), _p(valid, c_u8p)))
            d["keys"] = keys[:ci.size]; d["valid"] = valid[:ci.size]
            if with_eval:
                cap = max(ci.size + 128, 256)
                P = np.zeros((cap, 3)); Q = np.zeros((cap, 3)); T = np.zeros(12); rc = C.c_int32()
                m = self._check(lib().uzl_filter_cluster_last_eval(self._h, C.c_int32(i), C.c_int32(cap), _p(P, c_f64p), _p(Q, c_f64p),
                                                                   _p(T, c_f64p), C.byref(rc)))
                d.update(P=P[:m].copy(), Q=Q[:m].copy(), T=T, ransac_consensus=rc.value)
            out.append(d)
        return out


# --------------------------------------------------------------------------------------- edge acceptance gate
class Gate:
    """uzl_gate_* (GraphSlamNode::newEdgeCallback / checkEdgeHeuristic / SlamGraph::astar)."""

    def __init__(self, **cfg):
        L = lib()
        c = GateCfg()
        L.uzl_gate_cfg_default(C.byref(c))
        for k, v in cfg.items():
            setattr(c, k, v)
        self.cfg = c
        self._h = C.c_void_p()
        rc = L.uzl_gate_create(C.byref(c), C.byref(self._h))
        if rc != UZL_OK:
            raise UzlError(rc, L.uzl_status_string(rc).decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().uzl_gate_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise UzlError(rc, lib().uzl_gate_last_error(self._h).decode())
        return rc

    def set_graph(self, poses, edges, merged=None):
        """poses (n,12); edges: GATE_EDGE_DTYPE array (from, to, type, valid); merged (n) u8 or None."""
        P = np.ascontiguousarray(poses, np.float64).reshape(-1, 12)
        E = np.ascontiguousarray(edges, GATE_EDGE_DTYPE)
        m = None if merged is None else np.ascontiguousarray(merged, np.uint8)
        self._check(lib().uzl_gate_set_graph(self._h, C.c_int32(len(P)), _p(P, c_f64p), _p(m, c_u8p), C.c_int32(len(E)),
                                             _p(E, C.c_void_p) if len(E) else None))

    def check(self, cand, want_dist=True):
        """-> (accept u8, valid u8, astar_dist f64 or None) per candidate, in order.  want_dist=False passes astar_dist = NULL: the verdicts
        are the same, and the searches whose verdict the straight-line distance between the nodes already decides are not run."""
        Cn = np.ascontiguousarray(cand, GATE_EDGE_DTYPE)
        n = len(Cn)
        acc = np.zeros(max(n, 1), np.uint8); val = np.zeros(max(n, 1), np.uint8); dist = np.zeros(max(n, 1)) if want_dist else None
        self._check(lib().uzl_gate_check(self._h, C.c_int32(n), _p(Cn, C.c_void_p) if n else None, _p(acc, c_u8p), _p(val, c_u8p),
                                         _p(dist, c_f64p) if want_dist else None))
        return acc[:n], val[:n], (dist[:n] if want_dist else None)

    def edge_count(self):
        return self._check(lib().uzl_gate_edge_count(self._h))


def schur_plan(row_ptr, col, cap=24):
    """uzl_pgo_schur_plan (host only) -> dict(red_row, run_id, run_pos, row_ptr, col, n_reduced, n_runs)."""
    rp = np.ascontiguousarray(row_ptr, np.int32); cl = np.ascontiguousarray(col, np.int32)
    nb = len(rp) - 1
    red_row = np.empty(max(nb, 1), np.int32); run_id = np.empty(max(nb, 1), np.int32); run_pos = np.empty(max(nb, 1), np.int32)
    rrp = np.zeros(nb + 1, np.int32); cap_slots = len(cl) + 2 * nb + 2; rcol = np.empty(cap_slots, np.int32)
    nr = C.c_int32(); nruns = C.c_int32()
    i32 = C.POINTER(C.c_int32)
    rc = lib().uzl_pgo_schur_plan(C.c_int32(nb), rp.ctypes.data_as(i32), cl.ctypes.data_as(i32), C.c_int32(cap), red_row.ctypes.data_as(i32),
                                  run_id.ctypes.data_as(i32), run_pos.ctypes.data_as(i32), rrp.ctypes.data_as(i32), rcol.ctypes.data_as(i32),
                                  C.c_int32(cap_slots), C.byref(nr), C.byref(nruns))
    if rc != UZL_OK:
        raise UzlError(rc, lib().uzl_status_string(rc).decode())
    return dict(red_row=red_row[:nb], run_id=run_id[:nb], run_pos=run_pos[:nb], row_ptr=rrp[:nr.value + 1], col=rcol[:rrp[nr.value]],
                n_reduced=nr.value, n_runs=nruns.value)


def schur_plan_strong(row_ptr, col, slot_w, cap=24, strong_min=1, theta=0.25, one_level_max=0):
    """uzl_pgo_schur_plan_strong (host only) -> dict(red_row, sep_rows, n_reduced, n_sep, n_groups, n_blocks)."""
    rp = np.ascontiguousarray(row_ptr, np.int32); cl = np.ascontiguousarray(col, np.int32); w = np.ascontiguousarray(slot_w, np.float64)
    nb = len(rp) - 1
    assert len(w) == len(cl)
    red_row = np.empty(max(nb, 1), np.int32); cap_rows = 32 * nb + 32; sep = np.empty(cap_rows, np.int32); counts = np.zeros(5, np.int32)
    i32 = C.POINTER(C.c_int32)
    rc = lib().uzl_pgo_schur_plan_strong(C.c_int32(nb), rp.ctypes.data_as(i32), cl.ctypes.data_as(i32), C.c_int32(cap), w.ctypes.data_as(c_f64p),
                                         C.c_int32(strong_min), C.c_double(theta), C.c_int32(one_level_max), red_row.ctypes.data_as(i32), sep.ctypes.data_as(i32),
                                         C.c_int32(cap_rows), counts.ctypes.data_as(i32))
    if rc != UZL_OK:
        raise UzlError(rc, lib().uzl_status_string(rc).decode())
    return dict(red_row=red_row[:nb], sep_rows=sep[:counts[0]].copy(), n_reduced=int(counts[0]), n_sep=int(counts[1]), n_groups=int(counts[2]),
                n_blocks=int(counts[3]), contiguous=counts[4] / 1000.)


# --------------------------------------------------------------------------------------- distance loop-closure candidates
class Radius:
    """uzl_radius_* (SlamGraph::getNodesWithinRadius + the caller's filters, graph_slam_node.cpp:272-289)."""

    def __init__(self, **cfg):
        L = lib()
        c = RadiusCfg()
        L.uzl_radius_cfg_default(C.byref(c))
        for k, v in cfg.items():
            setattr(c, k, v)
        self.cfg = c
        self._h = C.c_void_p()
        rc = L.uzl_radius_create(C.byref(c), C.byref(self._h))
        if rc != UZL_OK:
            raise UzlError(rc, L.uzl_status_string(rc).decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().uzl_radius_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise UzlError(rc, lib().uzl_radius_last_error(self._h).decode())
        return rc

    def set_nodes(self, poses, stamps_front_ns):
        P = np.ascontiguousarray(poses, np.float64).reshape(-1, 12); st = np.ascontiguousarray(stamps_front_ns, np.int64)
        self.n = len(P)
        self._check(lib().uzl_radius_set_nodes(self._h, C.c_int32(len(P)), _p(P, c_f64p), st.ctypes.data_as(C.POINTER(C.c_int64))))

    def query(self, queries, cap=None):
        q = np.ascontiguousarray(queries, np.int32)
        cap = int(cap if cap is not None else max(1, self.n * max(len(q), 1)))
        f = np.zeros(max(cap, 1), np.int32); t = np.zeros(max(cap, 1), np.int32); cnt = np.zeros(max(len(q), 1), np.int32)
        tot = C.c_int64()
        self._check(lib().uzl_radius_query(self._h, C.c_int32(len(q)), _p(q, c_i32p), C.c_int64(cap), _p(f, c_i32p), _p(t, c_i32p),
                                           _p(cnt, c_i32p), C.byref(tot)))
        w = min(tot.value, cap)
        return f[:w].copy(), t[:w].copy(), cnt[:len(q)].copy(), tot.value


# --------------------------------------------------------------------------------------- appearance-based candidates
class Places:
    """uzl_places_* (FastLshSet / LshSetRecognizer / PlaceRecognizer, place_recognition/src)."""

    def __init__(self, **cfg):
        L = lib()
        c = PlacesCfg()
        L.uzl_places_cfg_default(C.byref(c))
        for k, v in cfg.items():
            setattr(c, k, v)
        self.cfg = c
        self._h = C.c_void_p()
        rc = L.uzl_places_create(C.byref(c), C.byref(self._h))
        if rc != UZL_OK:
            raise UzlError(rc, L.uzl_status_string(rc).decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().uzl_places_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc < 0:
            raise UzlError(rc, lib().uzl_places_last_error(self._h).decode())
        return rc

    @staticmethod
    def _d(desc):
        d = np.ascontiguousarray(desc, np.uint8)
        return d, (d.shape[0] if d.ndim == 2 else 0), (d.shape[1] if d.ndim == 2 else 32)

    def search_and_add(self, desc, stamp_ns, cap=64):
        d, rows, nb = self._d(desc)
        out = np.zeros(max(cap, 1), np.int32); n = C.c_int32(); idx = C.c_int32()
        self._check(lib().uzl_places_search_and_add(self._h, _p(d, c_u8p) if rows else None, C.c_int32(rows), C.c_int32(nb), C.c_int64(int(stamp_ns)),
                                                    C.c_int32(cap), _p(out, c_i32p), C.byref(n), C.byref(idx)))
        return out[:min(n.value, cap)].copy(), idx.value

    def add(self, desc, stamp_ns):
        d, rows, nb = self._d(desc)
        idx = C.c_int32()
        self._check(lib().uzl_places_add(self._h, _p(d, c_u8p) if rows else None, C.c_int32(rows), C.c_int32(nb), C.c_int64(int(stamp_ns)), C.byref(idx)))
        return idx.value

    def search(self, desc, stamp_ns, query_place=-1, cap=64):
        d, rows, nb = self._d(desc)
        out = np.zeros(max(cap, 1), np.int32); n = C.c_int32()
        self._check(lib().uzl_places_search(self._h, _p(d, c_u8p) if rows else None, C.c_int32(rows), C.c_int32(nb), C.c_int64(int(stamp_ns)),
                                            C.c_int32(query_place), C.c_int32(cap), _p(out, c_i32p), C.byref(n)))
        return out[:min(n.value, cap)].copy()

    def remove(self, place, desc):
        d, rows, nb = self._d(desc)
        self._check(lib().uzl_places_remove(self._h, C.c_int32(place), _p(d, c_u8p) if rows else None, C.c_int32(rows), C.c_int32(nb)))

    def count(self):
        return self._check(lib().uzl_places_count(self._h))

    def last_counts(self):
        n = self._check(lib().uzl_places_last_counts(self._h, C.c_int32(0), None))
        out = np.zeros(max(n, 1), np.int32)
        self._check(lib().uzl_places_last_counts(self._h, C.c_int32(len(out)), _p(out, c_i32p)))
        return out[:n]
