"""CPU ORACLE binding (test infrastructure, NOT product code).

ctypes wrapper over oracle/libuzl_oracle.so (plain-C restatement of the reference's CPU path,
see oracle/uzl_oracle.h).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package.  PARITY UNPINNED: the reference ships no golden vectors for this path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libuzl_oracle.so")
_lib = None

c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("uzl_oracle_match.c", "uzl_oracle_pgo.c", "uzl_oracle_filter.c", "uzl_oracle_gate.c", "uzl_oracle_radius.c", "uzl_oracle_places.c", "uzl_oracle.h")]
    if not force and os.path.exists(_LIB_PATH) and all(
            os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


class Frame(C.Structure):
    _fields_ = [("desc", c_u8p), ("n", C.c_int32), ("bytes_per_desc", C.c_int32),
                ("pos_xyz", c_f64p), ("valid3d", c_u8p),
                ("feature_type", C.c_int32), ("sensor_frame", C.c_int32)]


class EdgeResult(C.Structure):
    _fields_ = [("ok", C.c_int32), ("consensus", C.c_int32), ("n_matches", C.c_int32),
                ("n_corr", C.c_int32), ("frame_from", C.c_int32), ("frame_to", C.c_int32),
                ("iterations_run", C.c_int32), ("best_iteration", C.c_int32),
                ("mse", C.c_double), ("T", C.c_double * 12), ("information", C.c_double * 36)]


class Node(C.Structure):
    _fields_ = [("pose", C.c_double * 12), ("fixed", C.c_int32)]


class Edge(C.Structure):
    _fields_ = [("from_", C.c_int32), ("to", C.c_int32), ("type", C.c_int32),
                ("sensor_from", C.c_int32), ("sensor_to", C.c_int32), ("valid", C.c_int32),
                ("transform", C.c_double * 12), ("displacement_from", C.c_double * 12),
                ("displacement_to", C.c_double * 12), ("information", C.c_double * 36), ("diff_time", C.c_double)]


class PgoStats(C.Structure):
    _fields_ = [("iterations_done", C.c_int32), ("lm_trials", C.c_int32),
                ("terminated_early", C.c_int32), ("n_vertices", C.c_int32), ("n_edges", C.c_int32),
                ("n_gauge_fixed", C.c_int32),
                ("chi2_initial", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
                ("t_order_ms", C.c_double), ("t_symbolic_ms", C.c_double), ("t_numeric_ms", C.c_double),
                ("t_linearize_ms", C.c_double), ("t_total_ms", C.c_double),
                ("factor_blocks", C.c_int64)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.uzlo_chi2.restype = C.c_double
        _lib.uzlo_filter_sort.restype = C.c_int32
        _lib.uzlo_consensus3d.restype = C.c_int32
        _lib.uzlo_prosac_prefix.restype = C.c_int32
        _lib.uzlo_flatten_graph.restype = C.c_int32
        _lib.uzlo_set_fixed_nodes.restype = C.c_int32
        _lib.uzlo_pgo_optimize.restype = C.c_int32
    return _lib


_native = None


def native_lib():
    """The same C sources built on THIS machine with -O3 -march=native -fopenmp (oracle/Makefile target `native`): the CPU
    baseline of bench.py.  Never used as a checker (the portable build is), never shipped (oracle/_native/ is ignored)."""
    global _native
    if _native is None:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native"])
        _native = C.CDLL(os.path.join(_HERE, "_native", "libuzl_oracle_native.so"))
        _native.uzlo_pgo_optimize.restype = C.c_int32
        _native.uzlo_has_openmp.restype = C.c_int32
    return _native


def _p(a, t):
    return a.ctypes.data_as(t)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ------------------------------------------------------------------ matching half
def knn2(query, train):
    """M1. query (nq, bytes) u8, train (nt, bytes) u8 -> idx0, d0, idx1, d1 (int32)."""
    query = _u8(query); train = _u8(train)
    nq = query.shape[0]; nt = train.shape[0]
    nbytes = query.shape[1] if query.ndim == 2 else train.shape[1]
    out = [np.empty(nq, np.int32) for _ in range(4)]
    lib().uzlo_knn2(_p(query, c_u8p), C.c_int32(nq), _p(train, c_u8p), C.c_int32(nt), C.c_int32(nbytes),
                    *[_p(o, c_i32p) for o in out])
    return tuple(out)


def filter_sort(idx0, d0, idx1, d1, valid_train, valid_query):
    """M2+M4 -> (query_idx, train_idx, dist, n_ratio)."""
    nq = len(idx0)
    oq = np.empty(nq, np.int32); ot = np.empty(nq, np.int32); od = np.empty(nq, np.int32)
    nr = C.c_int32(0)
    vt = _u8(valid_train); vq = _u8(valid_query)
    m = lib().uzlo_filter_sort(C.c_int32(nq), _p(_i32(idx0), c_i32p), _p(_i32(d0), c_i32p),
                               _p(_i32(idx1), c_i32p), _p(_i32(d1), c_i32p), _p(vt, c_u8p), _p(vq, c_u8p),
                               _p(oq, c_i32p), _p(ot, c_i32p), _p(od, c_i32p), C.byref(nr))
    return oq[:m].copy(), ot[:m].copy(), od[:m].copy(), nr.value


def sample3(seed, job_id, it, iterations, m, do_prosac=True):
    out = (C.c_int32 * 3)()
    lib().uzlo_sample3(C.c_uint64(seed), C.c_uint64(job_id), C.c_int32(it), C.c_int32(iterations),
                       C.c_int32(m), C.c_int32(1 if do_prosac else 0), out)
    return [out[0], out[1], out[2]]


def prosac_prefix(it, iterations, m):
    return lib().uzlo_prosac_prefix(C.c_int32(it), C.c_int32(iterations), C.c_int32(m))


def svd3f(A):
    A = np.ascontiguousarray(A, np.float32).reshape(9)
    U = np.empty(9, np.float32); S = np.empty(3, np.float32); V = np.empty(9, np.float32)
    f32p = C.POINTER(C.c_float)
    lib().uzlo_svd3f(_p(A, f32p), _p(U, f32p), _p(S, f32p), _p(V, f32p))
    return U.reshape(3, 3), S, V.reshape(3, 3)


def pose_svd(P, Q, idx=None):
    """M7. P, Q: (3, M) arrays (column i = point i). Returns T (3,4) with Q ~= T*P."""
    Pc = _f64(np.asarray(P).T); Qc = _f64(np.asarray(Q).T)   # (M,3) row-major == 3xM column-major
    T = np.empty(12, np.float64)
    if idx is None:
        lib().uzlo_pose_svd(_p(Pc, c_f64p), _p(Qc, c_f64p), None, C.c_int32(Pc.shape[0]), _p(T, c_f64p))
    else:
        ii = _i32(idx)
        lib().uzlo_pose_svd(_p(Pc, c_f64p), _p(Qc, c_f64p), _p(ii, c_i32p), C.c_int32(len(ii)), _p(T, c_f64p))
    return T.reshape(3, 4)


def consensus3d(P, Q, T, thresh):
    Pc = _f64(np.asarray(P).T); Qc = _f64(np.asarray(Q).T)
    m = Pc.shape[0]
    s = np.zeros(m, np.uint8)
    Tc = _f64(np.asarray(T).reshape(12))
    c = lib().uzlo_consensus3d(_p(Pc, c_f64p), _p(Qc, c_f64p), C.c_int32(m), _p(Tc, c_f64p),
                               C.c_double(thresh), _p(s, c_u8p))
    return c, s


def prosac(P, Q, max_error, iterations, break_percentage, do_prosac=True, seed=0, job_id=0):
    """M6. P,Q (3,M). -> dict(T (3,4), consensus, mse, mask, iterations_run, best_iteration)."""
    Pc = _f64(np.asarray(P).T); Qc = _f64(np.asarray(Q).T)
    m = Pc.shape[0]
    T = np.empty(12, np.float64); mask = np.zeros(max(m, 1), np.uint8)
    cons = C.c_int32(0); mse = C.c_double(0); itr = C.c_int32(0); bi = C.c_int32(0)
    lib().uzlo_prosac(_p(Pc, c_f64p), _p(Qc, c_f64p), C.c_int32(m), C.c_double(max_error),
                      C.c_int32(iterations), C.c_double(break_percentage), C.c_int32(1 if do_prosac else 0),
                      C.c_uint64(seed), C.c_uint64(job_id), _p(T, c_f64p), C.byref(cons), C.byref(mse),
                      _p(mask, c_u8p), C.byref(itr), C.byref(bi))
    return dict(T=T.reshape(3, 4), consensus=cons.value, mse=mse.value, mask=mask[:m].copy(),
                iterations_run=itr.value, best_iteration=bi.value)


def information(consensus, mse):
    out = np.empty(36, np.float64)
    lib().uzlo_information(C.c_int32(consensus), C.c_double(mse), _p(out, c_f64p))
    return out.reshape(6, 6)


def _mk_frames(frames):
    """frames: list of dict(desc (n,bytes) u8, pos (3,n) f64, valid (n) u8, feature_type, sensor_frame)."""
    arr = (Frame * max(len(frames), 1))()
    keep = []
    for i, f in enumerate(frames):
        d = _u8(f["desc"]); p = _f64(np.asarray(f["pos"]).T); v = _u8(f["valid"])
        keep += [d, p, v]
        arr[i].desc = _p(d, c_u8p); arr[i].n = d.shape[0]; arr[i].bytes_per_desc = d.shape[1]
        arr[i].pos_xyz = _p(p, c_f64p); arr[i].valid3d = _p(v, c_u8p)
        arr[i].feature_type = int(f.get("feature_type", 2)); arr[i].sensor_frame = int(f.get("sensor_frame", 0))
    return arr, keep


def estimate_edge(frames_from, frames_to, ransac_threshold=0.2, ransac_iteration=100,
                  break_percentage=0.6, do_prosac=True, seed=0, job_id=0):
    """estimateEdgeDirect for one node pair. Returns dict incl. sorted correspondences and inlier mask."""
    af, k1 = _mk_frames(frames_from); at, k2 = _mk_frames(frames_to)
    max_corr = max([f["desc"].shape[0] for f in frames_to] + [1])
    cq = np.full(max_corr, -1, np.int32); ct = np.full(max_corr, -1, np.int32); cd = np.full(max_corr, -1, np.int32)
    mk = np.zeros(max_corr, np.uint8)
    res = EdgeResult()
    lib().uzlo_estimate_edge(af, C.c_int32(len(frames_from)), at, C.c_int32(len(frames_to)),
                             C.c_double(ransac_threshold), C.c_int32(ransac_iteration), C.c_double(break_percentage),
                             C.c_int32(1 if do_prosac else 0), C.c_uint64(seed), C.c_uint64(job_id),
                             C.byref(res), C.c_int32(max_corr), _p(cq, c_i32p), _p(ct, c_i32p), _p(cd, c_i32p),
                             _p(mk, c_u8p))
    m = res.n_corr
    return dict(ok=res.ok, consensus=res.consensus, n_matches=res.n_matches, n_corr=m,
                frame_from=res.frame_from, frame_to=res.frame_to, iterations_run=res.iterations_run,
                best_iteration=res.best_iteration, mse=res.mse,
                T=np.array(res.T[:]).reshape(3, 4), information=np.array(res.information[:]).reshape(6, 6),
                corr_query=cq[:m].copy(), corr_train=ct[:m].copy(), corr_dist=cd[:m].copy(), mask=mk[:m].copy())


def set_vote_recipe(recipe):
    """0 = fused (default, = the HIP kernels), 1 = the reference build's unfused evaluation order.  Process-global; tests only."""
    lib().uzlo_set_vote_recipe(C.c_int32(int(recipe)))


def vote_recipe_diff(P, Q, max_error, iterations, do_prosac=True, seed=0, job_id=0):
    """(tests, differing verdicts, smallest |distance - threshold|) over every hypothesis of a PROSAC run, both recipes."""
    Pc = _f64(np.asarray(P).T); Qc = _f64(np.asarray(Q).T)
    nt = C.c_int64(); nd = C.c_int64(); mm = C.c_double()
    lib().uzlo_vote_recipe_diff(_p(Pc, c_f64p), _p(Qc, c_f64p), C.c_int32(Pc.shape[0]), C.c_double(max_error), C.c_int32(iterations),
                                C.c_int32(1 if do_prosac else 0), C.c_uint64(seed), C.c_uint64(job_id), C.byref(nt), C.byref(nd), C.byref(mm))
    return nt.value, nd.value, mm.value


class PreparedPairs:
    """Frame structs of a list of (frame_from, frame_to), built once: timing a batch must not time this Python loop."""

    def __init__(self, pairs):
        self.n = len(pairs)
        self.af, self._k1 = _mk_frames([p[0] for p in pairs])
        self.at, self._k2 = _mk_frames([p[1] for p in pairs])


def estimate_edge_batch(pairs, ransac_threshold=0.2, ransac_iteration=100, break_percentage=0.6, do_prosac=True, seed=0,
                        job_id0=0, threads=1, native=True):
    """bench.py's all-core matching baseline: pairs = [(frame_from, frame_to), ...], `threads` estimator threads inside the
    -march=native / OpenMP build.  Returns (ok, consensus) arrays."""
    prep = pairs if isinstance(pairs, PreparedPairs) else PreparedPairs(pairs)
    res = (EdgeResult * prep.n)()
    L = native_lib() if native else lib()
    L.uzlo_estimate_edge_batch(C.c_int32(prep.n), prep.af, prep.at, C.c_double(ransac_threshold), C.c_int32(ransac_iteration),
                               C.c_double(break_percentage), C.c_int32(1 if do_prosac else 0), C.c_uint64(seed), C.c_uint64(job_id0),
                               C.c_int32(threads), res)
    return np.array([r.ok for r in res]), np.array([r.consensus for r in res])


# ------------------------------------------------------------------ pose-graph half
def quat_from_R(R):
    q = np.empty(4); lib().uzlo_quat_from_R(_p(_f64(R).reshape(9), c_f64p), _p(q, c_f64p)); return q


def R_from_quat(q):
    R = np.empty(9); lib().uzlo_R_from_quat(_p(_f64(q), c_f64p), _p(R, c_f64p)); return R.reshape(3, 3)


def to_vector_mqt(T):
    v = np.empty(6); lib().uzlo_to_vector_mqt(_p(_f64(T).reshape(12), c_f64p), _p(v, c_f64p)); return v


def from_vector_mqt(v):
    T = np.empty(12); lib().uzlo_from_vector_mqt(_p(_f64(v), c_f64p), _p(T, c_f64p)); return T.reshape(3, 4)


def to_euler(R):
    v = np.empty(3); lib().uzlo_to_euler(_p(_f64(R).reshape(9), c_f64p), _p(v, c_f64p)); return v


def from_euler(rpy):
    R = np.empty(9); lib().uzlo_from_euler(_p(_f64(rpy), c_f64p), _p(R, c_f64p)); return R.reshape(3, 3)


def edge_error(Xi, Xj, Z):
    e = np.empty(6)
    lib().uzlo_edge_error(_p(_f64(Xi).reshape(12), c_f64p), _p(_f64(Xj).reshape(12), c_f64p),
                          _p(_f64(Z).reshape(12), c_f64p), _p(e, c_f64p))
    return e


def edge_jacobians(Xi, Xj, Z):
    Ji = np.empty(36); Jj = np.empty(36)
    lib().uzlo_edge_jacobians(_p(_f64(Xi).reshape(12), c_f64p), _p(_f64(Xj).reshape(12), c_f64p),
                              _p(_f64(Z).reshape(12), c_f64p), _p(Ji, c_f64p), _p(Jj, c_f64p))
    return Ji.reshape(6, 6), Jj.reshape(6, 6)


def huber(e2, delta=1.0):
    r = np.empty(3); lib().uzlo_huber(C.c_double(e2), C.c_double(delta), _p(r, c_f64p)); return r


def odom_convert(x, y, theta, dt):
    """g2o OdomConvert round trip (motion -> wheel velocities -> motion), wheel base 1."""
    out = np.zeros(3)
    lib().uzlo_odom_convert(C.c_double(x), C.c_double(y), C.c_double(theta), C.c_double(dt), _p(out, c_f64p))
    return out


def flatten_graph(nodes_pose, nodes_fixed, edges, sensors=None, optimize_xy_only=False, use_odometry_parameters=False):
    """G1. edges: dict of arrays (from, to, type, sensor_from, sensor_to, valid, transform (E,12),
    displacement_from (E,12), displacement_to (E,12), information (E,36))."""
    n = len(nodes_fixed); ne = len(edges["from"])
    na = (Node * max(n, 1))()
    P = _f64(nodes_pose).reshape(n, 12)
    for i in range(n):
        na[i].pose[:] = P[i].tolist(); na[i].fixed = int(nodes_fixed[i])
    ea = (Edge * max(ne, 1))()
    for k in range(ne):
        ea[k].from_ = int(edges["from"][k]); ea[k].to = int(edges["to"][k]); ea[k].type = int(edges["type"][k])
        ea[k].sensor_from = int(edges["sensor_from"][k]); ea[k].sensor_to = int(edges["sensor_to"][k])
        ea[k].valid = int(edges["valid"][k])
        ea[k].transform[:] = np.asarray(edges["transform"][k]).reshape(12).tolist()
        ea[k].displacement_from[:] = np.asarray(edges["displacement_from"][k]).reshape(12).tolist()
        ea[k].displacement_to[:] = np.asarray(edges["displacement_to"][k]).reshape(12).tolist()
        ea[k].information[:] = np.asarray(edges["information"][k]).reshape(36).tolist()
        ea[k].diff_time = float(edges["diff_time"][k]) if "diff_time" in edges else 0.0
    S = _f64(sensors).reshape(-1, 12) if sensors is not None and len(sensors) else np.zeros((0, 12))
    poses = np.empty((n, 12)); fixed = np.empty(n, np.uint8); ij = np.empty((max(ne, 1), 2), np.int32)
    meas = np.empty((max(ne, 1), 12)); info = np.empty((max(ne, 1), 36)); robust = np.empty(max(ne, 1), np.uint8)
    src = np.empty(max(ne, 1), np.int32)
    m = lib().uzlo_flatten_graph(C.c_int32(n), na, C.c_int32(ne), ea, C.c_int32(S.shape[0]),
                                 _p(S, c_f64p) if S.size else None, C.c_int32(1 if optimize_xy_only else 0),
                                 C.c_int32(1 if use_odometry_parameters else 0),
                                 _p(poses, c_f64p), _p(fixed, c_u8p), _p(ij, c_i32p), _p(meas, c_f64p),
                                 _p(info, c_f64p), _p(robust, c_u8p), _p(src, c_i32p))
    return dict(poses=poses, fixed=fixed, ij=ij[:m].copy(), meas=meas[:m].copy(), info=info[:m].copy(),
                robust=robust[:m].copy(), src_edge=src[:m].copy())


def set_fixed_nodes(fixed, ij):
    f = _u8(fixed).copy(); ijc = _i32(ij).reshape(-1, 2)
    c = lib().uzlo_set_fixed_nodes(C.c_int32(len(f)), _p(f, c_u8p), C.c_int32(ijc.shape[0]), _p(ijc, c_i32p))
    return f, c


def pgo_optimize(poses, fixed, ij, meas, info, robust, iterations=20, huber_delta=1.0, native_threads=None):
    """G3-G9 on the flattened problem. Returns (poses_out (n,12), stats dict).
    native_threads = k: run the -march=native / OpenMP baseline build with k threads over the edges (bench.py only)."""
    P = _f64(poses).reshape(-1, 12).copy(); f = _u8(fixed); ijc = _i32(ij).reshape(-1, 2)
    Z = _f64(meas).reshape(-1, 12); Om = _f64(info).reshape(-1, 36); rb = _u8(robust)
    st = PgoStats()
    L = lib()
    if native_threads is not None:
        L = native_lib()
        L.uzlo_set_threads(C.c_int32(int(native_threads)))
    L.uzlo_pgo_optimize(C.c_int32(P.shape[0]), _p(P, c_f64p), _p(f, c_u8p), C.c_int32(ijc.shape[0]),
                            _p(ijc, c_i32p), _p(Z, c_f64p), _p(Om, c_f64p), _p(rb, c_u8p),
                            C.c_double(huber_delta), C.c_int32(iterations), C.byref(st))
    return P, st.as_dict()


def chi2(poses, ij, meas, info, robust, huber_delta=1.0):
    P = _f64(poses).reshape(-1, 12); ijc = _i32(ij).reshape(-1, 2)
    Z = _f64(meas).reshape(-1, 12); Om = _f64(info).reshape(-1, 36); rb = _u8(robust)
    return lib().uzlo_chi2(C.c_int32(P.shape[0]), _p(P, c_f64p), C.c_int32(ijc.shape[0]), _p(ijc, c_i32p),
                           _p(Z, c_f64p), _p(Om, c_f64p), _p(rb, c_u8p), C.c_double(huber_delta))


def edge_error_norms(poses, ij, meas):
    P = _f64(poses).reshape(-1, 12); ijc = _i32(ij).reshape(-1, 2); Z = _f64(meas).reshape(-1, 12)
    err = np.empty(ijc.shape[0])
    lib().uzlo_edge_error_norms(C.c_int32(P.shape[0]), _p(P, c_f64p), C.c_int32(ijc.shape[0]), _p(ijc, c_i32p),
                                _p(Z, c_f64p), _p(err, c_f64p))
    return err


def build_dense(poses, fixed, ij, meas, info, robust, huber_delta=1.0):
    P = _f64(poses).reshape(-1, 12); f = _u8(fixed); ijc = _i32(ij).reshape(-1, 2)
    Z = _f64(meas).reshape(-1, 12); Om = _f64(info).reshape(-1, 36); rb = _u8(robust)
    n = P.shape[0]
    H = np.empty((6 * n, 6 * n)); b = np.empty(6 * n)
    lib().uzlo_build_dense(C.c_int32(n), _p(P, c_f64p), _p(f, c_u8p), C.c_int32(ijc.shape[0]), _p(ijc, c_i32p),
                           _p(Z, c_f64p), _p(Om, c_f64p), _p(rb, c_u8p), C.c_double(huber_delta),
                           _p(H, c_f64p), _p(b, c_f64p))
    return H, b


# ------------------------------------------------------------------------------- edge filter (uzl_oracle_filter.c)
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


class ClusterInfo(C.Structure):
    _fields_ = [("uid", C.c_uint64), ("from_start_ns", C.c_int64), ("from_end_ns", C.c_int64),
                ("to_start_ns", C.c_int64), ("to_end_ns", C.c_int64), ("size", C.c_int32),
                ("consensus", C.c_int32), ("changed", C.c_int32), ("evaluations", C.c_int32)]


_IDENT12 = (1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0)


def _pack_filter_edges(edges):
    arr = (FilterEdge * max(len(edges), 1))()
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


class Filter:
    """CPU checker twin of uzliti_slam_amd.capi.Filter (same method names and return shapes)."""

    def __init__(self, **cfg):
        L = lib()
        L.uzlo_filter_create.restype = C.c_void_p
        L.uzlo_filter_destroy.argtypes = [C.c_void_p]
        c = FilterCfg()
        L.uzlo_filter_cfg_default(C.byref(c))
        for k, v in cfg.items():
            setattr(c, k, v)
        self.cfg = c
        self._h = C.c_void_p(L.uzlo_filter_create(C.byref(c)))

    def close(self):
        if getattr(self, "_h", None):
            lib().uzlo_filter_destroy(self._h)
            self._h = None

    __del__ = close

    def set_sensors(self, sensors):
        s = np.ascontiguousarray(sensors, np.float64).reshape(-1, 12)
        lib().uzlo_filter_set_sensors(self._h, C.c_int32(len(s)), _p(s, c_f64p))

    def add(self, edges):
        arr, keep = _pack_filter_edges(edges)
        lib().uzlo_filter_add(self._h, C.c_int32(len(edges)), arr)

    def remove(self, keys):
        k = np.ascontiguousarray(keys, np.uint64)
        lib().uzlo_filter_remove(self._h, C.c_int32(len(k)), k.ctypes.data_as(C.POINTER(C.c_uint64)))

    def all_edges(self):
        n = lib().uzlo_filter_all_edges(self._h, C.c_int32(0), None)
        out = np.zeros(max(n, 1), np.uint64)
        lib().uzlo_filter_all_edges(self._h, C.c_int32(len(out)), out.ctypes.data_as(C.POINTER(C.c_uint64)))
        return out[:n]

    def valid_edges(self):
        n = lib().uzlo_filter_valid_edges(self._h, C.c_int32(0), None)
        out = np.zeros(max(n, 1), np.uint64)
        lib().uzlo_filter_valid_edges(self._h, C.c_int32(len(out)), out.ctypes.data_as(C.POINTER(C.c_uint64)))
        return out[:n]

    def calc_valid_edges(self):
        return lib().uzlo_filter_calc_valid_edges(self._h)

    def clusters(self, with_eval=False):
        out = []
        L = lib()
        for i in range(L.uzlo_filter_cluster_count(self._h)):
            ci = ClusterInfo()
            L.uzlo_filter_cluster_info(self._h, C.c_int32(i), C.byref(ci))
            d = {f: getattr(ci, f) for f, _ in ci._fields_}
            keys = np.zeros(max(ci.size, 1), np.uint64); valid = np.zeros(max(ci.size, 1), np.uint8)
            L.uzlo_filter_cluster_edges(self._h, C.c_int32(i), keys.ctypes.data_as(C.POINTER(C.c_uint64)), _p(valid, c_u8p))
            d["keys"] = keys[:ci.size]; d["valid"] = valid[:ci.size]
            if with_eval:
                cap = max(ci.size + 128, 256)
                P = np.zeros((cap, 3)); Q = np.zeros((cap, 3)); T = np.zeros(12); rc = C.c_int32()
                m = L.uzlo_filter_cluster_last_eval(self._h, C.c_int32(i), _p(P, c_f64p), _p(Q, c_f64p), _p(T, c_f64p), C.byref(rc))
                d.update(P=P[:m].copy(), Q=Q[:m].copy(), T=T, ransac_consensus=rc.value)
            out.append(d)
        return out


# ------------------------------------------------------------------------------- edge acceptance gate (uzl_oracle_gate.c)
class GateCfg(C.Structure):
    _fields_ = [("min_matching_score", C.c_double), ("max_edge_distance_T", C.c_double), ("max_edge_distance_R", C.c_double),
                ("scope_size_factor", C.c_double), ("min_accept_valid", C.c_double), ("device", C.c_int32), ("pad", C.c_int32)]


GATE_EDGE_DTYPE = np.dtype([("from", "<i4"), ("to", "<i4"), ("type", "<i4"), ("valid", "<i4"), ("matching_score", "<f8"),
                            ("transform", "<f8", (12,))], align=True)


class Gate:
    """CPU checker twin of uzliti_slam_amd.capi.Gate."""

    def __init__(self, **cfg):
        L = lib()
        L.uzlo_gate_create.restype = C.c_void_p
        L.uzlo_gate_destroy.argtypes = [C.c_void_p]
        L.uzlo_gate_astar.restype = C.c_double
        L.uzlo_gate_astar.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.uzlo_gate_last_expansions.restype = C.c_int64
        L.uzlo_gate_last_expansions.argtypes = [C.c_void_p]
        c = GateCfg()
        L.uzlo_gate_cfg_default(C.byref(c))
        for k, v in cfg.items():
            setattr(c, k, v)
        self.cfg = c
        self._h = C.c_void_p(L.uzlo_gate_create(C.byref(c)))

    def close(self):
        if getattr(self, "_h", None):
            lib().uzlo_gate_destroy(self._h)
            self._h = None

    __del__ = close

    def set_graph(self, poses, edges, merged=None):
        P = np.ascontiguousarray(poses, np.float64).reshape(-1, 12)
        E = np.ascontiguousarray(edges, GATE_EDGE_DTYPE)
        m = None if merged is None else np.ascontiguousarray(merged, np.uint8)
        lib().uzlo_gate_set_graph(self._h, C.c_int32(len(P)), _p(P, c_f64p), None if m is None else _p(m, c_u8p), C.c_int32(len(E)),
                                  E.ctypes.data_as(C.c_void_p) if len(E) else None)

    def astar(self, source, target):
        return lib().uzlo_gate_astar(self._h, int(source), int(target))

    def last_expansions(self):
        return lib().uzlo_gate_last_expansions(self._h)

    def check(self, cand):
        Cn = np.ascontiguousarray(cand, GATE_EDGE_DTYPE)
        n = len(Cn)
        acc = np.zeros(max(n, 1), np.uint8); val = np.zeros(max(n, 1), np.uint8); dist = np.zeros(max(n, 1))
        lib().uzlo_gate_check(self._h, C.c_int32(n), Cn.ctypes.data_as(C.c_void_p) if n else None, _p(acc, c_u8p), _p(val, c_u8p), _p(dist, c_f64p))
        return acc[:n], val[:n], dist[:n]

    def edge_count(self):
        return lib().uzlo_gate_edge_count(self._h)


# ------------------------------------------------------------------------------- distance loop-closure candidates
def radius_candidates(poses, stamps_front_ns, queries, radius=0.5, new_edge_time=5.0, max_rotation_deg=30.0):
    """-> (from, to, count_per_query): jobs (close node, query node) in query order, then node order."""
    P = np.ascontiguousarray(poses, np.float64).reshape(-1, 12); n = len(P)
    st = np.ascontiguousarray(stamps_front_ns, np.int64); q = np.ascontiguousarray(queries, np.int32)
    L = lib(); L.uzlo_radius_candidates.restype = C.c_int64
    cap = max(1, n * len(q))
    f = np.zeros(cap, np.int32); t = np.zeros(cap, np.int32); cnt = np.zeros(max(len(q), 1), np.int32)
    tot = L.uzlo_radius_candidates(C.c_int32(n), _p(P, c_f64p), st.ctypes.data_as(C.POINTER(C.c_int64)), C.c_double(radius),
                                   C.c_double(new_edge_time), C.c_double(max_rotation_deg), C.c_int32(len(q)), _p(q, c_i32p),
                                   C.c_int64(cap), _p(f, c_i32p), _p(t, c_i32p), _p(cnt, c_i32p))
    return f[:tot].copy(), t[:tot].copy(), cnt[:len(q)].copy()


# ------------------------------------------------------------------------------- appearance-based candidates (uzl_oracle_places.c)
class PlacesCfg(C.Structure):
    _fields_ = [("key_width", C.c_int32), ("min_rows_to_add", C.c_int32), ("T", C.c_double), ("k_nearest_neighbors", C.c_int32),
                ("device", C.c_int32), ("min_time_gap", C.c_double)]


class Places:
    """CPU checker twin of uzliti_slam_amd.capi.Places."""

    def __init__(self, **cfg):
        L = lib()
        L.uzlo_places_create.restype = C.c_void_p
        L.uzlo_places_destroy.argtypes = [C.c_void_p]
        for f in ("uzlo_places_search_and_add", "uzlo_places_add", "uzlo_places_search", "uzlo_places_count", "uzlo_places_last_counts", "uzlo_places_num_tables"):
            getattr(L, f).restype = C.c_int32
        c = PlacesCfg()
        L.uzlo_places_cfg_default(C.byref(c))
        for k, v in cfg.items():
            setattr(c, k, v)
        self.cfg = c
        self._h = C.c_void_p(L.uzlo_places_create(C.byref(c)))

    def close(self):
        if getattr(self, "_h", None):
            lib().uzlo_places_destroy(self._h)
            self._h = None

    __del__ = close

    @staticmethod
    def _d(desc):
        d = np.ascontiguousarray(desc, np.uint8)
        return d, (d.shape[0] if d.ndim == 2 else 0), (d.shape[1] if d.ndim == 2 else 32)

    def search_and_add(self, desc, stamp_ns, cap=64):
        d, rows, nb = self._d(desc)
        out = np.zeros(max(cap, 1), np.int32); idx = C.c_int32()
        n = lib().uzlo_places_search_and_add(self._h, _p(d, c_u8p) if rows else None, C.c_int32(rows), C.c_int32(nb), C.c_int64(int(stamp_ns)),
                                             C.c_int32(cap), _p(out, c_i32p), C.byref(idx))
        return out[:min(n, cap)].copy(), idx.value

    def add(self, desc, stamp_ns):
        d, rows, nb = self._d(desc)
        return lib().uzlo_places_add(self._h, _p(d, c_u8p) if rows else None, C.c_int32(rows), C.c_int32(nb), C.c_int64(int(stamp_ns)))

    def search(self, desc, stamp_ns, query_place=-1, cap=64):
        d, rows, nb = self._d(desc)
        out = np.zeros(max(cap, 1), np.int32)
        n = lib().uzlo_places_search(self._h, _p(d, c_u8p) if rows else None, C.c_int32(rows), C.c_int32(nb), C.c_int64(int(stamp_ns)),
                                     C.c_int32(query_place), C.c_int32(cap), _p(out, c_i32p))
        return out[:min(n, cap)].copy()

    def remove(self, place, desc):
        d, rows, nb = self._d(desc)
        lib().uzlo_places_remove(self._h, C.c_int32(place), _p(d, c_u8p) if rows else None, C.c_int32(rows), C.c_int32(nb))

    def count(self):
        return lib().uzlo_places_count(self._h)

    def num_tables(self):
        return lib().uzlo_places_num_tables(self._h)

    def last_counts(self):
        n = lib().uzlo_places_last_counts(self._h, C.c_int32(0), None)
        out = np.zeros(max(n, 1), np.int32)
        lib().uzlo_places_last_counts(self._h, C.c_int32(len(out)), _p(out, c_i32p))
        return out[:n]
