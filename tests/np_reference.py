"""Independent NumPy/SciPy second implementation of the hot path's arithmetic (test infrastructure).

It exists to pin the C oracle (oracle/uzl_oracle_*.c), which has no reference golden vectors to be
checked against.  It shares no code with the oracle and deliberately takes different
routes to the same numbers:
  * Hamming distances through np.unpackbits, 2-NN through a stable argsort;
  * the rigid fit through np.linalg.svd in float64 (Kabsch), compared within tolerance;
  * EdgeSE3 error through scipy.spatial.transform.Rotation, Jacobians through central differences
    (g2o's own fallback is numeric differentiation [EXT]), the LM linear solve through
    scipy.sparse.linalg.spsolve (sparse direct, like the reference's CSparse).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl
from scipy.spatial.transform import Rotation


# ----------------------------------------------------------------------------- matching
def hamming_matrix(query, train):
    q = np.unpackbits(np.ascontiguousarray(query, np.uint8), axis=1).astype(np.int32)
    t = np.unpackbits(np.ascontiguousarray(train, np.uint8), axis=1).astype(np.int32)
    # |a xor b| = |a| + |b| - 2 a.b
    return q.sum(1)[:, None] + t.sum(1)[None, :] - 2 * (q @ t.T)


def knn2(query, train):
    D = hamming_matrix(query, train)
    nq, nt = D.shape
    order = np.argsort(D, axis=1, kind="stable")          # ties -> lower train index first
    idx0 = order[:, 0] if nt >= 1 else np.full(nq, -1)
    idx1 = order[:, 1] if nt >= 2 else np.full(nq, -1)
    d0 = D[np.arange(nq), idx0] if nt >= 1 else np.full(nq, -1)
    d1 = D[np.arange(nq), idx1] if nt >= 2 else np.full(nq, -1)
    return idx0.astype(np.int32), d0.astype(np.int32), idx1.astype(np.int32), d1.astype(np.int32)


def filter_sort(idx0, d0, idx1, d1, valid_train, valid_query):
    ok = (idx0 >= 0) & (idx1 >= 0)
    ratio = ok & (d0.astype(np.float32).astype(np.float64) < 0.99 * d1.astype(np.float32).astype(np.float64))
    n_ratio = int(ratio.sum())
    q = np.nonzero(ratio)[0]
    keep = (np.asarray(valid_train)[idx0[q]] != 0) & (np.asarray(valid_query)[q] != 0)
    q = q[keep]
    order = np.lexsort((q, d0[q]))                         # (distance, queryIdx)
    q = q[order]
    return q.astype(np.int32), idx0[q].astype(np.int32), d0[q].astype(np.int32), n_ratio


def kabsch(P, Q):
    """Least-squares rigid T (3,4) with Q ~= T P, float64 (Arun/Kabsch)."""
    mp = P.mean(1, keepdims=True); mq = Q.mean(1, keepdims=True)
    C = (Q - mq) @ (P - mp).T
    U, S, Vt = np.linalg.svd(C)
    s = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        s[2, 2] = -1
    R = U @ s @ Vt
    return np.concatenate([R, mq - R @ mp], axis=1)


def point_distances(P, Q, T):
    return np.linalg.norm(T[:, :3] @ P + T[:, 3:4] - Q, axis=0)


# ----------------------------------------------------------------------------- SE(3)
def se3_mul(A, B):
    R = A[..., :3, :3] @ B[..., :3, :3]
    t = (A[..., :3, :3] @ B[..., :3, 3:4])[..., 0] + A[..., :3, 3]
    return np.concatenate([R, t[..., None]], axis=-1)


def se3_inv(A):
    Rt = np.swapaxes(A[..., :3, :3], -1, -2)
    t = -(Rt @ A[..., :3, 3:4])[..., 0]
    return np.concatenate([Rt, t[..., None]], axis=-1)


def to_vector_mqt(T):
    """(...,3,4) -> (...,6): translation + (qx,qy,qz) of the unit quaternion with w >= 0."""
    T = np.asarray(T)
    q = Rotation.from_matrix(T[..., :3, :3].reshape(-1, 3, 3)).as_quat()     # (x,y,z,w)
    q = q * np.where(q[:, 3:4] < 0, -1.0, 1.0)
    v = np.concatenate([T[..., :3, 3].reshape(-1, 3), q[:, :3]], axis=1)
    return v.reshape(T.shape[:-2] + (6,))


def from_vector_mqt(v):
    v = np.asarray(v, np.float64)
    vv = v.reshape(-1, 6)
    w2 = 1.0 - (vv[:, 3:] ** 2).sum(1)
    q = np.concatenate([vv[:, 3:], np.sqrt(np.maximum(w2, 0.0))[:, None]], axis=1)
    R = Rotation.from_quat(q).as_matrix()
    R[w2 < 0] = np.eye(3)
    T = np.concatenate([R, vv[:, :3, None]], axis=2)
    return T.reshape(v.shape[:-1] + (3, 4))


def edge_errors(poses, ij, meas):
    X = np.asarray(poses).reshape(-1, 3, 4); Z = np.asarray(meas).reshape(-1, 3, 4)
    E = se3_mul(se3_inv(Z), se3_mul(se3_inv(X[ij[:, 0]]), X[ij[:, 1]]))
    return to_vector_mqt(E)


def huber(e2, delta=1.0):
    e2 = np.asarray(e2, np.float64)
    big = e2 > delta * delta
    sq = np.sqrt(np.where(big, e2, 1.0))
    rho0 = np.where(big, 2 * sq * delta - delta * delta, e2)
    rho1 = np.where(big, delta / sq, 1.0)
    return rho0, rho1


def chi2(poses, ij, meas, info, robust, delta=1.0):
    e = edge_errors(poses, ij, meas)
    Om = np.asarray(info).reshape(-1, 6, 6)
    c = np.einsum("ki,kij,kj->k", e, Om, e)
    r0, _ = huber(c, delta)
    return float(np.where(np.asarray(robust) != 0, r0, c).sum())


def numeric_jacobians(poses, ij, meas, h=1e-6):
    X = np.asarray(poses).reshape(-1, 3, 4); Z = np.asarray(meas).reshape(-1, 3, 4)
    Xi = X[ij[:, 0]]; Xj = X[ij[:, 1]]
    Zi = se3_inv(Z)

    def err(Xi_, Xj_):
        return to_vector_mqt(se3_mul(Zi, se3_mul(se3_inv(Xi_), Xj_)))

    E = ij.shape[0]
    Ji = np.empty((E, 6, 6)); Jj = np.empty((E, 6, 6))
    for k in range(6):
        d = np.zeros(6); d[k] = h
        Dp = from_vector_mqt(d); Dm = from_vector_mqt(-d)
        Ji[:, :, k] = (err(se3_mul(Xi, Dp), Xj) - err(se3_mul(Xi, Dm), Xj)) / (2 * h)
        Jj[:, :, k] = (err(Xi, se3_mul(Xj, Dp)) - err(Xi, se3_mul(Xj, Dm))) / (2 * h)
    return Ji, Jj


def build_system(poses, fixed, ij, meas, info, robust, delta=1.0, jac=None):
    """Sparse H (BSR over all vertices; fixed rows/cols empty), b, chi2."""
    n = np.asarray(poses).reshape(-1, 12).shape[0]
    ij = np.asarray(ij).reshape(-1, 2)
    e = edge_errors(poses, ij, meas)
    Om = np.asarray(info).reshape(-1, 6, 6)
    c = np.einsum("ki,kij,kj->k", e, Om, e)
    r0, r1 = huber(c, delta)
    rb = np.asarray(robust) != 0
    w = np.where(rb, r1, 1.0)
    chi = float(np.where(rb, r0, c).sum())
    Ji, Jj = jac if jac is not None else numeric_jacobians(poses, ij, meas)
    Ow = Om * w[:, None, None]
    OJi = Ow @ Ji; OJj = Ow @ Jj
    Hii = np.swapaxes(Ji, 1, 2) @ OJi; Hjj = np.swapaxes(Jj, 1, 2) @ OJj; Hij = np.swapaxes(Ji, 1, 2) @ OJj
    Oe = np.einsum("kij,kj->ki", Ow, e)
    bi = -np.einsum("kji,kj->ki", Ji, Oe); bj = -np.einsum("kji,kj->ki", Jj, Oe)
    fi = np.asarray(fixed)[ij[:, 0]] == 0; fj = np.asarray(fixed)[ij[:, 1]] == 0
    rows = []; cols = []; blocks = []
    rows.append(ij[fi, 0]); cols.append(ij[fi, 0]); blocks.append(Hii[fi])
    rows.append(ij[fj, 1]); cols.append(ij[fj, 1]); blocks.append(Hjj[fj])
    both = fi & fj
    rows.append(ij[both, 0]); cols.append(ij[both, 1]); blocks.append(Hij[both])
    rows.append(ij[both, 1]); cols.append(ij[both, 0]); blocks.append(np.swapaxes(Hij[both], 1, 2))
    rows = np.concatenate(rows); cols = np.concatenate(cols); blocks = np.concatenate(blocks)
    # scalar COO
    rr = (6 * rows[:, None, None] + np.arange(6)[None, :, None]) + np.zeros((1, 1, 6), np.int64)
    cc = (6 * cols[:, None, None] + np.arange(6)[None, None, :]) + np.zeros((1, 6, 1), np.int64)
    H = sp.coo_matrix((blocks.ravel(), (rr.ravel(), cc.ravel())), shape=(6 * n, 6 * n)).tocsr()
    b = np.zeros((n, 6))
    np.add.at(b, ij[fi, 0], bi[fi]); np.add.at(b, ij[fj, 1], bj[fj])
    return H, b.reshape(-1), chi


def pgo_lm(poses, fixed, ij, meas, info, robust, iterations=20, delta=1.0):
    """LM as g2o's OptimizationAlgorithmLevenberg [EXT] with a sparse direct solve."""
    X = np.asarray(poses, np.float64).reshape(-1, 3, 4).copy()
    fixed = np.asarray(fixed); ij = np.asarray(ij).reshape(-1, 2)
    free = np.repeat(fixed == 0, 6)
    fidx = np.nonzero(free)[0]
    lam = 0.0; ni = 2.0
    stats = dict(iterations_done=0, lm_trials=0, terminated_early=0)
    for it in range(iterations):
        H, b, cur = build_system(X, fixed, ij, meas, info, robust, delta)
        Hf = H[fidx][:, fidx].tocsc(); bf = b[fidx]
        if it == 0:
            stats["chi2_initial"] = cur
            lam = 1e-5 * np.abs(Hf.diagonal()).max()
            ni = 2.0
        rho = 0.0; qmax = 0
        while True:
            A = Hf + lam * sp.identity(Hf.shape[0], format="csc")
            dx = spl.spsolve(A, bf)
            stats["lm_trials"] += 1
            full = np.zeros(free.shape[0]); full[fidx] = dx
            Xn = se3_mul(X, from_vector_mqt(full.reshape(-1, 6)))
            tmp = chi2(Xn, ij, meas, info, robust, delta)
            rho = (cur - tmp) / (float(dx @ (lam * dx + bf)) + 1e-3)
            if rho > 0 and np.isfinite(tmp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha); ni = 2.0
                cur = tmp; X = Xn
            else:
                lam *= ni; ni *= 2
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        stats["iterations_done"] = it + 1
        stats["chi2_final"] = cur
        if qmax == 10 or rho == 0:
            stats["terminated_early"] = 1
            break
    stats["lambda_final"] = lam
    return X.reshape(-1, 12), stats


# ------------------------------------------------------------------------------------------------------------
# Edge filter: a third, pure-Python statement of TransformationFilter / EdgeCluster
# (transformation_estimation/src/transformation_filter.cpp:43-350) with the reference's object semantics
# (shared cluster objects, only the first listing repointed on merge).  Small cases only.  Point pairs are
# computed with Python floats in the oracle's operation order, so they are bit-identical; the RANSAC itself is
# passed in (it has its own independent checks above).
# ------------------------------------------------------------------------------------------------------------
def _iso_mul(A, B):
    o = [0.0] * 12
    for r in range(3):
        for c in range(3):
            o[r * 4 + c] = (A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c]) + A[r * 4 + 2] * B[2 * 4 + c]
        o[r * 4 + 3] = ((A[r * 4 + 0] * B[3] + A[r * 4 + 1] * B[7]) + A[r * 4 + 2] * B[11]) + A[r * 4 + 3]
    return o


def _iso_inv(A):
    o = [0.0] * 12
    for r in range(3):
        for c in range(3):
            o[r * 4 + c] = A[c * 4 + r]
        o[r * 4 + 3] = -((A[0 * 4 + r] * A[3] + A[1 * 4 + r] * A[7]) + A[2 * 4 + r] * A[11])
    return o


_I12 = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0]


class _Cluster:
    def __init__(self, uid, e, tf, tt):
        self.uid = uid
        self.fs = self.fe = tf
        self.ts = self.te = tt
        self.changed = False
        self.consensus = 0
        self.evals = 0
        self.edges = {}                       # key -> dict (python dicts keep insertion order)
        self.put(e, tf, tt)

    def put(self, e, tf, tt):
        d = dict(e); d["t_from"] = tf; d["t_to"] = tt; d["valid_"] = bool(e["valid"])
        self.edges[e["key"]] = d              # a present key keeps its slot
        if e["valid"]:
            self.consensus += 1

    def add(self, e, tf, tt):
        self.fs = min(tf, self.fs); self.fe = max(tf, self.fe); self.ts = min(tt, self.ts); self.te = max(tt, self.te)
        self.changed = True
        self.put(e, tf, tt)

    def is_part(self, tf, tt, max_dt):
        s = lambda a, b: (a - b) * 1e-9
        return s(tf, self.fs) > -max_dt and s(tf, self.fe) < max_dt and s(tt, self.ts) > -max_dt and s(tt, self.te) < max_dt

    def merge(self, o):
        self.fs = min(o.fs, self.fs); self.fe = max(o.fe, self.fe); self.ts = min(o.ts, self.ts); self.te = max(o.te, self.te)
        self.changed = True
        self.consensus += o.consensus
        for k, d in o.edges.items():
            self.edges.setdefault(k, d)


class FilterRef:
    def __init__(self, max_dt=5.0, min_size=8.0, max_cluster_size=100, ransac_iterations=200, max_error=0.3,
                 min_time_span=2.0, max_edges=5, seed=0):
        self.__dict__.update(locals())
        self.clusters = []
        self.edges = {}
        self.sensors = []
        self.next_uid = 0

    def set_sensors(self, sensors):
        self.sensors = [list(map(float, s)) for s in np.asarray(sensors).reshape(-1, 12)]

    def add(self, edges):
        for e in edges:
            if e["key"] in self.edges:
                for c in self.edges[e["key"]]:
                    if e["key"] in c.edges:
                        d = c.edges[e["key"]]
                        for f in ("pose_from", "pose_to", "transform", "displacement_from", "displacement_to", "sensor_from",
                                  "sensor_to", "matching_score", "valid"):
                            d[f] = e[f]
                continue
            for tf in map(int, e["stamps_from"]):
                for tt in map(int, e["stamps_to"]):
                    matched = [i for i, c in enumerate(self.clusters)
                               if len(c.edges) < self.max_cluster_size and c.is_part(tf, tt, self.max_dt)]
                    if not matched:
                        c = _Cluster(self.next_uid, e, tf, tt); self.next_uid += 1
                        self.clusters.append(c)
                        self.edges.setdefault(e["key"], []).append(c)
                    else:
                        c0 = self.clusters[matched[0]]
                        c0.add(e, tf, tt)
                        self.edges.setdefault(e["key"], []).append(c0)
                        for i in reversed(matched[1:]):
                            ci = self.clusters[i]
                            if len(c0.edges) + len(ci.edges) < self.max_cluster_size:
                                for k in ci.edges:
                                    lst = self.edges.get(k, [])
                                    for u in range(len(lst)):
                                        if lst[u] is ci:
                                            lst[u] = c0
                                            break
                                c0.merge(ci)
                                del self.clusters[i]

    def remove(self, keys):
        for key in map(int, keys):
            if key not in self.edges:
                continue
            for c in self.edges[key]:
                if key in c.edges:
                    if c.edges[key]["valid_"]:
                        c.consensus -= 1
                    del c.edges[key]
                if len(c.edges) == 0:
                    self.clusters = [x for x in self.clusters if x is not c]
            del self.edges[key]

    def all_edges(self):
        return np.array(sorted(self.edges), np.uint64)

    def _points(self, d):
        Sf = self.sensors[d["sensor_from"]] if 0 <= d["sensor_from"] < len(self.sensors) else _I12
        St = self.sensors[d["sensor_to"]] if 0 <= d["sensor_to"] < len(self.sensors) else _I12
        f = lambda v: list(map(float, v))
        a = _iso_mul(f(d["pose_from"]), f(d["displacement_from"]))
        a = _iso_mul(a, Sf)
        a = _iso_mul(a, f(d["transform"]))
        a = _iso_mul(a, _iso_inv(St))
        b = _iso_mul(f(d["pose_to"]), f(d["displacement_to"]))
        return [a[3], a[7], a[11]], [b[3], b[7], b[11]]

    def calc_valid_edges(self, ransac):
        """ransac(P (m,3), Q (m,3), job_id) -> (T(12), set(m) of consensus3D with that T)"""
        n = 0
        for c in self.clusters:
            if len(c.edges) < self.min_size or not c.changed:
                continue
            if abs((c.fs - c.fe) * 1e-9) < self.min_time_span or abs((c.ts - c.te) * 1e-9) < self.min_time_span:
                continue
            c.changed = False
            pq = [self._points(d) for d in c.edges.values()]
            P = np.array([p for p, _ in pq]); Q = np.array([q for _, q in pq])
            T, s = ransac(P, Q, (c.uid << 20) + c.evals)
            c.evals += 1; n += 1
            c.lastP, c.lastQ = P, Q
            cons = int(np.sum(s))
            if cons >= self.min_size and cons >= c.consensus:
                c.consensus = cons
                for d, v in zip(c.edges.values(), s):
                    d["valid_"] = bool(v)
        return n

    def valid_edges(self):
        ids = set()
        for c in self.clusters:
            v = [d for d in c.edges.values() if d["valid_"]]
            if len(v) > 2 * self.max_edges:
                v = sorted(v, key=lambda d: -d["matching_score"])          # python's sort is stable
                ids.update(d["key"] for d in v[:self.max_edges])
                inc = len(v) / self.max_edges
                ids.update(v[int(np.floor(inc * i))]["key"] for i in range(self.max_edges - 1))
                ids.add(v[-1]["key"])
            else:
                ids.update(d["key"] for d in v)
        return np.array(sorted(ids), np.uint64)

    def state(self):
        return [dict(uid=c.uid, from_start_ns=c.fs, from_end_ns=c.fe, to_start_ns=c.ts, to_end_ns=c.te, size=len(c.edges),
                     consensus=c.consensus, changed=int(c.changed), evaluations=c.evals,
                     keys=np.array(list(c.edges), np.uint64), valid=np.array([d["valid_"] for d in c.edges.values()], np.uint8))
                for c in self.clusters]
