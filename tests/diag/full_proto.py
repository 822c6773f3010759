"""numpy/scipy prototype: PCG iteration counts of two-level preconditioners on the FULL system (no Schur reduction) of a loopy graph -
BASELINE config 2 / 4, first linearisation - for different aggregations of the vertices (measure before building).
   python tests/diag/full_proto.py N E [lambda_scale] [lm_iteration]"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
import oracle as O, np_reference as NP
from uzliti_slam_amd import synth

N, E = int(sys.argv[1]), int(sys.argv[2])
lam_scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
lm_it = int(sys.argv[4]) if len(sys.argv) > 4 else 0
g = synth.make_pose_graph(N, E, seed=12345)
fl = O.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
fixed, _ = O.set_fixed_nodes(fl["fixed"], fl["ij"])
poses = fl["poses"]
if lm_it > 0:      # a later linearisation: the oracle's poses after lm_it iterations
    poses, _ = O.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=lm_it)
t0 = time.time()
H, b, chi = NP.build_system(poses, fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"])
free_v = np.nonzero(fixed == 0)[0]
nb = len(free_v)
fidx = (6 * free_v[:, None] + np.arange(6)).ravel()
A0 = H[fidx][:, fidx].tocsr()
lam = 1e-5 * np.abs(A0.diagonal()).max() * lam_scale
A = (A0 + lam * sp.identity(A0.shape[0])).tocsr()
bf = b[fidx]
print("system: %d free vertices, %d edges, lambda %.3g, build %.1fs" % (nb, len(fl["ij"]) // 2 if np.ndim(fl["ij"]) == 1 else len(fl["ij"]), lam, time.time() - t0), flush=True)
X = np.asarray(poses).reshape(-1, 3, 4)[free_v]
R, t = X[:, :, :3], X[:, :, 3]
Ab = A.tobsr((6, 6)); Ab.sort_indices()
Wn = sp.csr_matrix((np.linalg.norm(Ab.data.reshape(-1, 36), axis=1), Ab.indices.copy(), Ab.indptr.copy()), shape=(nb, nb))
Wn.setdiag(0); Wn.eliminate_zeros()
# information-trace weights (what the library has on the host: no block norms before the first linearisation)
v2b = -np.ones(len(fixed), int); v2b[free_v] = np.arange(nb)
ij = np.asarray(fl["ij"]).reshape(-1, 2)
tr = np.asarray(fl["info"]).reshape(-1, 6, 6)[:, np.arange(6), np.arange(6)].sum(1)
ia, ic = v2b[ij[:, 0]], v2b[ij[:, 1]]
ok = (ia >= 0) & (ic >= 0)
Wt = sp.coo_matrix((np.r_[tr[ok], tr[ok]], (np.r_[ia[ok], ic[ok]], np.r_[ic[ok], ia[ok]])), shape=(nb, nb)).tocsr()
print("graph: degree mean %.2f max %d" % (np.diff(Wn.indptr).mean(), np.diff(Wn.indptr).max()))


def skew(v):
    zz = np.zeros(len(v))
    return np.stack([np.stack([zz, -v[:, 2], v[:, 1]], 1), np.stack([v[:, 2], zz, -v[:, 0]], 1), np.stack([-v[:, 1], v[:, 0], zz], 1)], 1)


def prolong(groups):
    ng = groups.max() + 1
    cen = np.stack([np.bincount(groups, t[:, k], ng) / np.maximum(np.bincount(groups, minlength=ng), 1) for k in range(3)], 1)
    Rt = np.swapaxes(R, 1, 2); d = t - cen[groups]
    blk = np.zeros((nb, 6, 6)); blk[:, :3, :3] = Rt; blk[:, :3, 3:] = -Rt @ skew(d); blk[:, 3:, 3:] = 0.5 * Rt
    rr = (6 * np.arange(nb)[:, None, None] + np.arange(6)[None, :, None]) + np.zeros((1, 1, 6), int)
    cc = (6 * groups[:, None, None] + np.arange(6)[None, None, :]) + np.zeros((1, 6, 1), int)
    return sp.coo_matrix((blk.ravel(), (rr.ravel(), cc.ravel())), shape=(6 * nb, 6 * ng)).tocsr()


def block_inv(groups):
    out_r, out_c, out_v = [], [], []
    order = np.argsort(groups, kind="stable"); bounds = np.searchsorted(groups[order], np.arange(groups.max() + 2))
    for gi in range(groups.max() + 1):
        ent = order[bounds[gi]:bounds[gi + 1]]
        idx = (6 * ent[:, None] + np.arange(6)).ravel()
        Wm = np.linalg.inv(A[idx][:, idx].toarray())
        rr, cc = np.meshgrid(idx, idx, indexing="ij")
        out_r.append(rr.ravel()); out_c.append(cc.ravel()); out_v.append(Wm.ravel())
    return sp.coo_matrix((np.concatenate(out_v), (np.concatenate(out_r), np.concatenate(out_c))), shape=A.shape).tocsr()


def pcg(Minv, tol=1e-5, maxit=3000):
    x = np.zeros_like(bf); r = bf.copy(); zv = Minv(r); p = zv.copy(); rz = r @ zv; thr = tol * tol * rz
    for it in range(1, maxit + 1):
        Ap = A @ p; a = rz / (p @ Ap); x += a * p; r -= a * Ap
        zv = Minv(r); rzn = r @ zv
        if not rzn > thr: return it
        p = zv + (rzn / rz) * p; rz = rzn
    return maxit


def two_level(groups, name, smooth_groups=None):
    P1 = prolong(groups); S = block_inv(groups if smooth_groups is None else smooth_groups); A1 = (P1.T @ A @ P1).tocsc(); lu1 = spl.splu(A1)
    add = pcg(lambda r: S @ r + P1 @ lu1.solve(P1.T @ r))
    def mult(r):
        y = S @ r; y = y + P1 @ lu1.solve(P1.T @ (r - A @ y)); return y + S @ (r - A @ y)
    sizes = np.bincount(groups)
    print("%-72s aggs %5d (max %2d, mean %.1f)  additive %4d its   multiplicative %4d its" % (name, groups.max() + 1, sizes.max(), sizes.mean(), add, pcg(mult)), flush=True)
    return add


def match_capped(Wm, cap, theta, rounds=3):
    """the library's schur_plan grouping: size-capped heavy-edge matching along edges at least theta x as stiff as the stiffest at either end"""
    n = Wm.shape[0]; grp = np.arange(n); Wc = Wm.tocsr().copy(); size = np.ones(n, int)
    for _ in range(rounds):
        m = Wc.shape[0]
        coo = Wc.tocoo(); mx = np.zeros(m); np.maximum.at(mx, coo.row, coo.data)
        strong = coo.data >= theta * np.maximum(mx[coo.row], mx[coo.col])
        o = np.argsort(-coo.data, kind="stable"); mate = -np.ones(m, int)
        for k in o:
            if not strong[k]: continue
            a, c = coo.row[k], coo.col[k]
            if a != c and mate[a] < 0 and mate[c] < 0 and size[a] + size[c] <= cap: mate[a] = c; mate[c] = a
        new = -np.ones(m, int); cnt = 0
        for v in range(m):
            if new[v] < 0:
                new[v] = cnt
                if mate[v] >= 0: new[mate[v]] = cnt
                cnt += 1
        Pm = sp.coo_matrix((np.ones(m), (np.arange(m), new)), shape=(m, cnt)).tocsr()
        Wc = (Pm.T @ Wc @ Pm).tocsr(); Wc.setdiag(0); Wc.eliminate_zeros()
        size = np.bincount(new, size, cnt).astype(int)
        grp = new[grp]
    return grp


def greedy_grow(Wm, size):
    n = Wm.shape[0]; grp = -np.ones(n, int); ind, ptr, dat = Wm.indices, Wm.indptr, Wm.data; cnt = 0
    for s in range(n):
        if grp[s] >= 0: continue
        members = [s]; grp[s] = cnt; conn = {}
        def add_nb(v):
            for q in range(ptr[v], ptr[v + 1]):
                u = ind[q]
                if grp[u] < 0: conn[u] = conn.get(u, 0.) + dat[q]
        add_nb(s)
        while len(members) < size and conn:
            u = max(conn.items(), key=lambda kv: (kv[1], -kv[0]))[0]
            del conn[u]; grp[u] = cnt; members.append(u); add_nb(u)
        cnt += 1
    return grp


def spatial(cell, cap):
    """vertices of one grid cell (poses' x, y) in trajectory order, at most `cap` to an aggregate"""
    key = np.floor(t[:, :2] / cell).astype(np.int64)
    key = key[:, 0] * 100003 + key[:, 1]
    o = np.lexsort((np.arange(nb), key))
    grp = np.empty(nb, int); cnt = -1; last = None; fill = 0
    for v in o:
        if key[v] != last or fill == cap: cnt += 1; fill = 0; last = key[v]
        grp[v] = cnt; fill += 1
    return grp


base = two_level(np.arange(nb) // 8, "8 consecutive vertices (what the library does)")
two_level(np.arange(nb) // 4, "4 consecutive vertices")
two_level(np.arange(nb) // 16, "16 consecutive vertices (96 x 96 blocks)")
for th in (0.25, 0.1, 0.0):
    two_level(match_capped(Wn, 8, th), "heavy-edge matching, block norms, cap 8, theta %.2f" % th)
two_level(match_capped(Wt, 8, 0.25), "heavy-edge matching, info-trace weights, cap 8, theta 0.25")
two_level(match_capped(Wn, 16, 0.1, rounds=4), "heavy-edge matching, block norms, cap 16, theta 0.10")
two_level(greedy_grow(Wn, 8), "greedy growth to 8 (block norms)")
for cell in (0.5, 1.0, 1.5):
    two_level(spatial(cell, 8), "spatial cells of %.1f m, <= 8 per aggregate" % cell)
# same coarse space as today, larger smoother blocks: what do exact 96 x 96 sibling blocks buy?
two_level(np.arange(nb) // 8, "8 consecutive for the coarse space, 16 consecutive for the smoother", smooth_groups=np.arange(nb) // 16)
two_level(np.arange(nb) // 4, "4 consecutive for the coarse space, 8 consecutive for the smoother", smooth_groups=np.arange(nb) // 8)


# ---- overlapping blocks (additive Schwarz) with the same coarse space: does an overlap along the chain order buy iterations?
def overlap_blocks(ov, size=8):
    blocks = []
    for k in range((nb + size - 1) // size):
        lo, hi = max(0, size * k - ov), min(nb, size * k + size + ov)
        blocks.append(np.arange(lo, hi))
    return blocks


def schwarz(blocks, restricted=False, size=8, weight=None):
    invs = []
    for ent in blocks:
        idx = (6 * ent[:, None] + np.arange(6)).ravel()
        invs.append((idx, ent, np.linalg.inv(A[idx][:, idx].toarray())))
    def apply(r):
        y = np.zeros_like(r)
        for k, (idx, ent, Wm) in enumerate(invs):
            z = Wm @ r[idx]
            if restricted:                      # only the block's own rows keep their result
                own = (ent // size == k)
                z = z * np.repeat(own, 6)
            elif weight is not None:
                z = z * np.repeat(weight[ent], 6)
            y[idx] += z
        return y
    return apply


P8 = prolong(np.arange(nb) // 8); lu8 = spl.splu((P8.T @ A @ P8).tocsc())
for ov in (0, 1, 2, 4):
    bl = overlap_blocks(ov)
    cnt = np.zeros(nb); [np.add.at(cnt, b, 1) for b in bl]
    S_as = schwarz(bl)
    S_w = schwarz(bl, weight=1. / np.sqrt(cnt))       # symmetric weighting D^-1/2 ... D^-1/2 needs both sides: applied on the way out only -> not symmetric; kept as a data point
    S_ras = schwarz(bl, restricted=True)
    for om in (1.0,):
        it_as = pcg(lambda r: S_as(r) + om * (P8 @ lu8.solve(P8.T @ r)))
    it_ras = pcg(lambda r: S_ras(r) + P8 @ lu8.solve(P8.T @ r))
    print("overlap %d along the chain order (blocks of %d vertices): additive Schwarz + coarse %4d its; restricted (non-symmetric, CG not guaranteed) %4d its" % (ov, 8 + 2 * ov, it_as, it_ras), flush=True)
# relative weight of the coarse term in the additive combination
S8 = block_inv(np.arange(nb) // 8)
for om in (0.5, 0.75, 1.0, 1.5, 2.0, 3.0):
    print("coarse term x %.2f: %4d its" % (om, pcg(lambda r: S8 @ r + om * (P8 @ lu8.solve(P8.T @ r)))), flush=True)
