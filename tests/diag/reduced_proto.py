"""numpy/scipy prototype: PCG iteration counts of two-level preconditioners on the SCHUR-REDUCED system of config 5's last re-optimisation
(tests/diag/data/c5_last.npz), for different aggregations of the separators.   python tests/diag/reduced_proto.py [lambda_scale]"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
import oracle as O, np_reference as NP
from uzliti_slam_amd import capi

_here = os.path.dirname(os.path.abspath(__file__))
_cands = [os.path.join(_here, "data", "c5_last.npz"), os.path.join(os.path.dirname(os.path.dirname(_here)), "gpurun_out", "c5_last.npz")]
_found = [f for f in _cands if os.path.exists(f)]
if not _found:
    sys.exit("no c5_last.npz: run tests/diag/c5_last.py on a GPU box first (it writes gpurun_out/c5_last.npz; 4 MB, not tracked)")
z = np.load(_found[0])
edges = {k[2:]: z[k] for k in z.files if k.startswith("e_")}
fl = O.flatten_graph(z["poses"], z["fixed"], edges)
fixed, _ = O.set_fixed_nodes(fl["fixed"], fl["ij"])
H, b, chi = NP.build_system(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"])
free_v = np.nonzero(fixed == 0)[0]
nb = len(free_v)
fidx = (6 * free_v[:, None] + np.arange(6)).ravel()
A0 = H[fidx][:, fidx].tocsr()
lam = 1e-5 * np.abs(A0.diagonal()).max() * (float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
A = (A0 + lam * sp.identity(A0.shape[0])).tocsr()
bf = b[fidx]
print("full system:", nb, "free vertices", len(fl["ij"]), "edges  lambda %.3g" % lam)
# block structure over free vertices (index order = trajectory order for this run)
v2b = -np.ones(len(fixed), int); v2b[free_v] = np.arange(nb)
ij = fl["ij"].reshape(-1, 2)
rows = [[] for _ in range(nb)]
for a, c in ij:
    ba, bc = v2b[a], v2b[c]
    if ba >= 0: rows[ba].append(bc)
    if bc >= 0: rows[bc].append(ba)
rp = np.zeros(nb + 1, np.int32); rp[1:] = np.cumsum([len(r) for r in rows]); col = np.array([c for r in rows for c in r], np.int32)
P = capi.schur_plan(rp, col, 24)
red = P["red_row"]; sep = np.nonzero(red >= 0)[0]; inter = np.nonzero(red < 0)[0]
nr = len(sep)
print("reduced:", nr, "separators,", len(inter), "interiors,", P["n_runs"], "runs")
si = (6 * sep[:, None] + np.arange(6)).ravel(); ii = (6 * inter[:, None] + np.arange(6)).ravel()
Ass = A[si][:, si]; Asi = A[si][:, ii]; Aii = A[ii][:, ii].tocsc()
lu = spl.splu(Aii)
Ar = (Ass - Asi @ sp.csc_matrix(lu.solve(Asi.T.toarray()))).tocsr()
Ar = 0.5 * (Ar + Ar.T); Ar.eliminate_zeros()
br = bf[si] - Asi @ lu.solve(bf[ii])
X = fl["poses"].reshape(-1, 3, 4)[free_v][sep]
R, t = X[:, :, :3], X[:, :, 3]
# reduced graph: weights = trace of the off-diagonal blocks' "stiffness" (Frobenius norm of the 6x6 block)
Ab = Ar.tobsr((6, 6)); Ab.sort_indices()
W = sp.csr_matrix((np.linalg.norm(Ab.data.reshape(-1, 36), axis=1), Ab.indices.copy(), Ab.indptr.copy()), shape=(nr, nr))
dn = W.diagonal().copy()
W.setdiag(0); W.eliminate_zeros()
deg = np.diff(W.indptr)
print("reduced graph: %d off-diagonal blocks, degree mean %.2f max %d" % (W.nnz, deg.mean(), deg.max()))

def skew(v):
    zz = np.zeros(len(v))
    return np.stack([np.stack([zz, -v[:, 2], v[:, 1]], 1), np.stack([v[:, 2], zz, -v[:, 0]], 1), np.stack([-v[:, 1], v[:, 0], zz], 1)], 1)

def prolong(groups):
    ng = groups.max() + 1
    cen = np.stack([np.bincount(groups, t[:, k], ng) / np.maximum(np.bincount(groups, minlength=ng), 1) for k in range(3)], 1)
    Rt = np.swapaxes(R, 1, 2); d = t - cen[groups]
    blk = np.zeros((nr, 6, 6)); blk[:, :3, :3] = Rt; blk[:, :3, 3:] = -Rt @ skew(d); blk[:, 3:, 3:] = 0.5 * Rt
    rr = (6 * np.arange(nr)[:, None, None] + np.arange(6)[None, :, None]) + np.zeros((1, 1, 6), int)
    cc = (6 * groups[:, None, None] + np.arange(6)[None, None, :]) + np.zeros((1, 6, 1), int)
    return sp.coo_matrix((blk.ravel(), (rr.ravel(), cc.ravel())), shape=(6 * nr, 6 * ng)).tocsr()

def block_inv(groups):
    out_r, out_c, out_v = [], [], []
    order = np.argsort(groups, kind="stable"); bounds = np.searchsorted(groups[order], np.arange(groups.max() + 2))
    for g in range(groups.max() + 1):
        ent = order[bounds[g]:bounds[g + 1]]
        idx = (6 * ent[:, None] + np.arange(6)).ravel()
        Wm = np.linalg.inv(Ar[idx][:, idx].toarray())
        rr, cc = np.meshgrid(idx, idx, indexing="ij")
        out_r.append(rr.ravel()); out_c.append(cc.ravel()); out_v.append(Wm.ravel())
    return sp.coo_matrix((np.concatenate(out_v), (np.concatenate(out_r), np.concatenate(out_c))), shape=Ar.shape).tocsr()

def pcg(Minv, tol=1e-5, maxit=2000):
    x = np.zeros_like(br); r = br.copy(); zv = Minv(r); p = zv.copy(); rz = r @ zv; thr = tol * tol * rz
    for it in range(1, maxit + 1):
        Ap = Ar @ p; a = rz / (p @ Ap); x += a * p; r -= a * Ap
        zv = Minv(r); rzn = r @ zv
        if not rzn > thr: return it
        p = zv + (rzn / rz) * p; rz = rzn
    return maxit

def two_level(groups, name):
    P1 = prolong(groups); S = block_inv(groups); A1 = (P1.T @ Ar @ P1).tocsc(); lu1 = spl.splu(A1)
    add = pcg(lambda r: S @ r + P1 @ lu1.solve(P1.T @ r))
    def mult(r):
        y = S @ r; y = y + P1 @ lu1.solve(P1.T @ (r - Ar @ y)); return y + S @ (r - Ar @ y)
    sizes = np.bincount(groups)
    print("%-64s aggs %4d (size max %2d)  additive %4d its   multiplicative(level 0) %4d its" % (name, groups.max() + 1, sizes.max(), add, pcg(mult)), flush=True)

def walk_order(Wm):
    """the library's aggregation_order on the reduced graph: follow the heaviest edge to an unvisited vertex; extend backwards; repeat"""
    n = Wm.shape[0]; seen = np.zeros(n, bool); order = []
    ind, ptr, dat = Wm.indices, Wm.indptr, Wm.data
    def nxt(v):
        best, bw = -1, -1.
        for q in range(ptr[v], ptr[v + 1]):
            u = ind[q]
            if seen[u]: continue
            if dat[q] > bw or (dat[q] == bw and u < best): bw, best = dat[q], u
        return best
    for s in range(n):
        if seen[s]: continue
        fwd = []; v = s
        while v >= 0: seen[v] = True; fwd.append(v); v = nxt(v)
        back = []; v = nxt(s)
        while v >= 0: seen[v] = True; back.append(v); v = nxt(v)
        order += back[::-1] + fwd
    return np.array(order)

def greedy_match(Wm, size):
    """pairwise heavy-edge matching repeated log2(size) times (classic AMG aggregation)"""
    groups = np.arange(Wm.shape[0]); Wc = Wm.copy()
    for _ in range(int(np.log2(size))):
        n = Wc.shape[0]; mate = -np.ones(n, int)
        coo = Wc.tocoo(); o = np.argsort(-coo.data, kind="stable")
        for k in o:
            a, c = coo.row[k], coo.col[k]
            if a != c and mate[a] < 0 and mate[c] < 0: mate[a] = c; mate[c] = a
        new = -np.ones(n, int); cnt = 0
        for v in range(n):
            if new[v] < 0:
                new[v] = cnt
                if mate[v] >= 0: new[mate[v]] = cnt
                cnt += 1
        Pm = sp.coo_matrix((np.ones(n), (np.arange(n), new)), shape=(n, cnt)).tocsr()
        Wc = (Pm.T @ Wc @ Pm).tocsr(); Wc.setdiag(0); Wc.eliminate_zeros()
        groups = new[groups]
    return groups

two_level(np.arange(nr) // 8, "8 consecutive separators (what the library does)")
two_level(np.arange(nr) // 4, "4 consecutive separators")
two_level(np.arange(nr) // 16, "16 consecutive separators")
o = walk_order(W); g = np.empty(nr, int); g[o] = np.arange(nr) // 8
two_level(g, "heaviest-edge walk on the reduced graph, 8 consecutive of it")
g4 = np.empty(nr, int); g4[o] = np.arange(nr) // 4
two_level(g4, "heaviest-edge walk, 4 consecutive of it")
two_level(greedy_match(W, 8), "heavy-edge matching x3 (groups <= 8)")
two_level(greedy_match(W, 4), "heavy-edge matching x2 (groups <= 4)")

def greedy_grow(Wm, size, frag_merge=True):
    """seed = lowest unassigned index; repeatedly add the unassigned vertex most strongly connected to the group, until `size`"""
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
    if frag_merge:      # fragments (< size / 2) join the neighbouring group they are most strongly connected to, if it has room
        sizes = np.bincount(grp)
        for g in np.argsort(sizes, kind="stable"):
            if sizes[g] == 0 or sizes[g] > size // 2: continue
            mem = np.nonzero(grp == g)[0]; tot = {}
            for v in mem:
                for q in range(ptr[v], ptr[v + 1]):
                    h = grp[ind[q]]
                    if h != g and sizes[h] + sizes[g] <= size + 2: tot[h] = tot.get(h, 0.) + dat[q]
            if tot:
                h = max(tot.items(), key=lambda kv: kv[1])[0]; grp[mem] = h; sizes[h] += sizes[g]; sizes[g] = 0
        _, grp = np.unique(grp, return_inverse=True)
    return grp

two_level(greedy_grow(W, 8, False), "greedy growth to 8, no fragment merge")
two_level(greedy_grow(W, 8, True), "greedy growth to 8, fragments merged (groups <= 10)")
gm = greedy_match(W, 8)
print("matching x3 group sizes:", np.bincount(np.bincount(gm)))

def forced_match(Wm, rounds):
    """heavy-edge matching, leftovers paired with each other in index order: groups of exactly 2^rounds (but for one remainder)"""
    groups = np.arange(Wm.shape[0]); Wc = Wm.copy()
    for _ in range(rounds):
        n = Wc.shape[0]; mate = -np.ones(n, int)
        coo = Wc.tocoo(); o = np.argsort(-coo.data, kind="stable")
        for k in o:
            a, c = coo.row[k], coo.col[k]
            if a != c and mate[a] < 0 and mate[c] < 0: mate[a] = c; mate[c] = a
        left = np.nonzero(mate < 0)[0]
        for k in range(0, len(left) - 1, 2): mate[left[k]] = left[k + 1]; mate[left[k + 1]] = left[k]
        new = -np.ones(n, int); cnt = 0
        for v in range(n):
            if new[v] < 0:
                new[v] = cnt
                if mate[v] >= 0: new[mate[v]] = cnt
                cnt += 1
        Pm = sp.coo_matrix((np.ones(n), (np.arange(n), new)), shape=(n, cnt)).tocsr()
        Wc = (Pm.T @ Wc @ Pm).tocsr(); Wc.setdiag(0); Wc.eliminate_zeros()
        groups = new[groups]
    return groups

two_level(forced_match(W, 3), "heavy-edge matching x3, leftovers paired in index order (exact 8)")
two_level(forced_match(W, 2), "heavy-edge matching x2, leftovers paired (exact 4)")
# normalised strength: |A_ij| / sqrt(|A_ii| |A_jj|)
Wn = sp.diags(1 / np.sqrt(dn)) @ W @ sp.diags(1 / np.sqrt(dn)); Wn = Wn.tocsr()
two_level(forced_match(Wn, 3), "same, normalised strengths (exact 8)")
two_level(greedy_match(Wn, 8), "heavy-edge matching x3, normalised strengths (groups <= 8)")

def capped_match(Wm, cap, rounds, theta=0.0):
    """size-capped agglomeration: per round, heaviest-edge matching of groups whose sizes add up to <= cap; an edge only counts if it is
    at least theta x the heaviest edge at either end"""
    groups = np.arange(Wm.shape[0]); Wc = Wm.copy(); sizes = np.ones(Wm.shape[0], int)
    for _ in range(rounds):
        n = Wc.shape[0]; mate = -np.ones(n, int)
        coo = Wc.tocoo(); o = np.argsort(-coo.data, kind="stable")
        wmax = np.zeros(n); np.maximum.at(wmax, coo.row, coo.data)
        for k in o:
            a, c = coo.row[k], coo.col[k]
            if a != c and mate[a] < 0 and mate[c] < 0 and sizes[a] + sizes[c] <= cap and coo.data[k] >= theta * max(wmax[a], wmax[c]): mate[a] = c; mate[c] = a
        new = -np.ones(n, int); cnt = 0
        for v in range(n):
            if new[v] < 0:
                new[v] = cnt
                if mate[v] >= 0: new[mate[v]] = cnt
                cnt += 1
        if cnt == n: break
        Pm = sp.coo_matrix((np.ones(n), (np.arange(n), new)), shape=(n, cnt)).tocsr()
        Wc = (Pm.T @ Wc @ Pm).tocsr(); Wc.setdiag(0); Wc.eliminate_zeros()
        sizes = np.bincount(new, sizes, cnt).astype(int)
        groups = new[groups]
    return groups

for rounds in (4, 6, 10):
    for theta in (0.0, 0.25, 0.5):
        two_level(capped_match(W, 8, rounds, theta), "size-capped (<= 8) matching, %d rounds, theta %.2f" % (rounds, theta))
