"""numpy/scipy prototype: PCG iteration counts of multilevel variants on a config-4-sized system (first linearisation)."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
import oracle as O, np_reference as NP
from uzliti_slam_amd import synth

N, E = int(sys.argv[1]), int(sys.argv[2])
g = synth.make_pose_graph(N, E, seed=12345)
fl = O.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
fixed, _ = O.set_fixed_nodes(fl["fixed"], fl["ij"])
t0 = time.time()
H, b, chi = NP.build_system(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"])
free_v = np.nonzero(fixed == 0)[0]
fidx = (6 * free_v[:, None] + np.arange(6)).ravel()
A0 = H[fidx][:, fidx].tocsr()
lam = 1e-5 * np.abs(A0.diagonal()).max() * (float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
A = (A0 + lam * sp.identity(A0.shape[0])).tocsr()
bf = b[fidx]
nb = len(free_v)
print("system", nb, "free vertices, build %.1fs" % (time.time() - t0), "lambda %.3g" % lam)
X = fl["poses"].reshape(-1, 3, 4)[free_v]
R, t = X[:, :, :3], X[:, :, 3]

def skew(v):
    z = np.zeros(len(v))
    return np.stack([np.stack([z, -v[:, 2], v[:, 1]], 1), np.stack([v[:, 2], z, -v[:, 0]], 1), np.stack([-v[:, 1], v[:, 0], z], 1)], 1)

def prolong(groups, centers):
    """P: (6 nb) x (6 n_agg); groups[i] = aggregate of vertex i"""
    Rt = np.swapaxes(R, 1, 2)
    d = t - centers[groups]
    blk = np.zeros((nb, 6, 6))
    blk[:, :3, :3] = Rt; blk[:, :3, 3:] = -Rt @ skew(d); blk[:, 3:, 3:] = 0.5 * Rt
    rows = (6 * np.arange(nb)[:, None, None] + np.arange(6)[None, :, None]) + np.zeros((1, 1, 6), int)
    cols = (6 * groups[:, None, None] + np.arange(6)[None, None, :]) + np.zeros((1, 6, 1), int)
    return sp.coo_matrix((blk.ravel(), (rows.ravel(), cols.ravel())), shape=(6 * nb, 6 * (groups.max() + 1))).tocsr()

def block_inv(Al, member):
    """inverse of A_l restricted to groups of entities: member[i] = block id of entity i (consecutive)"""
    n = Al.shape[0] // 6
    out_r, out_c, out_v = [], [], []
    Al = Al.tocsr()
    for gid in range(member.max() + 1):
        ent = np.nonzero(member == gid)[0]
        idx = (6 * ent[:, None] + np.arange(6)).ravel()
        W = np.linalg.inv(Al[idx][:, idx].toarray())
        rr, cc = np.meshgrid(idx, idx, indexing="ij")
        out_r.append(rr.ravel()); out_c.append(cc.ravel()); out_v.append(W.ravel())
    return sp.coo_matrix((np.concatenate(out_v), (np.concatenate(out_r), np.concatenate(out_c))), shape=Al.shape).tocsr()

def hierarchy(fans):
    lv = []
    groups = np.arange(nb); cen = t.copy(); n = nb
    for f in fans:
        par = np.arange(n) // f
        cen = np.stack([np.bincount(par, cen[:, k]) / np.bincount(par) for k in range(3)], 1)
        groups = par[groups]; n = par.max() + 1
        P = prolong(groups, cen)
        lv.append(dict(n=n, P=P, A=(P.T @ A @ P).tocsr(), parent_fan=None))
    return lv

def pcg(Minv, tol=1e-5, maxit=3000):
    x = np.zeros_like(bf); r = bf.copy(); z = Minv(r); p = z.copy(); rz = r @ z; thr = tol * tol * rz
    for it in range(1, maxit + 1):
        Ap = A @ p; a = rz / (p @ Ap); x += a * p; r -= a * Ap
        z = Minv(r); rzn = r @ z
        if not rzn > thr: return it, x
        p = z + (rzn / rz) * p; rz = rzn
    return maxit, x

xs = spl.spsolve(A.tocsc(), bf)
def report(name, Minv):
    t0 = time.time(); it, x = pcg(Minv)
    print("%-58s %5d its   rel.err %.1e   (%.1fs)" % (name, it, np.linalg.norm(x - xs) / np.linalg.norm(xs), time.time() - t0), flush=True)

D0 = block_inv(A, np.arange(nb))                 # 6x6 diagonal blocks
S0 = block_inv(A, np.arange(nb) // 8)            # sibling blocks of level 0 (8 vertices)
report("block-Jacobi 6x6", lambda r: D0 @ r)

# ---- AGG=4 layout: fans 8, 4, 8, 8 ...
fans = [8, 4]
n = -(-(-(-nb // 8)) // 4)
while n > 8: fans.append(8); n = -(-n // 8)
L = hierarchy(fans)
print("levels", [l["n"] for l in L])
W = []
for k, l in enumerate(L[:-1]):
    W.append(block_inv(l["A"], np.arange(l["n"]) // fans[k + 1]))
top = spl.splu(L[-1]["A"].tocsc())
def additive(r, S_fine):
    z = S_fine @ r
    for k, l in enumerate(L[:-1]): z += l["P"] @ (W[k] @ (l["P"].T @ r))
    return z + L[-1]["P"] @ top.solve(L[-1]["P"].T @ r)
report("V0  additive ML, level-0 6x6 (the AGG=4 path)", lambda r: additive(r, D0))
report("V0s additive ML, level-0 sibling 48x48", lambda r: additive(r, S0))
lu1 = spl.splu(L[0]["A"].tocsc()); lu2 = spl.splu(L[1]["A"].tocsc())
P1, P2 = L[0]["P"], L[1]["P"]
report("V1  D0 + P1 W1 P1^T + P2 A2^-1 P2^T (exact at level 2)", lambda r: D0 @ r + P1 @ (W[0] @ (P1.T @ r)) + P2 @ lu2.solve(P2.T @ r))
report("V1s S0 + P1 W1 P1^T + P2 A2^-1 P2^T", lambda r: S0 @ r + P1 @ (W[0] @ (P1.T @ r)) + P2 @ lu2.solve(P2.T @ r))
report("V1x S0 + P2 A2^-1 P2^T (two-level, aggregates of 32)", lambda r: S0 @ r + P2 @ lu2.solve(P2.T @ r))
S0_32 = block_inv(A, np.arange(nb) // 32)
report("V1y S(32 vertices, 192x192) + P2 A2^-1 P2^T", lambda r: S0_32 @ r + P2 @ lu2.solve(P2.T @ r))
report("V2  S0 + P1 A1^-1 P1^T (two-level exact at level 1)", lambda r: S0 @ r + P1 @ lu1.solve(P1.T @ r))
report("V3  D0 + P1 A1^-1 P1^T", lambda r: D0 @ r + P1 @ lu1.solve(P1.T @ r))
# level-1 solve by one multiplicative cycle (S1 = W[0], exact A2 above)
A1 = L[0]["A"]; Q12 = None
def cyc1(r1):
    P12 = None
    y = W[0] @ r1
    res = r1 - A1 @ y
    # level-2 correction through P2 = P1 * P12  ->  restrict via least squares-free route: use P2^T directly on fine vectors
    return y, res
# P12 (level 1 -> level 2) built from centroids
cen1 = np.stack([np.bincount(np.arange(nb) // 8, t[:, k]) / np.bincount(np.arange(nb) // 8) for k in range(3)], 1)
par2 = np.arange(L[0]["n"]) // 4
cen2 = np.stack([np.bincount(par2, cen1[:, k]) / np.bincount(par2) for k in range(3)], 1)
d12 = cen1 - cen2[par2]
blk = np.tile(np.eye(6), (L[0]["n"], 1, 1)); blk[:, :3, 3:] = -skew(d12)
rows = (6 * np.arange(L[0]["n"])[:, None, None] + np.arange(6)[None, :, None]) + np.zeros((1, 1, 6), int)
cols = (6 * par2[:, None, None] + np.arange(6)[None, None, :]) + np.zeros((1, 6, 1), int)
P12 = sp.coo_matrix((blk.ravel(), (rows.ravel(), cols.ravel())), shape=(6 * L[0]["n"], 6 * L[1]["n"])).tocsr()
print("P2 == P1 P12:", abs(P1 @ P12 - P2).max())
A2 = (P12.T @ A1 @ P12).tocsc(); lu2b = spl.splu(A2)
def Y1_mult(r1):
    y = W[0] @ r1
    y = y + P12 @ lu2b.solve(P12.T @ (r1 - A1 @ y))
    return y + W[0] @ (r1 - A1 @ y)
report("V4  S0 + P1 [mult cycle(S1, A2^-1)] P1^T", lambda r: S0 @ r + P1 @ Y1_mult(P1.T @ r))
report("V4d D0 + P1 [mult cycle(S1, A2^-1)] P1^T", lambda r: D0 @ r + P1 @ Y1_mult(P1.T @ r))

# ---- exact level-2 operator by Newton-Schulz from the (scaled) additive operator of the levels >= 2
if len(L) >= 3:
    A2d = L[1]["A"].toarray()
    n2 = L[1]["n"]
    def rel_prolong(k):      # level k+1 (index k) entities -> level k+2: P_{k+1,k+2}, via least squares on the fine prolongations (exact)
        Pk, Pk1 = L[k]["P"], L[k + 1]["P"]
        return np.linalg.lstsq((Pk.T @ Pk).toarray(), (Pk.T @ Pk1).toarray(), rcond=None)[0]
    Y = np.linalg.inv(L[-1]["A"].toarray())
    for k in range(len(L) - 2, 0, -1):        # levels index k = len-2 .. 1  (level number k+1)
        Pr = rel_prolong(k)
        Y = W[k].toarray() + Pr @ Y @ Pr.T
    ev = np.linalg.eigvals(Y @ A2d).real
    print("additive Y2: eig(Y2 A2) in [%.3g, %.3g]" % (ev.min(), ev.max()))
    for omega in (ev.max() * 1.02, 3.0):
        Xk = Y / omega
        for k in range(0, 9):
            if k: Xk = 2 * Xk - Xk @ A2d @ Xk
            Xs = 0.5 * (Xk + Xk.T)
            it, _ = pcg(lambda r: D0 @ r + P1 @ (W[0] @ (P1.T @ r)) + P2 @ (Xs @ (P2.T @ r)))
            e = np.linalg.eigvals(Xk @ A2d).real
            print("omega %.2f  NS steps %d: PCG %4d its   eig(X A2) in [%.3g, %.3g]" % (omega, k, it, e.min(), e.max()), flush=True)
