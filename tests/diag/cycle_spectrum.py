"""numpy prototype: spectrum of the multiplicative cycle of the composite operator on a given graph / lambda.
   python tests/diag/cycle_spectrum.py N E lambda_scale    (large-graph layout: fans 8, 4, 8, 8 ...; cycle at level 2)
Prints eig(S A_2) of the sibling-block smoother, eig(Y_3 A_3) of the additive coarse operator, and eig(X0 A_2) of the cycle with
the additive and with the exact coarse solve, then what Newton-Schulz makes of each."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, scipy.sparse as sp
import oracle as O, np_reference as NP
from uzliti_slam_amd import synth

N, E = int(sys.argv[1]), int(sys.argv[2])
g = synth.make_pose_graph(N, E, seed=12345)
fl = O.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
fixed, _ = O.set_fixed_nodes(fl["fixed"], fl["ij"])
H, b, chi = NP.build_system(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"])
free_v = np.nonzero(fixed == 0)[0]
fidx = (6 * free_v[:, None] + np.arange(6)).ravel()
A0 = H[fidx][:, fidx].tocsr()
lam = 1e-5 * np.abs(A0.diagonal()).max() * (float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
A = (A0 + lam * sp.identity(A0.shape[0])).tocsr()
nb = len(free_v)
X = fl["poses"].reshape(-1, 3, 4)[free_v]
R, t = X[:, :, :3], X[:, :, 3]
agg1 = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # 1: small-graph layout (fans 8, 8, 8: cycle at level 1)


def skew(v):
    z = np.zeros(len(v))
    return np.stack([np.stack([z, -v[:, 2], v[:, 1]], 1), np.stack([v[:, 2], z, -v[:, 0]], 1), np.stack([-v[:, 1], v[:, 0], z], 1)], 1)


def prolong0(groups, centers):
    Rt = np.swapaxes(R, 1, 2)
    d = t - centers[groups]
    blk = np.zeros((nb, 6, 6))
    blk[:, :3, :3] = Rt; blk[:, :3, 3:] = -Rt @ skew(d); blk[:, 3:, 3:] = 0.5 * Rt
    rows = (6 * np.arange(nb)[:, None, None] + np.arange(6)[None, :, None]) + np.zeros((1, 1, 6), int)
    cols = (6 * groups[:, None, None] + np.arange(6)[None, None, :]) + np.zeros((1, 6, 1), int)
    return sp.coo_matrix((blk.ravel(), (rows.ravel(), cols.ravel())), shape=(6 * nb, 6 * (groups.max() + 1))).tocsr()


def prolong_rel(par, cen_f, cen_c):
    n = len(par)
    d = cen_f - cen_c[par]
    blk = np.tile(np.eye(6), (n, 1, 1)); blk[:, :3, 3:] = -skew(d)
    rows = (6 * np.arange(n)[:, None, None] + np.arange(6)[None, :, None]) + np.zeros((1, 1, 6), int)
    cols = (6 * par[:, None, None] + np.arange(6)[None, None, :]) + np.zeros((1, 6, 1), int)
    return sp.coo_matrix((blk.ravel(), (rows.ravel(), cols.ravel())), shape=(6 * n, 6 * (par.max() + 1))).tocsr()


def block_inv(Al, member):
    Ad = Al.toarray() if sp.issparse(Al) else Al
    out = np.zeros_like(Ad)
    for gid in range(member.max() + 1):
        ent = np.nonzero(member == gid)[0]
        idx = (6 * ent[:, None] + np.arange(6)).ravel()
        out[np.ix_(idx, idx)] = np.linalg.inv(Ad[np.ix_(idx, idx)])
    return out


fans = [8, 8] if agg1 else [8, 4]
n = nb
for f in fans: n = -(-n // f)
while n > 8: fans.append(8); n = -(-n // 8)
cl = 1 if agg1 else 2
# level matrices and relative prolongations
par = np.arange(nb) // fans[0]
cen = np.stack([np.bincount(par, t[:, k]) / np.bincount(par) for k in range(3)], 1)
P01 = prolong0(par, cen)
As = [A, (P01.T @ A @ P01).toarray()]
Ps = [P01]
cens = [t, cen]
for f in fans[1:]:
    nl = len(cens[-1]); par = np.arange(nl) // f
    c2 = np.stack([np.bincount(par, cens[-1][:, k]) / np.bincount(par) for k in range(3)], 1)
    Pr = prolong_rel(par, cens[-1], c2).toarray()
    Ps.append(Pr); As.append(Pr.T @ As[-1] @ Pr); cens.append(c2)
Lv = len(fans)
print("levels n:", [nb] + [len(c) for c in cens[1:]], "fans", fans, "lambda %.3g" % lam, "cycle level", cl)


def eig_range(M):
    e = np.linalg.eigvals(M).real
    return e.min(), e.max()


Acl = As[cl]; ncl = len(cens[cl])
S = block_inv(Acl, np.arange(ncl) // fans[cl])
print("smoother: eig(S A_cl) in [%.4g, %.4g]" % eig_range(S @ Acl))
# additive operator of the levels above cl (what the GPU builds), and the exact one
Y = np.linalg.inv(As[Lv])
for l in range(Lv - 1, cl, -1):
    W = block_inv(As[l], np.arange(len(cens[l])) // fans[l])
    Y = W + Ps[l] @ Y @ Ps[l].T
Ac = As[cl + 1]
print("coarse additive: eig(Y A_c) in [%.4g, %.4g]" % eig_range(Y @ Ac))
Pc = Ps[cl]
for name, Yc in (("additive coarse", Y), ("exact coarse", np.linalg.inv(Ac))):
    Q = Pc - S @ Acl @ Pc
    X0 = 2 * S - S @ Acl @ S + Q @ Yc @ Q.T
    lo, hi = eig_range(X0 @ Acl)
    print("%-16s cycle: eig(X0 A_cl) in [%.4g, %.4g]" % (name, lo, hi))
    Xk = X0
    for k in range(1, 4):
        Xk = 2 * Xk - Xk @ Acl @ Xk
        lo, hi = eig_range(Xk @ Acl)
        print("    NS step %d: eig in [%.4g, %.4g]" % (k, lo, hi))
