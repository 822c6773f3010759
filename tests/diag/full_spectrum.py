"""Spectrum of the preconditioned operator M^-1 A on config 2's first linearisation (two-level additive: exact 48 x 48 blocks + exact coarse
solve on the rigid-body modes of 8 consecutive vertices): is there anything to deflate?   python tests/diag/full_spectrum.py"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl, scipy.linalg as la
import oracle as O, np_reference as NP
from uzliti_slam_amd import synth
g = synth.make_pose_graph(1000, 5000, seed=12345)
fl = O.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
fixed, _ = O.set_fixed_nodes(fl["fixed"], fl["ij"])
H, b, chi = NP.build_system(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"])
free_v = np.nonzero(fixed == 0)[0]; nb = len(free_v)
fidx = (6 * free_v[:, None] + np.arange(6)).ravel()
A0 = H[fidx][:, fidx].tocsr(); lam = 1e-5 * np.abs(A0.diagonal()).max()
A = (A0 + lam * sp.identity(A0.shape[0])).tocsr(); Ad = A.toarray(); n = Ad.shape[0]
X = np.asarray(fl["poses"]).reshape(-1, 3, 4)[free_v]; R, t = X[:, :, :3], X[:, :, 3]
def skew(v):
    zz = np.zeros(len(v)); return np.stack([np.stack([zz, -v[:, 2], v[:, 1]], 1), np.stack([v[:, 2], zz, -v[:, 0]], 1), np.stack([-v[:, 1], v[:, 0], zz], 1)], 1)
groups = np.arange(nb) // 8; ng = groups.max() + 1
cen = np.stack([np.bincount(groups, t[:, k], ng) / np.bincount(groups, minlength=ng) for k in range(3)], 1)
Rt = np.swapaxes(R, 1, 2); d = t - cen[groups]
blk = np.zeros((nb, 6, 6)); blk[:, :3, :3] = Rt; blk[:, :3, 3:] = -Rt @ skew(d); blk[:, 3:, 3:] = 0.5 * Rt
P = np.zeros((n, 6 * ng))
for i in range(nb): P[6 * i:6 * i + 6, 6 * groups[i]:6 * groups[i] + 6] = blk[i]
Minv = np.zeros((n, n))
for gi in range(ng):
    idx = np.arange(48 * gi, min(n, 48 * gi + 48)); Minv[np.ix_(idx, idx)] = np.linalg.inv(Ad[np.ix_(idx, idx)])
Minv += P @ np.linalg.inv(P.T @ Ad @ P) @ P.T
# eigenvalues of Minv A (similar to the symmetric L^T A L with Minv = L L^T)
L = np.linalg.cholesky(Minv)
ev = la.eigvalsh(L.T @ Ad @ L)
print("n", n, "eig(M^-1 A): min %.4f max %.4f cond %.1f" % (ev[0], ev[-1], ev[-1] / ev[0]))
print("smallest 12:", np.round(ev[:12], 4)); print("largest 8:", np.round(ev[-8:], 3))
for q in (0.05, 0.1, 0.2, 0.3, 0.5): print("eigenvalues below %.2f: %d" % (q, (ev < q).sum()))
for k in (0, 4, 8, 16, 32):
    c = ev[-1] / ev[k]; print("deflating the %2d smallest: cond %.1f -> CG bound ~ %.0f its for 1e-5" % (k, c, 0.5 * np.sqrt(c) * np.log(2e5)))

# ---- does taking the coarse modes OUT of the smoother blocks (they are solved exactly on the coarse level) remove the double counting?
def pcg_dense(Mi, tol=1e-5, maxit=500):
    bf = b[fidx]; x = np.zeros(n); r = bf.copy(); z = Mi @ r; p = z.copy(); rz = r @ z; thr = tol * tol * rz
    for it in range(1, maxit + 1):
        Ap = Ad @ p; a = rz / (p @ Ap); x += a * p; r -= a * Ap; z = Mi @ r; rzn = r @ z
        if not rzn > thr: return it
        p = z + (rzn / rz) * p; rz = rzn
    return maxit
S = np.zeros((n, n))
for gi in range(ng):
    idx = np.arange(48 * gi, min(n, 48 * gi + 48)); S[np.ix_(idx, idx)] = np.linalg.inv(Ad[np.ix_(idx, idx)])
C = P @ np.linalg.inv(P.T @ Ad @ P) @ P.T
print("additive S + C: %d its" % pcg_dense(S + C))
Pi2 = np.eye(n) - P @ np.linalg.inv(P.T @ P) @ P.T                       # l2 projection off the coarse modes (block diagonal per aggregate)
for name, Sm in (("Pi S Pi^T (l2 projection)", Pi2 @ S @ Pi2.T),):
    Mi = Sm + C; ev2 = la.eigvalsh(np.linalg.cholesky(Mi + 1e-12 * np.eye(n)).T @ Ad @ np.linalg.cholesky(Mi + 1e-12 * np.eye(n)))
    print("%-34s + C: %d its, eig %.4f .. %.3f cond %.1f" % (name, pcg_dense(Mi), ev2[0], ev2[-1], ev2[-1] / ev2[0]))
# A-orthogonal version: S' = S - S A P (P^T A S A P)^-1 P^T A S  (the block's solve with the coarse modes constrained out, per aggregate)
Sa = np.zeros((n, n))
for gi in range(ng):
    idx = np.arange(48 * gi, min(n, 48 * gi + 48)); Pk = P[idx][:, 6 * gi:6 * gi + 6]; Ak = Ad[np.ix_(idx, idx)]; Sk = np.linalg.inv(Ak)
    # minimise over block corrections A-orthogonal to the block's own rigid modes: Sk - Pk (Pk^T Ak Pk)^-1 Pk^T
    Sa[np.ix_(idx, idx)] = Sk - Pk @ np.linalg.inv(Pk.T @ Ak @ Pk) @ Pk.T
Mi = Sa + C; w = la.eigvalsh(Mi); print("min eig of M^-1 (must be > 0):", w[0])
ev3 = la.eigvalsh(np.linalg.cholesky(Mi).T @ Ad @ np.linalg.cholesky(Mi)) if w[0] > 0 else None
print("S_k - P_k (P_k^T A_k P_k)^-1 P_k^T     + C: %d its" % pcg_dense(Mi), ("eig %.4f .. %.3f cond %.1f" % (ev3[0], ev3[-1], ev3[-1] / ev3[0])) if ev3 is not None else "")

# ---- smoothed aggregation: P_s = (I - w S A) P (a denser prolongation; the coarse operator is dense anyway) - what would it buy?
for w_ in (0.5, 0.67, 1.0):
    Ps = P - w_ * (S @ (Ad @ P))
    Cs = Ps @ np.linalg.inv(Ps.T @ Ad @ Ps) @ Ps.T
    print("smoothed aggregation, omega %.2f: additive S + C_s %d its" % (w_, pcg_dense(S + Cs)))
