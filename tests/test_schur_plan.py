"""CPU tests (no GPU) of the Schur plan behind uzl_pgo_cfg::schur_reduce (csrc/pgo_schur.hpp): which vertices of a block system are
chain interiors, how they are grouped into runs, and the block structure of the reduced system.  Checked against a dense Schur
complement of a random SPD matrix with the same block structure (numpy)."""
import numpy as np
import pytest


def _csr(n, edges, fixed=()):
    """block-CSR over the free vertices as uzl_pgo builds it: one slot per (free endpoint, edge), col = -1 for a fixed neighbour"""
    free = [v for v in range(n) if v not in set(fixed)]
    v2b = {v: i for i, v in enumerate(free)}
    rows = [[] for _ in free]
    for a, b in edges:
        if a in v2b:
            rows[v2b[a]].append(v2b.get(b, -1))
        if b in v2b:
            rows[v2b[b]].append(v2b.get(a, -1))
    rp = np.zeros(len(free) + 1, np.int32)
    for i, r in enumerate(rows):
        rp[i + 1] = rp[i] + len(r)
    col = np.array([c for r in rows for c in r], np.int32)
    return rp, col


def _check(capi, rp, col, cap):
    nb = len(rp) - 1
    P = capi.schur_plan(rp, col, cap)
    red, rid, pos = P["red_row"], P["run_id"], P["run_pos"]
    interior = red < 0
    assert (interior == (rid >= 0)).all() and P["n_reduced"] == int((~interior).sum())
    assert sorted(red[~interior]) == list(range(P["n_reduced"])) and (np.diff(red[~interior]) > 0).all()      # separators keep their order
    deg = np.diff(rp)
    for a in np.nonzero(interior)[0]:
        cs = col[rp[a]:rp[a + 1]]
        assert 1 <= deg[a] <= 2 and not (deg[a] == 2 and cs[0] == cs[1] and cs[0] >= 0)
    # runs: consecutive positions are neighbours, no run longer than cap, interiors of different runs never touch
    for r in range(P["n_runs"]):
        mem = np.nonzero(rid == r)[0]
        mem = mem[np.argsort(pos[mem])]
        assert 1 <= len(mem) <= cap and list(pos[mem]) == list(range(len(mem)))
        for u, v in zip(mem[:-1], mem[1:]):
            assert v in col[rp[u]:rp[u + 1]]
    for a in np.nonzero(interior)[0]:
        for c in col[rp[a]:rp[a + 1]]:
            assert c < 0 or not interior[c] or rid[c] == rid[a]
    # dense check: Schur complement of a random SPD matrix with this structure has exactly the planned off-diagonal pattern
    rng = np.random.default_rng(nb)
    A = np.zeros((nb, nb))
    for a in range(nb):
        for c in col[rp[a]:rp[a + 1]]:
            if c > a:
                w = rng.uniform(0.5, 1.5)
                A[a, c] -= w; A[c, a] -= w
    A += np.diag(-A.sum(1) + rng.uniform(0.1, 1.0, nb))                      # diagonally dominant: SPD
    I = np.nonzero(interior)[0]; S = np.nonzero(~interior)[0]
    if len(S) and len(I):
        Sc = A[np.ix_(S, S)] - A[np.ix_(S, I)] @ np.linalg.solve(A[np.ix_(I, I)], A[np.ix_(I, S)])
        want = {(i, j) for i in range(len(S)) for j in range(len(S)) if i != j and abs(Sc[i, j]) > 1e-12}
        got = {(i, int(j)) for i in range(len(S)) for j in P["col"][P["row_ptr"][i]:P["row_ptr"][i + 1]]}
        assert want == got
    return P


def test_chain_with_loop_closures(capi):
    n = 400
    edges = [(i, i + 1) for i in range(n - 1)] + [(10, 200), (11, 201), (50, 300), (120, 380), (121, 381)]
    rp, col = _csr(n, edges, fixed=[0])
    P = _check(capi, rp, col, cap=24)
    assert P["n_reduced"] < 40 and (P["red_row"] < 0).sum() > 350


@pytest.mark.parametrize("cap", [1, 3, 8, 24])
def test_long_runs_are_cut(capi, cap):
    n = 130
    rp, col = _csr(n, [(i, i + 1) for i in range(n - 1)], fixed=[0])
    P = _check(capi, rp, col, cap)
    assert P["n_reduced"] == (n - 1) // (cap + 1)                           # every (cap+1)-th vertex of the chain stays


def test_ring_leaf_double_edge_and_isolated(capi):
    # a ring without any fixed vertex or separator on it (gauge elsewhere), a tree with leaves, a double edge, a vertex between two fixed ones
    ring = [(i, (i + 1) % 20) for i in range(20)]
    tree = [(20, 21), (21, 22), (21, 23), (23, 24), (24, 25)]
    dbl = [(30, 31), (30, 31), (31, 32), (32, 33)]
    iso = [(40, 41), (41, 42)]
    rp, col = _csr(43, ring + tree + dbl + iso, fixed=[26, 27, 28, 29, 34, 35, 36, 37, 38, 39, 40, 42])
    P = _check(capi, rp, col, cap=6)
    red = P["red_row"]
    free = [v for v in range(43) if v not in (26, 27, 28, 29, 34, 35, 36, 37, 38, 39, 40, 42)]
    b = {v: i for i, v in enumerate(free)}
    assert (red[[b[v] for v in range(20)]] >= 0).sum() >= 2                  # the ring was opened and cut
    assert red[b[21]] >= 0 and red[b[22]] < 0 and red[b[25]] < 0             # hub stays, leaves go
    assert red[b[30]] >= 0 and red[b[31]] >= 0                               # both ends of a double edge stay
    assert red[b[41]] < 0                                                    # between two fixed vertices: a run with no separator at either end


def test_random_graphs(capi):
    rng = np.random.default_rng(3)
    for trial in range(30):
        n = int(rng.integers(5, 300))
        edges = [(i, i + 1) for i in range(n - 1) if rng.random() < 0.95]
        for _ in range(int(rng.integers(0, n // 4 + 1))):
            a, b = rng.integers(0, n, 2)
            if a != b:
                edges.append((int(a), int(b)))
        fixed = [int(v) for v in rng.integers(0, n, int(rng.integers(1, 4)))]
        rp, col = _csr(n, edges, fixed)
        # rows without any slot cannot occur in uzl_pgo (gauge fixing); drop such graphs
        if (np.diff(rp) == 0).any():
            continue
        _check(capi, rp, col, cap=int(rng.integers(1, 12)))


# ---- strong-aggregate numbering of the reduced system (SchurPlan::strong; csrc/pgo_schur.hpp)
def _strong_case(seed, n=4000, closures=260):
    """an odometry chain with loop closures; closure edges are 50x stiffer than odometry, a few odometry edges are very soft"""
    rng = np.random.default_rng(seed)
    edges = [(i, i + 1) for i in range(n - 1)]
    w = list(np.where(rng.random(n - 1) < 0.02, 0.05, 1.0))
    for _ in range(closures):
        a, b = sorted(rng.choice(n, 2, replace=False))
        if b - a > 3:
            edges.append((int(a), int(b))); w.append(50.0)
    return n, edges, np.array(w)


def _slot_weights(n, edges, w, fixed=()):
    free = [v for v in range(n) if v not in set(fixed)]
    v2b = {v: i for i, v in enumerate(free)}
    rows = [[] for _ in free]
    for k, (a, b) in enumerate(edges):
        if a in v2b:
            rows[v2b[a]].append(w[k])
        if b in v2b:
            rows[v2b[b]].append(w[k])
    return np.array([x for r in rows for x in r], np.float64)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_strong_numbering_is_a_padded_permutation(capi, seed):
    n, edges, w = _strong_case(seed)
    rp, col = _csr(n, edges, fixed=(0,))
    sw = _slot_weights(n, edges, w, fixed=(0,))
    plain = capi.schur_plan(rp, col, 24)
    P = capi.schur_plan_strong(rp, col, sw, 24, strong_min=1, theta=0.25)
    Q = capi.schur_plan_strong(rp, col, sw, 24, strong_min=1, theta=0.25)
    assert np.array_equal(P["red_row"], Q["red_row"]) and np.array_equal(P["sep_rows"], Q["sep_rows"])         # deterministic
    red, sep = P["red_row"], P["sep_rows"]
    # the same separators as the plain plan, renumbered; 32 rows per block, 8 per group
    assert np.array_equal(red >= 0, plain["red_row"] >= 0) and P["n_sep"] == plain["n_reduced"]
    assert P["n_reduced"] == 32 * P["n_blocks"] == len(sep) and P["n_groups"] <= 4 * P["n_blocks"]
    real = np.nonzero(sep >= 0)[0]
    assert len(real) == P["n_sep"] and np.array_equal(red[sep[real]], real) and len(set(sep[real])) == len(real)
    occupied_groups = 0
    for g in range(len(sep) // 8):
        rows = sep[8 * g: 8 * g + 8]
        k = int((rows >= 0).sum())
        assert (rows[:k] >= 0).all() and (rows[k:] < 0).all()            # a group's rows come first, the padding behind them
        assert (np.diff(rows[:k]) > 0).all()                             # ... in row order
        occupied_groups += k > 0
        if k == 0:
            assert g % 4 != 0                                             # a block starts with an occupied group
    assert occupied_groups == P["n_groups"]
    for b in range(P["n_blocks"]):
        occ = [(sep[32 * b + 8 * j] >= 0) for j in range(4)]
        assert occ[0] and occ == sorted(occ, reverse=True)               # occupied groups first
    # below the threshold (or without weights) the plan is the plain one
    Z = capi.schur_plan_strong(rp, col, sw, 24, strong_min=10 ** 6, theta=0.25)
    assert Z["n_groups"] == 0 and np.array_equal(Z["red_row"], plain["red_row"])


def test_strong_numbering_groups_what_is_stiffly_coupled(capi):
    """Two separators tied by a stiff loop closure share a group of 8; the two ends of a soft odometry edge do not (unless something
    stiffer ties them another way round)."""
    n, edges, w = _strong_case(7)
    rp, col = _csr(n, edges, fixed=(0,))
    sw = _slot_weights(n, edges, w, fixed=(0,))
    P = capi.schur_plan_strong(rp, col, sw, 24, strong_min=1, theta=0.25)
    red = P["red_row"]
    v2b = {v: i for i, v in enumerate(range(1, n))}
    deg = np.zeros(n, int)
    for a, b in edges:
        deg[a] += 1; deg[b] += 1
    same = tot = 0
    for (a, b), wk in zip(edges, w):
        if wk == 50.0 and a in v2b and b in v2b and deg[a] == 3 and deg[b] == 3:      # a closure between two plain chain vertices
            ra, rb = red[v2b[a]], red[v2b[b]]
            assert ra >= 0 and rb >= 0
            tot += 1; same += (ra // 8 == rb // 8)
    assert tot > 50 and same >= 0.9 * tot, (same, tot)
    soft_same = soft_tot = 0
    for (a, b), wk in zip(edges, w):
        if wk == 0.05 and a in v2b and b in v2b:
            ra, rb = red[v2b[a]], red[v2b[b]]
            if ra >= 0 and rb >= 0:
                soft_tot += 1; soft_same += (ra // 8 == rb // 8)
    assert soft_tot == 0 or soft_same <= 0.2 * soft_tot, (soft_same, soft_tot)


def test_strong_numbering_one_level_layout(capi):
    """Up to `one_level_max` groups the strong aggregates are laid out as blocks of ONE group (8 rows): the level-1 path's geometry."""
    n, edges, w = _strong_case(5, n=1500, closures=90)
    rp, col = _csr(n, edges, fixed=(0,))
    sw = _slot_weights(n, edges, w, fixed=(0,))
    two = capi.schur_plan_strong(rp, col, sw, 24, strong_min=1, theta=0.25, one_level_max=0)
    one = capi.schur_plan_strong(rp, col, sw, 24, strong_min=1, theta=0.25, one_level_max=10 ** 6)
    assert one["n_groups"] == two["n_groups"] and one["n_blocks"] == 0 and two["n_blocks"] > 0
    assert one["n_reduced"] == 8 * one["n_groups"] and two["n_reduced"] == 32 * two["n_blocks"]
    sep = one["sep_rows"]
    for g in range(one["n_groups"]):
        rows = sep[8 * g: 8 * g + 8]
        k = int((rows >= 0).sum())
        assert k >= 1 and (rows[:k] >= 0).all() and (rows[k:] < 0).all() and (np.diff(rows[:k]) > 0).all()

    def groups(P):
        out = set()
        s_ = P["sep_rows"]
        for g in range(len(s_) // 8):
            m = tuple(int(x) for x in s_[8 * g: 8 * g + 8] if x >= 0)
            if m:
                out.add(m)
        return out
    assert groups(one) == groups(two)                     # the same groups either way
