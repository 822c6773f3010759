"""Online SLAM loop over the whole path (BASELINE config 5): node-pair match jobs feed a growing pose graph
that is re-optimised every `reopt_edges` new edges.

What the reference does per new node (graph_slam/src/graph_slam_node.cpp): the candidate producers hand node pairs to
`TransformationEstimator::estimateEdge` (:266,:284,:518); every estimate comes back through `newEdgeCallback` (:779-829: duplicate
check, score, transform size, `checkEdgeHeuristic`); a timer calls `GraphOptimizer::optimize` (:1138-1150), whose `addGraphImpl`
(graph_optimization/src/g2o_optimizer.cpp:55-104) re-adds every non-odometry edge to the `TransformationFilter`, takes its
`validEdges()` and rebuilds the problem from scratch, then `storeOptimizationResults` writes the poses back (:1252-1254).

Here every stage runs on the GPU behind the C ABI (capi.Match / Gate / Filter / Pgo); this module is the host-side driver and holds no
arithmetic of its own beyond dead reckoning of new nodes (pose_prev * odometry, as `GraphSlamNode::addNode` places a new node).
The reference's loop is driven by wall-clock timers and thread scheduling, i.e. not reproducible; this driver fixes the schedule so
that the result is a function of the inputs only (and in particular independent of the match batch size and of the number of
ranks the match jobs are sharded over):

  * nodes enter in index order; node j brings its odometry edge (j-1 -> j) and the node pairs whose later node is j, in pair order;
  * the graph is re-optimised at the first node boundary at which >= `reopt_edges` edges (odometry + accepted feature edges) have
    been added since the last solve, and once more at the end;
  * between two solves the poses are the last solve's (new nodes dead-reckoned from them); the acceptance gate sees the graph up to
    `lookahead` nodes past the last solved node - one gate call per solve interval, cut at the trigger (the gate replays
    newEdgeCallback's sequential semantics, so the verdicts of a prefix do not depend on what follows it).

Multi-GPU: pair jobs are independent units - rank r matches the contiguous shard `dist.shard_range` gives it of every batch and the
results are gathered in job order; gate, filter and solver run on rank 0 (`solve_rank`), whose solve overlaps the next batch's
matching (launched before the solve, collected after).  No data-path collective.
"""
import time

import numpy as np

from . import capi, synth
from . import dist as ud

I12 = np.eye(3, 4).reshape(12)


class OnlineStats(dict):
    pass


def _se3_chain(pose_prev, odo):
    """poses of consecutive new nodes: pose_prev * odo[0], that * odo[1], ...  (k,3,4).  A prefix product by doubling (log2 k batched
    3 x 3 products instead of k small ones in a Python loop, which was most of this driver's own time: 1 ms per interval)."""
    k = len(odo)
    if k == 0:
        return np.empty((0, 3, 4))
    A = np.asarray(odo, np.float64).reshape(k, 3, 4)
    Rm = A[:, :, :3].copy(); t = A[:, :, 3].copy()
    d = 1
    while d < k:                                    # after the step, entry i holds odo[i - 2d + 1 .. i] multiplied in order
        Rn = Rm.copy(); tn = t.copy()
        Rn[d:] = Rm[:-d] @ Rm[d:]
        tn[d:] = np.einsum("nij,nj->ni", Rm[:-d], t[d:]) + t[:-d]
        Rm, t = Rn, tn
        d *= 2
    P = np.asarray(pose_prev, np.float64).reshape(3, 4)
    out = np.empty((k, 3, 4))
    out[:, :, :3] = P[:, :3] @ Rm
    out[:, :, 3] = t @ P[:, :3].T + P[:, 3]
    return out


class OnlineSlam:
    """State of one online run.  `run` = synth.make_online_run(...) (or any dict of the same layout)."""

    def __init__(self, run, device=0, reopt_edges=256, lm_iterations=20, lookahead=None, match_cfg=None, gate_cfg=None,
                 filter_cfg=None, pgo_cfg=None, match_batch=512, rank=0, world=1, tdist=None, solve_rank=0, log=None, incremental=True):
        self.run = run
        self.N = len(run["fixed"])
        self.P = len(run["pair_from"])
        self.reopt_edges = int(reopt_edges)
        self.lookahead = int(lookahead if lookahead is not None else reopt_edges)   # one odometry edge per node: a trigger falls inside
        self.lm_iterations = int(lm_iterations)
        self.match_batch = int(match_batch)
        self.rank, self.world, self.tdist, self.solve_rank = rank, world, tdist, solve_rank
        self.is_solver = rank == solve_rank
        self.log = log
        mc = dict(ransac_threshold=0.1, ransac_iteration=500, ransac_break_percentage=0.6, do_prosac=1, seed=777)
        mc.update(match_cfg or {})
        self.gate = self.filt = self.pgo = None
        self._open_handles(device, mc, gate_cfg, filter_cfg, pgo_cfg)
        if self.is_solver:
            self.filt.set_sensors(I12.reshape(1, 12))
        o = run["odo"]
        self.odo_T = np.asarray(o["transform"], np.float64).reshape(-1, 3, 4)[: self.N - 1]
        self.odo_info = np.asarray(o["information"], np.float64).reshape(-1, 36)[: self.N - 1]
        self.poses = np.zeros((self.N, 3, 4)); self.poses[0] = np.asarray(run["init"]).reshape(-1, 3, 4)[0]
        self.cur = 1                                  # nodes [0, cur) are in the graph and solved
        self.edges_since = 0
        # accepted feature edges, in acceptance order
        self.f_key = np.zeros(0, np.int64); self.f_from = np.zeros(0, np.int32); self.f_to = np.zeros(0, np.int32)
        self.f_score = np.zeros(0); self.f_T = np.zeros((0, 12)); self.f_info = np.zeros((0, 36)); self.f_sticky = np.zeros(0, bool)
        self.results = np.zeros(self.P, capi.EDGE_RESULT_DTYPE)
        self.have = 0                                 # results of pairs [0, have) are in
        self._next_batch = 0                          # first pair of the next batch to launch
        self._inflight = None
        self.solves = []
        self.t = dict(match_wait=0.0, gate=0.0, gate_set_graph=0.0, filter=0.0, add_graph=0.0, optimize=0.0, store=0.0, host=0.0, upload=0.0)
        self.accept_log = []                          # (pair index, accepted) in gate order
        self.keep_poses_per_solve = 0                 # diagnostics / bench: keep a copy of the solved poses of the first k re-optimisations
        self.poses_at_solve = []
        self._t_first = None                          # start of the first interval: every solve records the wall clock since then
        self.incremental = bool(incremental)          # grow the solver's resident graph (uzl_pgo_append_graph) instead of re-sending all of it
        self._pgo_nodes = 0; self._pgo_feats = 0; self._pgo_edges = 0; self._f_in_idx = np.zeros(0, np.int32); self._last = None
        # the graph the gate sees, in the order its edges came into being (the reference's edge ids are time-ordered, graph_slam_node.cpp:294):
        # the odometry edges and accepted feature edges of every interval behind those of the interval before.  Kept as one growing array -
        # every interval's list repeats the last one's and adds a tail, which is what uzl_gate_set_graph recognises.
        self._ge = np.zeros(1024, capi.GATE_EDGE_DTYPE); self._ge["transform"] = I12
        self._ge_n = 0; self._ge_nodes = 1; self._ge_feats = 0; self._ge_fidx = np.zeros(0, np.int64)

    def _open_handles(self, device, mc, gate_cfg, filter_cfg, pgo_cfg):
        """The four C-ABI handles of the path (estimator on every rank; gate, filter and solver on the solver rank)."""
        self.matcher = capi.Match(device=device, **mc)
        if self.is_solver:
            self.gate = capi.Gate(device=device, **(gate_cfg or {}))
            fc = dict(seed=mc["seed"]); fc.update(filter_cfg or {})
            self.filt = capi.Filter(device=device, **fc)
            self.pgo = capi.Pgo(device=device, iterations=self.lm_iterations, **(pgo_cfg or {}))

    # ------------------------------------------------------------------ frames + matching
    def upload_frames(self):
        """FeatureData of this rank's pairs -> HBM (once; outside any timed region of the bench)."""
        t0 = time.perf_counter()
        self.fid = {}
        mine = []
        for b0 in range(0, self.P, self.match_batch):
            b1 = min(self.P, b0 + self.match_batch)
            lo, hi = ud.shard_range(b1 - b0, self.rank, self.world)
            mine += list(range(b0 + lo, b0 + hi))
        if hasattr(self.matcher, "add_frames") and hasattr(capi.Match, "pack_frames") and mine:
            # one uzl_match_add_frames call for all of this rank's frames (what the adapter's batching worker does)
            packed = capi.Match.pack_frames([(x["desc"], x["pos"], x["valid"]) for k in mine for x in self.run["frames"][k]])
            flat = self.matcher.add_frames(packed)
            for q, k in enumerate(mine):
                self.fid[k] = (int(flat[2 * q]), int(flat[2 * q + 1]))
        else:
            for k in mine:
                f, t = self.run["frames"][k]
                self.fid[k] = (self.matcher.add_frame(f["desc"], f["pos"], f["valid"]), self.matcher.add_frame(t["desc"], t["pos"], t["valid"]))
        self.t["upload"] = time.perf_counter() - t0

    def _launch_next(self):
        if self._inflight is not None or self._next_batch >= self.P:
            return
        b0 = self._next_batch; b1 = min(self.P, b0 + self.match_batch)
        lo, hi = ud.shard_range(b1 - b0, self.rank, self.world)
        mine = list(range(b0 + lo, b0 + hi))
        if mine:
            jobs, fids = capi.Match._jobs([self.fid[k] for k in mine], mine)     # job id = global pair index: keys the sampling stream
            self.matcher.launch_raw(jobs, fids)
        self._inflight = (b0, b1, mine)
        self._next_batch = b1

    def _collect(self):
        b0, b1, mine = self._inflight
        t0 = time.perf_counter()
        local = self.matcher.collect(np.zeros(len(mine), capi.EDGE_RESULT_DTYPE)) if mine else np.zeros(0, capi.EDGE_RESULT_DTYPE)
        self.t["match_wait"] += time.perf_counter() - t0
        allr = ud.gather_edge_results(local, b1 - b0, self.rank, self.world, self.tdist)
        self.results[b0:b1] = allr
        self.have = b1
        self._inflight = None

    def _need_results(self, upto_pair):
        """block until the results of pairs [0, upto_pair) are in; keeps one batch in flight behind them"""
        while self.have < upto_pair:
            if self._inflight is None:
                self._launch_next()
            self._collect()
            self._launch_next()

    # ------------------------------------------------------------------ one solve interval
    def _graph_edges(self, n_nodes):
        """SlamEdge arrays of the current graph: odometry chain up to n_nodes + the accepted feature edges between those nodes."""
        no = n_nodes - 1
        nf = len(self.f_key)
        E = no + nf
        e = {"from": np.concatenate([np.arange(no, dtype=np.int32), self.f_from]),
             "to": np.concatenate([np.arange(1, no + 1, dtype=np.int32), self.f_to]),
             "type": np.concatenate([np.full(no, synth.EDGE_TYPE_ODOM, np.int32), np.full(nf, synth.EDGE_TYPE_3D_FULL, np.int32)]),
             "sensor_from": np.full(E, -1, np.int32), "sensor_to": np.full(E, -1, np.int32),
             "valid": np.concatenate([np.ones(no, np.int32), self.f_sticky.astype(np.int32)]),
             "transform": np.concatenate([self.odo_T[:no].reshape(no, 12), self.f_T]),
             "displacement_from": np.tile(I12, (E, 1)), "displacement_to": np.tile(I12, (E, 1)),
             "information": np.concatenate([self.odo_info[:no], self.f_info]),
             "diff_time": np.concatenate([np.full(no, 0.5), np.zeros(nf)])}
        return e

    def _gate_graph(self, hi, nf):
        """The gate's edge list for nodes [0, hi) and the first nf accepted feature edges: last interval's list, the `valid` flags of its
        feature edges as they are now, then the new odometry edges (node i-1 -> i) and the feature edges accepted since."""
        no = hi - self._ge_nodes; nfn = nf - self._ge_feats
        need = self._ge_n + no + nfn
        if need > len(self._ge):
            grown = np.zeros(max(need, 2 * len(self._ge)), capi.GATE_EDGE_DTYPE); grown["transform"] = I12
            grown[:self._ge_n] = self._ge[:self._ge_n]
            self._ge = grown
        g = self._ge
        if self._ge_feats:
            g["valid"][self._ge_fidx] = self.f_sticky[:self._ge_feats]
        a = self._ge_n
        g["from"][a:a + no] = np.arange(self._ge_nodes - 1, hi - 1); g["to"][a:a + no] = np.arange(self._ge_nodes, hi)
        g["type"][a:a + no] = synth.EDGE_TYPE_ODOM; g["valid"][a:a + no] = 1
        b = a + no
        g["from"][b:b + nfn] = self.f_from[self._ge_feats:nf]; g["to"][b:b + nfn] = self.f_to[self._ge_feats:nf]
        g["type"][b:b + nfn] = synth.EDGE_TYPE_3D_FULL; g["valid"][b:b + nfn] = self.f_sticky[self._ge_feats:nf]
        self._ge_fidx = np.concatenate([self._ge_fidx, np.arange(b, b + nfn, dtype=np.int64)])
        self._ge_n = need; self._ge_nodes = hi; self._ge_feats = nf
        return g[:need]

    @property
    def last_input(self):
        """(poses, fixed, edges) of the last re-optimisation as addGraphImpl gets them: the full arrays (tests, diagnostics)."""
        n_nodes, poses, in_solve = self._last
        nf = len(in_solve)
        keep = (self.f_key, self.f_from, self.f_to, self.f_T, self.f_info, self.f_sticky)
        try:        # (feature edges accepted after that solve are not part of it)
            self.f_key, self.f_from, self.f_to, self.f_T, self.f_info, self.f_sticky = (x[:nf] for x in keep)
            e = self._graph_edges(n_nodes)
        finally:
            self.f_key, self.f_from, self.f_to, self.f_T, self.f_info, self.f_sticky = keep
        e["valid"][n_nodes - 1:] = in_solve                                         # only validEdges() enter the solve (:98-103)
        return poses, self.run["fixed"][:n_nodes], e

    def step(self):
        """One solve interval: admit nodes up to the next trigger, gate their candidates, filter, re-optimise.  Returns False at the end."""
        if self.cur >= self.N:
            return False
        if self._t_first is None:
            self._t_first = time.perf_counter()
        run = self.run
        hi = min(self.N, self.cur + self.lookahead)
        later = run["pair_later"]
        p_lo = int(np.searchsorted(later, self.cur, side="left")); p_hi = int(np.searchsorted(later, hi, side="left"))
        self._need_results(p_hi)
        t_host = time.perf_counter()
        self.poses[self.cur:hi] = _se3_chain(self.poses[self.cur - 1], self.odo_T[self.cur - 1:hi - 1])    # dead reckoning of the new nodes
        res = self.results[p_lo:p_hi]
        ok = np.nonzero(res["ok"] != 0)[0]
        cand_pair = p_lo + ok
        # ---- acceptance gate (newEdgeCallback) over the graph up to `hi`
        t0 = time.perf_counter()
        nf = len(self.f_key)
        self.gate.set_graph(self.poses[:hi].reshape(-1, 12), self._gate_graph(hi, nf))
        self.t["gate_set_graph"] += time.perf_counter() - t0
        if len(cand_pair):
            cands = capi.gate_edges(run["pair_from"][cand_pair], run["pair_to"][cand_pair], np.ones(len(cand_pair), int),
                                    score=res["consensus"][ok].astype(np.float64), transform=res["T"][ok])
            acc, val, _ = self.gate.check(cands, want_dist=False)
        else:
            acc = np.zeros(0, np.uint8); val = np.zeros(0, np.uint8)
        t_gate = time.perf_counter() - t0
        self.t["gate"] += t_gate
        # ---- trigger: first node boundary with >= reopt_edges new edges
        per_node = np.ones(hi - self.cur, np.int64)
        np.add.at(per_node, later[cand_pair[acc != 0]] - self.cur, 1)
        cum = self.edges_since + np.cumsum(per_node)
        trig = np.nonzero(cum >= self.reopt_edges)[0]
        last = self.cur + (int(trig[0]) if len(trig) else hi - self.cur - 1)        # last admitted node
        final = (last == self.N - 1)
        if not len(trig) and not final:
            # no trigger inside the lookahead (cannot happen with lookahead >= reopt_edges): admit everything, no solve yet
            commit = np.ones(len(cand_pair), bool)
        else:
            commit = later[cand_pair] <= last
        for k, a in zip(cand_pair[commit], acc[commit]):
            self.accept_log.append((int(k), int(a)))
        take = cand_pair[commit & (acc != 0)]
        r = self.results[take]
        self.f_key = np.concatenate([self.f_key, take.astype(np.int64)])
        self.f_from = np.concatenate([self.f_from, run["pair_from"][take].astype(np.int32)])
        self.f_to = np.concatenate([self.f_to, run["pair_to"][take].astype(np.int32)])
        self.f_score = np.concatenate([self.f_score, r["consensus"].astype(np.float64)])
        self.f_T = np.concatenate([self.f_T, r["T"]]); self.f_info = np.concatenate([self.f_info, r["information"]])
        # edge.valid_ = true for an accepted edge with matching_score >= min_accept_valid (graph_slam_node.cpp:809-811): it stays valid
        # in every later set_graph (A* walks valid edges only) and filter pass, whatever the filter says about it afterwards
        self.f_sticky = np.concatenate([self.f_sticky, val[commit & (acc != 0)] != 0])
        n_new_edges = (last + 1 - self.cur) + len(take)
        self.edges_since += n_new_edges
        n_nodes = last + 1
        self.cur = n_nodes
        if not len(trig) and not final:
            self.t["host"] += time.perf_counter() - t_host - t_gate
            return True
        # ---- keep the matcher busy while the solver runs: the next batch is launched before, collected after
        self._launch_next()
        # ---- edge filter (g2o_optimizer.cpp:74-103): every feature edge re-added with its end nodes' current poses
        t0 = time.perf_counter()
        nf = len(self.f_key)
        if nf:
            fe = np.zeros(nf, capi.FILTER_EDGE_DTYPE)
            fe["key"] = self.f_key; fe["matching_score"] = self.f_score; fe["valid"] = self.f_sticky
            fe["sensor_from"] = -1; fe["sensor_to"] = -1; fe["n_stamps_from"] = 1; fe["n_stamps_to"] = 1
            base = run["stamps_ns"].ctypes.data
            fe["stamps_from_ns"] = base + 8 * self.f_from.astype(np.uint64); fe["stamps_to_ns"] = base + 8 * self.f_to.astype(np.uint64)
            fe["transform"] = self.f_T; fe["displacement_from"] = I12; fe["displacement_to"] = I12
            fe["pose_from"] = self.poses[self.f_from].reshape(-1, 12); fe["pose_to"] = self.poses[self.f_to].reshape(-1, 12)
            self.filt.add_packed(fe)
        n_eval = self.filt.calc_valid_edges()
        valid_keys = self.filt.valid_edges().astype(np.int64)
        self.f_sticky |= np.isin(self.f_key, valid_keys)                            # graph.edge(id).valid_ = true (:101)
        in_solve = np.isin(self.f_key, valid_keys)
        t_filter = time.perf_counter() - t0
        self.t["filter"] += t_filter
        # ---- re-optimise (addGraphImpl: full rebuild; optimizeImpl; storeImpl)
        self._last = (n_nodes, self.poses[:n_nodes].reshape(-1, 12).copy(), in_solve.copy())      # what addGraphImpl is given (last_input)
        t0 = time.perf_counter()
        if self.incremental and self._pgo_nodes > 0 and hasattr(self.pgo, "append_graph"):
            # the graph resident in the solver, grown: the nodes and edges admitted since the last solve (odometry edge i-1 -> i of every
            # new node i, then the feature edges in acceptance order - the order addGraphImpl's two passes give the full arrays), and the
            # filter's verdict on the old feature edges.  The old nodes' poses ARE the solver's estimates (storeImpl wrote them here).
            a = self._pgo_nodes; f0 = self._pgo_feats
            no = n_nodes - a; nfn = nf - f0; E = no + nfn
            ne = {"from": np.concatenate([np.arange(a - 1, n_nodes - 1, dtype=np.int32), self.f_from[f0:]]),
                  "to": np.concatenate([np.arange(a, n_nodes, dtype=np.int32), self.f_to[f0:]]),
                  "type": np.concatenate([np.full(no, synth.EDGE_TYPE_ODOM, np.int32), np.full(nfn, synth.EDGE_TYPE_3D_FULL, np.int32)]),
                  "sensor_from": np.full(E, -1, np.int32), "sensor_to": np.full(E, -1, np.int32),
                  "valid": np.concatenate([np.ones(no, np.int32), in_solve[f0:].astype(np.int32)]),
                  "transform": np.concatenate([self.odo_T[a - 1:n_nodes - 1].reshape(no, 12), self.f_T[f0:]]),
                  "displacement_from": np.tile(I12, (E, 1)), "displacement_to": np.tile(I12, (E, 1)),
                  "information": np.concatenate([self.odo_info[a - 1:n_nodes - 1], self.f_info[f0:]]),
                  "diff_time": np.concatenate([np.full(no, 0.5), np.zeros(nfn)])}
            self.pgo.append_graph(self.poses[a:n_nodes].reshape(-1, 12), run["fixed"][a:n_nodes], ne, self._f_in_idx[:f0], in_solve[:f0])
            self._f_in_idx = np.concatenate([self._f_in_idx, (self._pgo_edges + no + np.arange(nfn)).astype(np.int32)])
            self._pgo_edges += E
        else:
            self.pgo.add_graph(*self.last_input)
            self._f_in_idx = (n_nodes - 1 + np.arange(nf)).astype(np.int32)
            self._pgo_edges = n_nodes - 1 + nf
        self._pgo_nodes = n_nodes; self._pgo_feats = nf
        t1 = time.perf_counter()
        st = self.pgo.optimize(self.lm_iterations)
        t2 = time.perf_counter()
        self.poses[:n_nodes] = self.pgo.store()[0].reshape(-1, 3, 4)
        t3 = time.perf_counter()
        self.t["add_graph"] += t1 - t0; self.t["optimize"] += t2 - t1; self.t["store"] += t3 - t2
        st = dict(st); st.update(n_nodes=n_nodes, n_feature_edges=nf, n_feature_valid=int(in_solve.sum()), clusters_evaluated=n_eval,
                                 add_graph_ms=1e3 * (t1 - t0), optimize_ms=1e3 * (t2 - t1), store_ms=1e3 * (t3 - t2),
                                 wall_s=t3 - self._t_first)
        self.solves.append(st)
        if len(self.poses_at_solve) < int(self.keep_poses_per_solve):
            self.poses_at_solve.append(self.poses[:n_nodes].copy())
        self.edges_since = 0
        if self.log:
            self.log("solve %3d: %5d nodes %5d feature edges (%d valid)  chi2 %.4g -> %.4g  %d LM its %d pcg  add %.1f ms opt %.1f ms"
                     % (len(self.solves), n_nodes, nf, int(in_solve.sum()), st["chi2_initial"], st["chi2_final"], st["iterations_done"],
                        st["pcg_iterations"], st["add_graph_ms"], st["optimize_ms"]))
        self.t["host"] += time.perf_counter() - t_host - (t3 - t0) - t_gate - t_filter        # this driver's own bookkeeping
        return self.cur < self.N

    def run_all(self):
        """The whole run.  Ranks other than the solver only match: they walk through the batches in the same order and meet the
        solver rank in every batch's result gather (the only communication), i.e. they wait there while it solves."""
        t0 = time.perf_counter()
        if self.is_solver:
            while self.step():
                pass
        self._need_results(self.P)
        return time.perf_counter() - t0

    def summary(self, wall):
        s = self.solves
        edges_iter = sum(x["n_edges"] * x["iterations_done"] for x in s)
        opt_s = self.t["optimize"]
        return OnlineStats(
            wall_s=wall, n_nodes=self.cur, n_pairs=self.P, n_solves=len(s), feature_edges_accepted=len(self.f_key),
            feature_edges_valid=int(self.f_sticky.sum()), pairs_ok=int((self.results["ok"] != 0).sum()),
            edges_x_iterations=edges_iter, edges_per_s_solver=edges_iter / opt_s if opt_s > 0 else 0.0,
            edges_per_s_wall=edges_iter / wall if wall > 0 else 0.0, pairs_per_s_wall=self.P / wall if wall > 0 else 0.0,
            add_graph_ms_per_solve=1e3 * self.t["add_graph"] / max(len(s), 1), optimize_ms_per_solve=1e3 * opt_s / max(len(s), 1),
            structure_ms_per_solve=float(np.mean([x.get("structure_ms", 0.0) for x in s])) if s else 0.0,
            seconds=dict((k, round(v, 4)) for k, v in self.t.items()),
            pcg_iterations=sum(x["pcg_iterations"] for x in s), lm_iterations=sum(x["iterations_done"] for x in s),
            not_converged=sum(1 for x in s if x["status"] != 0),
            solves_reduced=sum(1 for x in s if x.get("n_eliminated", 0) > 0), solves_strong_aggregates=sum(1 for x in s if x.get("reduced_strong", 0)))

    def close(self):
        for h in (self.matcher, self.gate, self.filt, self.pgo):
            if h is not None:
                h.close()
