#!/usr/bin/env python3
"""Rates that depend on how the pass driver sizes its PCG segments (uzl_pgo_cfg::pass_history = 1: nothing remembered from an earlier
optimize): config 2 alone, 16 config-2 graphs, 16 chain-like graphs, 64 config-1-sized graphs.   python tests/diag/pass_rates.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import capi, synth
def one(n, e, seed, reps=10):
    g = synth.make_pose_graph(n, e, seed=seed)
    p = capi.Pgo(pass_history=1); p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"]); p.optimize(20)
    for _ in range(3): p.reset(); p.optimize(20)
    t0 = time.perf_counter()
    for _ in range(reps): p.reset(); st = p.optimize(20)
    dt = (time.perf_counter() - t0) / reps
    p.close()
    print("%5d / %5d alone: %.3f ms, %d passes, %d PCG iterations" % (n, e, 1e3 * dt, st["lm_passes"], st["pcg_iterations"]), flush=True)
def batch(n, e, seed0, B, reps=5):
    bt = capi.PgoBatch(B, pass_history=1)
    for k in range(B):
        g = synth.make_pose_graph(n, e, seed=seed0 + (1000 * k if (n, e) == (1000, 5000) else k)); bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    bt.optimize(20)
    for _ in range(2):
        for p in bt.graphs: p.reset()
        bt.optimize(20)
    t0 = time.perf_counter(); ed = 0
    for _ in range(reps):
        for p in bt.graphs: p.reset()
        st = bt.optimize(20); ed += sum(x["n_edges"] * x["iterations_done"] for x in st)
    dt = time.perf_counter() - t0
    print("%2d x %5d / %5d: %.3f ms per batch, %.1f M edges/s, passes (graph 0) %d" % (B, n, e, 1e3 * dt / reps, ed / dt / 1e6, st[0]["lm_passes"]), flush=True)
    bt.close()
one(1000, 5000, 12345); one(1500, 1530, 4040); one(100, 300, 777)
batch(1000, 5000, 12345, 16); batch(1500, 1530, 4040, 16); batch(100, 300, 777, 64)
