#!/usr/bin/env python3
"""Diagnostic: does what a process did before (other batches created, solved and closed) change the rate of a batch?  The same 16 chain-like
graphs are solved as a fresh batch at the start, after a batch of 64 small graphs, after a batch of 16 config-2 graphs, and after importing torch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if "torch" in sys.argv[1:]:                  # as bench.py does: torch owns the device before the library touches it
    import torch
    torch.cuda.synchronize()
    if "alloc" in sys.argv[1:]:
        _x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
from uzliti_slam_amd import capi, synth    # noqa: E402


def run(graphs, label, reps=5):
    bt = capi.PgoBatch(len(graphs))
    for k, g in enumerate(graphs):
        bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    bt.optimize(20)
    ts = []
    e = 0
    for _ in range(reps):
        for p in bt.graphs:
            p.reset()
        t0 = time.perf_counter(); st = bt.optimize(20); ts.append(time.perf_counter() - t0)
        e = sum(x["n_edges"] * x["iterations_done"] for x in st)
    med = sorted(ts)[len(ts) // 2]
    print("%-44s %2d graphs: median %.2f ms best %.2f ms -> %.1f M edges/s" % (label, len(graphs), 1e3 * med, 1e3 * min(ts), e / med / 1e6), flush=True)
    bt.close()


chain = [synth.make_pose_graph(1500, 1530, seed=4040 + k) for k in range(16)]
small = [synth.make_pose_graph(100, 300, seed=777 + 1000 * k) for k in range(64)]
c2 = [synth.make_pose_graph(1000, 5000, seed=12345 + 1000 * k) for k in range(16)]
run(chain, "chain-like, fresh process")
run(chain, "chain-like, again")
run(small, "small graphs")
run(chain, "chain-like after the small batch")
run(c2, "config 2")
run(chain, "chain-like after the config-2 batch")
one = capi.Pgo(); one.add_graph(chain[0]["nodes_pose"], chain[0]["nodes_fixed"], chain[0]["edges"]); one.optimize(20); one.close()
run(chain, "chain-like after a single handle")
