"""Side-by-side of two bench.py JSON lines: python tests/diag/bench_diff.py OLD.json NEW.json"""
import json
import sys


def load(p):
    return json.loads(open(p).read().strip().splitlines()[-1])


def g(d, *k):
    for x in k:
        d = d.get(x, {}) if isinstance(d, dict) else {}
    return d


KEYS = [("value",), ("ms_per_step",), ("config", "pcg_iterations_per_solve"), ("kernels_ms_per_solve", "ml_cg"),
        ("kernels_ms_per_solve", "pcg_spmv"), ("xy_only", "ms_per_solve"), ("batched", "ms_per_batch"), ("batched", "vs_single_graph"),
        ("batched", "small_graphs", "ms_per_batch"), ("c4_1gpu", "ms_per_solve"), ("c4_1gpu", "pcg_iterations_per_solve"),
        ("c4_1gpu", "kernels_ms_per_solve", "ml_cg"), ("c4_1gpu", "kernels_ms_per_solve", "pcg_spmv"),
        ("online_c5", "wall_s"), ("online_c5", "seconds", "optimize"), ("online_c5", "seconds", "gate"), ("online_c5", "pcg_iterations"),
        ("online_c5", "not_converged"), ("online_c5", "ate_online_m"), ("secondary", "ms_per_step")]

if __name__ == "__main__":
    o, d = load(sys.argv[1]), load(sys.argv[2])
    for k in KEYS:
        print("%-55s %s -> %s" % (".".join(k), g(o, *k), g(d, *k)))
    for r in g(d, "batched", "rooflines") or []:
        print(r["kernel"], r["avg_launch_us"], "us", r["achieved"], r["unit"])
