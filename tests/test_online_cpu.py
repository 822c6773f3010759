"""The N > 1 path of BASELINE config 5 on the CPU: two gloo ranks run the online driver (uzliti_slam_amd/online.py) with the CPU
checker standing in for the four GPU handles.  What is tested is host logic: the per-batch sharding of the pair jobs, the gather in
job order, the solver rank's schedule - the outcome must equal the one-rank run bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np

from uzliti_slam_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_gloo_ranks_equal_one_rank(oracle, tmp_path):
    from online_stubs import oracle_online
    n_nodes, n_pairs, n_kp = 700, 100, 120
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "two_ranks_cpu.npz")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(HERE, "_online_cpu_worker.py"), out, str(n_nodes), str(n_pairs), str(n_kp)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ONLINE_CPU_OK world=2") == 2
    z = np.load(out)
    run = synth.make_online_run(n_nodes, n_pairs, n_kp=n_kp)
    o = oracle_online(oracle, run, ransac_iteration=60, match_batch=120, lm_iterations=4, reopt_edges=64)
    o.upload_frames()
    o.run_all()
    assert np.array_equal(z["consensus"], o.results["consensus"]) and np.array_equal(z["T"], o.results["T"])
    assert np.array_equal(z["accept"], np.array(o.accept_log)) and np.array_equal(z["f_key"], o.f_key) and np.array_equal(z["f_sticky"], o.f_sticky)
    assert np.array_equal(z["poses"], o.poses)
    assert int(z["n_solves"]) == len(o.solves) >= 5 and len(o.f_key) > 50


def test_gate_valid_flag_is_kept(oracle):
    """newEdgeCallback sets edge.valid_ when matching_score >= min_accept_valid (graph_slam_node.cpp:809-811; the launch files use
    150 / 200): such an edge must enter every later set_graph as valid (A* walks valid edges only), before any filter verdict."""
    from online_stubs import oracle_online
    run = synth.make_online_run(500, 80, n_kp=120)
    o = oracle_online(oracle, run, ransac_iteration=60, match_batch=80, lm_iterations=3, reopt_edges=64, gate_cfg=dict(min_accept_valid=60.0))
    o.upload_frames()
    seen = []
    real = o.gate.set_graph
    o.gate.set_graph = lambda poses, edges, merged=None: (seen.append(np.array(edges["valid"][len(poses) - 1:])), real(poses, edges, merged))[1]
    o.run_all()
    hi = o.f_score >= 60.0
    assert hi.any() and (~hi).any()
    assert o.f_sticky[hi].all()                                      # every accepted edge above the threshold is valid from its acceptance on
    k = int(np.nonzero(hi)[0][0])
    later = [v for v in seen if len(v) > k]
    assert later and all(v[k] for v in later)                        # ... and is handed back to the gate as valid in every later interval
    o.close()
