"""BASELINE config 5 through uzliti_slam_amd/online.py: node-pair match jobs feed a growing graph that is re-optimised every 256
edges (estimator -> acceptance gate -> edge filter -> solver, all on the GPU through the C ABI).

At reduced size every solve is compared with the CPU oracle's pipeline on the same schedule; at the full size of the config
(4096 pairs, 20k nodes: the oracle's direct solves would take tens of minutes) the run is checked through properties."""
import numpy as np
import pytest

from uzliti_slam_amd import online, synth

pytestmark = pytest.mark.gpu


def _edges_subset(e, keep):
    return {k: np.asarray(v)[keep] for k, v in e.items()}


def test_reduced_online_run_matches_the_oracle(capi, oracle):
    """1500 nodes / 384 pairs / 300 keypoints: same accepted edges, same filter verdicts and poses within 1e-3 m / 1e-4 rad of an
    oracle run that replays the identical schedule (gate, filter and solver of the CPU checker)."""
    run = synth.make_online_run(1500, 384, n_kp=300)
    mc = dict(ransac_iteration=200)
    o = online.OnlineSlam(run, match_batch=100, lm_iterations=8, match_cfg=mc)
    o.upload_frames()
    # ---- oracle replay, driven by the same class through stand-ins of the four handles (tests/online_stubs.py)
    from online_stubs import oracle_online
    c = oracle_online(oracle, run, ransac_iteration=200, match_batch=100, lm_iterations=8, match_cfg=mc)
    c.upload_frames()
    o.run_all(); c.run_all()
    assert len(o.solves) == len(c.solves) >= 5
    assert np.array_equal(o.results["consensus"], c.results["consensus"]) and np.array_equal(o.results["T"], c.results["T"])   # edges bit-exact
    assert o.accept_log == c.accept_log and len(o.accept_log) > 300                    # gate verdicts identical
    assert np.array_equal(o.f_key, c.f_key) and np.array_equal(o.f_sticky, c.f_sticky)     # filter verdicts identical
    assert int(o.f_sticky.sum()) >= 30
    for a, b in zip(o.solves, c.solves):
        assert (a["n_nodes"], a["n_feature_valid"], a["n_edges"]) == (b["n_nodes"], b["n_feature_valid"], b["n_edges"])
    dt, dr = synth.pose_errors(o.poses, c.poses)
    assert dt < 1e-3 and dr < 1e-4, (dt, dr)
    o.close()


def test_reduced_online_run_with_min_accept_valid(capi, oracle):
    """min_accept_valid finite (the launch files use 150 / 200; graph_slam_node.cpp:809-811): accepted edges above it are valid from
    acceptance on, which changes what A* can walk in every later interval.  GPU run == oracle replay."""
    from online_stubs import oracle_online
    run = synth.make_online_run(1200, 300, n_kp=200)
    mc = dict(ransac_iteration=100)
    gc = dict(min_accept_valid=97.0)
    o = online.OnlineSlam(run, match_batch=100, lm_iterations=6, match_cfg=mc, gate_cfg=gc)
    c = oracle_online(oracle, run, ransac_iteration=100, match_batch=100, lm_iterations=6, match_cfg=mc, gate_cfg=gc)
    o.upload_frames(); c.upload_frames()
    o.run_all(); c.run_all()
    hi = o.f_score >= 97.0
    assert hi.any() and (~hi).any() and o.f_sticky[hi].all()
    assert o.accept_log == c.accept_log and np.array_equal(o.f_key, c.f_key) and np.array_equal(o.f_sticky, c.f_sticky)
    assert len(o.solves) == len(c.solves) >= 4
    dt, dr = synth.pose_errors(o.poses, c.poses)
    assert dt < 1e-3 and dr < 1e-4, (dt, dr)
    o.close()


@pytest.fixture(scope="module")
def full_run():
    return synth.make_online_run(20000, 4096, n_kp=1000)


def test_full_size_run_properties(capi, oracle, full_run):
    """BASELINE config 5 at its size: 4096 pairs x 1000 ORB-256 keypoints, graph growing to 20k nodes, re-optimised every 256 edges."""
    run = full_run
    a = online.OnlineSlam(run, match_batch=512)
    a.upload_frames()
    a.run_all()
    assert a.cur == 20000 and len(a.solves) >= 75
    assert all(s["status"] == 0 and s["pcg_not_converged"] == 0 for s in a.solves)
    # chi2 never rises within a re-optimisation (LM only accepts descent steps)
    assert all(s["chi2_final"] <= s["chi2_initial"] * (1 + 1e-12) for s in a.solves)
    # the map is better than dead reckoning, by a wide margin
    gt = run["gt"]
    ate0 = np.linalg.norm(run["init"][:, :, 3] - gt[:, :, 3], axis=1).mean(); ate1 = np.linalg.norm(a.poses[:, :, 3] - gt[:, :, 3], axis=1).mean()
    assert ate1 < 0.5 * ate0, (ate0, ate1)
    # aliased pairs (wrong place) mostly do not survive gate + filter
    alias = run["pair_alias"]
    assert alias[a.f_key[a.f_sticky]].mean() < 0.5 * alias.mean()
    # ---- the accepted-edge list, the filter verdicts and the poses do not depend on the match batch size
    b = online.OnlineSlam(run, match_batch=4096)
    b.upload_frames()
    b.run_all()
    assert a.accept_log == b.accept_log and np.array_equal(a.f_key, b.f_key) and np.array_equal(a.f_sticky, b.f_sticky)
    assert np.array_equal(a.poses, b.poses)
    assert [s["n_nodes"] for s in a.solves] == [s["n_nodes"] for s in b.solves]
    # ---- the last re-optimisation equals a solve of the same input on a fresh handle: nothing of the 90-odd earlier structures
    #      leaks into the last one
    fresh = capi.Pgo()
    fresh.add_graph(*a.last_input)
    st = fresh.optimize(20)
    assert st["status"] == 0 and st["n_edges"] == a.solves[-1]["n_edges"]
    dt, dr = synth.pose_errors(fresh.store()[0].reshape(-1, 3, 4), a.poses)
    assert dt < 1e-3 and dr < 1e-4, (dt, dr)
    # ---- and it is within the parity bar of the CPU checker's direct solve of the same input at full size (20k nodes: the graph is
    #      near-tree, so the oracle's Cholesky has little fill and takes seconds).  G2oOptimizer::optimizeImpl, g2o_optimizer.cpp:137-149
    fl = oracle.flatten_graph(*a.last_input)
    fx, _ = oracle.set_fixed_nodes(fl["fixed"], fl["ij"])
    P, so = oracle.pgo_optimize(fl["poses"], fx, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=20)
    assert so["n_edges"] == st["n_edges"]
    dt, dr = synth.pose_errors(a.poses, P.reshape(-1, 3, 4))
    assert dt < 1e-3 and dr < 1e-4, (dt, dr)
    assert abs(a.solves[-1]["chi2_final"] - so["chi2_final"]) <= 1e-6 * abs(so["chi2_final"]) + 1e-9
    fresh.close(); a.close(); b.close()


def test_two_ranks_equal_one_rank(capi, tmp_path):
    """Multi-process path of config 5: torchrun, 2 ranks (both on cuda:0 of the one-GPU box, gloo for the result gather): pair jobs
    sharded per batch, gate / filter / solver on rank 0.  Outcome must be identical to the one-rank run, bit for bit."""
    import os
    import socket
    import subprocess
    import sys
    n_nodes, n_pairs, n_kp = 1200, 300, 200
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    here = os.path.dirname(os.path.abspath(__file__))
    out = str(tmp_path / "two_ranks.npz")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(here, "_online_worker.py"), out, str(n_nodes), str(n_pairs), str(n_kp)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ONLINE_OK world=2") == 2
    z = np.load(out)
    run = synth.make_online_run(n_nodes, n_pairs, n_kp=n_kp)
    o = online.OnlineSlam(run, match_batch=300, lm_iterations=6, match_cfg=dict(ransac_iteration=100))
    o.upload_frames()
    o.run_all()
    assert np.array_equal(z["consensus"], o.results["consensus"]) and np.array_equal(z["T"], o.results["T"])
    assert np.array_equal(z["accept"], np.array(o.accept_log)) and np.array_equal(z["f_key"], o.f_key) and np.array_equal(z["f_sticky"], o.f_sticky)
    assert np.array_equal(z["poses"], o.poses)
    assert len(o.solves) >= 4 and int(o.f_sticky.sum()) > 0
    o.close()


def test_online_two_ranks_native(capi, tmp_path):
    """Config 5's multi-process path with ONE GPU PER RANK (skipped on a one-GPU box; the first multi-GPU box validates it by itself):
    pair jobs sharded per batch over two devices, gate / filter / solver on rank 0.  Outcome identical to the one-rank run, bit for bit."""
    import os
    import socket
    import subprocess
    import sys
    if capi.device_count() < 2:
        pytest.skip("needs two GPUs (one process per GPU)")
    n_nodes, n_pairs, n_kp = 1200, 300, 200
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", UZL_ONE_GPU_PER_RANK="1")
    here = os.path.dirname(os.path.abspath(__file__))
    out = str(tmp_path / "two_ranks_native.npz")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(here, "_online_worker.py"), out, str(n_nodes), str(n_pairs), str(n_kp)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ONLINE_OK world=2 rank=0 device=0" in r.stdout and "ONLINE_OK world=2 rank=1 device=1" in r.stdout
    z = np.load(out)
    run = synth.make_online_run(n_nodes, n_pairs, n_kp=n_kp)
    o = online.OnlineSlam(run, match_batch=300, lm_iterations=6, match_cfg=dict(ransac_iteration=100))
    o.upload_frames()
    o.run_all()
    assert np.array_equal(z["consensus"], o.results["consensus"]) and np.array_equal(z["T"], o.results["T"])
    assert np.array_equal(z["accept"], np.array(o.accept_log)) and np.array_equal(z["poses"], o.poses)
    o.close()

