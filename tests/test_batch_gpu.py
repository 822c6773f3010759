"""Batched multi-graph solve (uzl_pgo_batch_*): B independent graphs through shared launches.  Every graph's result must be
bit-identical to uzl_pgo_optimize of that graph alone, and (through that) within the north-star tolerance of the oracle."""
import numpy as np
import pytest

from uzliti_slam_amd import synth

pytestmark = pytest.mark.gpu


def _single(capi, g, its):
    p = capi.Pgo()
    p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    st = p.optimize(its)
    poses, err, used = p.store()
    p.close()
    return st, poses, err


@pytest.mark.parametrize("n,e,B,its", [(1000, 5000, 16, 20), (300, 1200, 5, 8), (2000, 9000, 3, 6)])
def test_batch_is_bit_identical_to_single_solves(capi, oracle, n, e, B, its):
    graphs = [synth.make_pose_graph(n, e, seed=100 + 7 * k, outlier_frac=0.05 + 0.02 * (k % 3)) for k in range(B)]
    bt = capi.PgoBatch(B)
    for k, g in enumerate(graphs):
        bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    stats = bt.optimize(its)
    assert bt.n_batched == B                                     # same-size graphs: one batch, no fallback
    for k, g in enumerate(graphs):
        st1, poses1, err1 = _single(capi, g, its)
        poses, err, _ = bt.graphs[k].store()
        assert np.array_equal(poses, poses1), k                  # bit for bit
        assert np.array_equal(err, err1, equal_nan=True)
        for f in ("iterations_done", "lm_trials", "pcg_iterations", "precond_builds", "terminated_early", "n_edges", "n_gauge_fixed"):
            assert stats[k][f] == st1[f], (k, f, stats[k][f], st1[f])
        assert stats[k]["chi2_initial"] == st1["chi2_initial"] and stats[k]["chi2_final"] == st1["chi2_final"] and stats[k]["lambda_final"] == st1["lambda_final"]
    # and one of them against the oracle's direct solve
    g = graphs[1]
    fl = oracle.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    fixed, _ = oracle.set_fixed_nodes(fl["fixed"], fl["ij"])
    P, _ = oracle.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=its)
    dt, dr = synth.pose_errors(bt.graphs[1].store()[0].reshape(-1, 3, 4), P.reshape(-1, 3, 4))
    assert dt < 1e-3 and dr < 1e-4, (dt, dr)
    # a second solve continues from the current poses, like uzl_pgo_optimize does; reset() restores the inputs
    for k in range(B):
        bt.graphs[k].reset()
    again = bt.optimize(its)
    assert all(np.array_equal(bt.graphs[k].store()[0], _single(capi, graphs[k], its)[1]) for k in range(min(B, 3)))
    assert [a["pcg_iterations"] for a in again] == [s["pcg_iterations"] for s in stats]
    bt.close()


def test_mixed_shapes_fall_back_to_single_solves(capi):
    """Graphs that do not share a hierarchy shape (or are too large for the small-graph class) are solved one by one inside the
    call: same results, n_batched = 0."""
    shapes = [(400, 1500), (1000, 5000), (3000, 12000)]
    graphs = [synth.make_pose_graph(n, e, seed=5 + k) for k, (n, e) in enumerate(shapes)]
    bt = capi.PgoBatch(len(graphs))
    for k, g in enumerate(graphs):
        bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    stats = bt.optimize(6)
    assert bt.n_batched == 0
    for k, g in enumerate(graphs):
        st1, poses1, _ = _single(capi, g, 6)
        assert np.array_equal(bt.graphs[k].store()[0], poses1) and stats[k]["pcg_iterations"] == st1["pcg_iterations"]
    bt.close()


def test_chain_like_graphs_in_a_batch(capi, oracle):
    """Chain-like graphs - an odometry chain plus a few loop closures, the shape of the reference's local-scope graphs
    (graph_slam_node.cpp:578-663, g2o_optimizer.cpp:190-259) - are Schur-reduced (csrc/pgo_schur.hpp) and batch on their REDUCED systems,
    whose sizes differ from graph to graph: every launch takes the largest graph's grid.  Results equal the single solves bit for bit."""
    graphs = [synth.make_pose_graph(1500, 1530, seed=40 + k) for k in range(16)]
    bt = capi.PgoBatch(len(graphs))
    for k, g in enumerate(graphs):
        bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    stats = bt.optimize(12)
    assert bt.n_batched == len(graphs), "chain-like graphs must batch (round 3 sent them one by one through the single-graph path)"
    sizes = set()
    for k, g in enumerate(graphs):
        st1, poses1, err1 = _single(capi, g, 12)
        assert st1["n_eliminated"] > 0 and stats[k]["n_eliminated"] == st1["n_eliminated"]
        sizes.add(st1["n_vertices"] - st1["n_eliminated"])
        poses, err, _ = bt.graphs[k].store()
        assert np.array_equal(poses, poses1), k
        assert np.array_equal(err, err1, equal_nan=True)
        for f in ("iterations_done", "lm_trials", "pcg_iterations", "precond_builds", "terminated_early", "chi2_final", "lambda_final"):
            assert stats[k][f] == st1[f], (k, f, stats[k][f], st1[f])
    assert len(sizes) > 1                                          # the reduced systems really were of different sizes
    g = graphs[5]
    fl = oracle.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    fixed, _ = oracle.set_fixed_nodes(fl["fixed"], fl["ij"])
    P, _ = oracle.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=12)
    dt, dr = synth.pose_errors(bt.graphs[5].store()[0].reshape(-1, 3, 4), P.reshape(-1, 3, 4))
    assert dt < 1e-3 and dr < 1e-4, (dt, dr)
    bt.close()


def test_zero_residual_and_anomalous_graphs_in_a_batch(capi):
    """Few loop closures and a zero-residual graph in one batch: rejected trials, early termination and, where the solver meets an
    anomaly, the per-graph fallback - results still equal the single solves."""
    graphs = [synth.make_pose_graph(1500, 1500 + 10 * k, seed=40 + k) for k in range(4)]
    graphs[3] = synth.make_pose_graph(1500, 1499, seed=9)              # a pure odometry chain: chi2 = 0 from the start
    bt = capi.PgoBatch(len(graphs))
    for k, g in enumerate(graphs):
        bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    stats = bt.optimize(12)
    for k, g in enumerate(graphs):
        st1, poses1, _ = _single(capi, g, 12)
        assert np.array_equal(bt.graphs[k].store()[0], poses1), k
        assert stats[k]["iterations_done"] == st1["iterations_done"] and stats[k]["lm_trials"] == st1["lm_trials"]
    bt.close()


def test_thirteen_chain_like_graphs_with_fallbacks(capi):
    """13 chain-like graphs, among them a zero-residual chain and graphs with few loop closures (rejected trials, the per-graph fallback
    where the solver meets an anomaly).  Every graph must equal its own solve."""
    graphs = [synth.make_pose_graph(1500, 1530 + 3 * k, seed=70 + k) for k in range(13)]
    graphs[9] = synth.make_pose_graph(1500, 1499, seed=9)              # a pure odometry chain: chi2 = 0 from the start
    graphs[11] = synth.make_pose_graph(1500, 1503, seed=11)
    bt = capi.PgoBatch(len(graphs))
    for k, g in enumerate(graphs):
        bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    stats = bt.optimize(12)
    for k, g in enumerate(graphs):
        st1, poses1, _ = _single(capi, g, 12)
        assert np.array_equal(bt.graphs[k].store()[0], poses1), k
        for f in ("iterations_done", "lm_trials", "pcg_iterations", "terminated_early", "n_eliminated"):
            assert stats[k][f] == st1[f], (k, f, stats[k][f], st1[f])
    bt.close()


@pytest.mark.parametrize("resident", [1, 3, 8])
def test_queue_with_fewer_resident_slots_than_graphs(capi, resident):
    """uzl_pgo_batch_set_resident: 20 graphs through 1 / 3 / 8 slots - a finished graph hands its slot to the next one of the queue.
    Graphs with different numbers of LM trials (some start far off, some terminate early) sit in different phases of the loop at the
    same time; every graph must still come out bit-identical to its own uzl_pgo_optimize."""
    rng = np.random.default_rng(3)
    graphs = []
    for k in range(20):
        g = synth.make_pose_graph(600, 2600, seed=300 + k, outlier_frac=0.05 * (k % 4))
        if k % 5 == 0:                                            # a poor start: rejected trials, more rounds than the others
            P0 = g["nodes_pose"].reshape(-1, 3, 4).copy()
            P0[1:] = synth.se3_mul(P0[1:], synth.se3_from_noise(rng.normal(0, 0.8, (599, 3)), rng.normal(0, 0.4, (599, 3))))
            g["nodes_pose"] = P0.reshape(-1, 12)
        if k % 7 == 3:                                            # already at the optimum of a noise-free graph: Terminate after a trial or two
            g["nodes_pose"] = g["gt_pose"].copy()
        graphs.append(g)
    bt = capi.PgoBatch(len(graphs))
    bt.set_resident(resident)
    for k, g in enumerate(graphs):
        bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    stats = bt.optimize(10)
    assert bt.n_batched == len(graphs)
    trials = set()
    for k, g in enumerate(graphs):
        st1, poses1, _ = _single(capi, g, 10)
        assert np.array_equal(bt.graphs[k].store()[0], poses1), k
        for f in ("iterations_done", "lm_trials", "pcg_iterations", "terminated_early"):
            assert stats[k][f] == st1[f], (k, f, stats[k][f], st1[f])
        assert stats[k]["chi2_final"] == st1["chi2_final"]
        trials.add(st1["lm_trials"])
    assert len(trials) >= 2                                       # the graphs really did not march in step
    bt.close()


def test_one_launch_sequence_in_the_diagnostic_build():
    """UZL_BATCH_LANES=1 (diagnostic build): every batch as ONE launch sequence (the default drives the second half of 12 and more graphs
    from a second host thread on streams of its own).  Random batches of 2 - 24 graphs, some through a queue: every graph bit-identical
    to its own solve - as in the default, which tests/diag/stress_batch.py checks the same way."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    diag = os.path.join(os.path.dirname(here), "uzliti_slam_amd", "libuzl_mi355x_diag.so")
    assert os.path.exists(diag), "build the diagnostic library: make -C uzliti_slam_amd/csrc diag"
    e = dict(os.environ, UZL_LIB=diag, UZL_BATCH_LANES="1")
    out = subprocess.run([sys.executable, os.path.join(here, "diag", "stress_batch.py"), "5", "11"], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    assert "0 misses" in out.stdout


def test_anomaly_fallback_from_a_helper_thread_leaves_the_other_sequence_alone():
    """A graph of the SECOND launch sequence (driven from a helper thread) that meets an anomaly is solved again by the host-driven loop
    on streams of its own.  Every handle of a batch borrows sequence 0's streams, and sequence 0 may be capturing on them at that
    moment: the fallback must not synchronize them (round 5's advisor finding).  UZL_BATCH_FORCE_ANOMALY_N (diagnostic build) sends the
    one graph of 301 nodes down that path in the batch's FIRST optimize (the one that captures); all 16 graphs still equal their own
    solves bit for bit, 15 of them batched."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    diag = os.path.join(root, "uzliti_slam_amd", "libuzl_mi355x_diag.so")
    assert os.path.exists(diag), "build the diagnostic library: make -C uzliti_slam_amd/csrc diag"
    code = (
        "import numpy as np\n"
        "from uzliti_slam_amd import capi, synth\n"
        "gs = [synth.make_pose_graph(301 if k == 12 else 300, 1200, seed=900 + k) for k in range(16)]\n"
        "bt = capi.PgoBatch(16)\n"
        "for k, g in enumerate(gs): bt.graphs[k].add_graph(g['nodes_pose'], g['nodes_fixed'], g['edges'])\n"
        "for rnd in range(2):\n"
        "    for p in bt.graphs: p.reset()\n"
        "    st = bt.optimize(8)\n"
        "    assert bt.n_batched == 15, bt.n_batched\n"
        "    for k, g in enumerate(gs):\n"
        "        p = capi.Pgo(); p.add_graph(g['nodes_pose'], g['nodes_fixed'], g['edges']); s1 = p.optimize(8); ref = p.store()[0].copy(); p.close()\n"
        "        assert np.array_equal(bt.graphs[k].store()[0], ref), (rnd, k)\n"
        "        assert st[k]['lm_trials'] == s1['lm_trials'] and st[k]['chi2_final'] == s1['chi2_final'], (rnd, k)\n"
        "bt.close()\n"
        "print('fallback ok')\n")
    e = dict(os.environ, UZL_LIB=diag, UZL_BATCH_FORCE_ANOMALY_N="301", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0 and "fallback ok" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


def test_a_batch_handle_solved_on_its_own_and_from_two_threads(capi):
    """The graphs of a batch run on the batch's streams (a handle of its own costs two hipStreamCreates).  A batch's handle is still an
    ordinary handle: solved directly through uzl_pgo_optimize it takes streams of its own at that moment - two such handles driven from
    two threads at once (captures, rebuild events) give what fresh handles give, and the batch still solves all of its graphs afterwards."""
    import threading
    graphs = [synth.make_pose_graph(700, 3000, seed=500 + k) for k in range(4)]
    ref = [_single(capi, g, 10) for g in graphs]
    bt = capi.PgoBatch(len(graphs))
    for k, g in enumerate(graphs):
        bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    out = {}

    def work(k):
        for _ in range(3):                                        # the second and third solve replay captured segments
            bt.graphs[k].reset()
            st = bt.graphs[k].optimize(10)
        out[k] = (st, bt.graphs[k].store()[0].copy())
    th = [threading.Thread(target=work, args=(k,)) for k in (1, 2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in (1, 2):
        assert np.array_equal(out[k][1], ref[k][1]), k
        assert out[k][0]["pcg_iterations"] == ref[k][0]["pcg_iterations"]
    for p in bt.graphs:
        p.reset()
    stats = bt.optimize(10)
    assert bt.n_batched == len(graphs)
    for k in range(len(graphs)):
        assert np.array_equal(bt.graphs[k].store()[0], ref[k][1]), k
        assert stats[k]["lm_trials"] == ref[k][0]["lm_trials"]
    bt.close()
