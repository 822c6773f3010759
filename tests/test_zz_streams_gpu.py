"""The device's stream pool (csrc/uzl_streams.hip): streams that have to run side by side - a batch's launch sequences and their rebuild
streams - are leased from one pool per device and process, measured against each other at most once.  The file sorts last on purpose:
what it looks at is a property of the box's queues and pipes under whatever else runs on the GPU, and the driver runs the suite with
-x - a noisy box must not hide the parity tests behind it.  Nothing here asserts a timing ratio: the tests read decisions."""
import ctypes

import numpy as np
import pytest

from uzliti_slam_amd import synth

pytestmark = pytest.mark.gpu


def test_stream_pair_verdicts_have_the_shape_of_queues_and_pipes(capi):
    """Standing in each other's way is a property of the PAIR: the test reads the pool's own cached verdicts (one measurement per
    unordered pair, so both sides of a pair see the same decision by construction), a stream is never measured against itself, and
    independent pairs must exist among six streams of two priorities (their queues sit on four compute pipes)."""
    n = 6
    lib = capi.diag_lib()
    P32 = ctypes.POINTER(ctypes.c_int32)
    v = np.zeros((n, n), np.int32)
    r = np.zeros((n, n), np.int32)
    ms = ctypes.c_double(0.)
    assert lib.uzl_debug_stream_pairs(ctypes.c_int(n), ctypes.c_int(200), v.ctypes.data_as(P32), r.ctypes.data_as(P32), ctypes.byref(ms)) == 0
    assert np.all(np.diag(v) == -1)
    off = ~np.eye(n, dtype=bool)
    assert np.array_equal(v, v.T) and set(np.unique(v[off])) <= {0, 1}
    assert (v[off] == 1).sum() >= 4, "fewer than two independent pairs:\n%s\n%s" % (v, r)       # (each pair appears twice)


def test_pool_measures_a_pair_once_and_hands_the_same_streams_back(capi):
    """Batches of 16 graphs one after the other: a batch's streams are leased from the pool and go back to it, a pair of streams is
    measured at most once per process - so after a round or two (the first batches may still meet pooled streams they have not been paired
    with: solver handles of earlier tests left theirs) a batch is created without a single probe launch, on the same streams as the one
    before; every round solves all 16 graphs to the same bits."""
    import gc
    graphs = [synth.make_pose_graph(300, 1200, seed=900 + k) for k in range(16)]
    res, measured, pooled = [], [], []
    gc.collect()          # solver handles of earlier tests give their stream pairs back when their wrappers are collected: not in the middle of this test
    s0 = capi.stream_stats(0)
    for rnd in range(4):
        bt = capi.PgoBatch(len(graphs))
        for k, g in enumerate(graphs):
            bt.graphs[k].add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
        bt.optimize(6)
        assert bt.n_batched == len(graphs)
        res.append([bt.graphs[k].store()[0].copy() for k in range(len(graphs))])
        st = capi.stream_stats(0)
        assert st["leased"] >= s0["leased"] + 2
        measured.append(st["pairs_measured"]); pooled.append(st["pooled"])
        bt.close()
    assert capi.stream_stats(0)["leased"] == s0["leased"]               # everything went back
    assert measured[3] == measured[2] and pooled[3] == pooled[2], (s0, measured, pooled)      # a settled pool: no probe, no new stream
    assert measured[3] - measured[0] <= 6, measured                      # and it settles at once
    for r in res[1:]:
        for x, y in zip(res[0], r):
            assert np.array_equal(x, y)


def test_no_probe_means_one_launch_sequence_and_the_same_results(capi):
    """UZL_STREAM_PROBE=0: no stream pair is measured, a batch of 16 runs as one launch sequence on whatever streams it gets - the
    deterministic layout for a GPU that is shared with other processes.  Results are those of the default layout, bit for bit."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, hashlib, numpy as np\n"
        "from uzliti_slam_amd import capi, synth\n"
        "bt = capi.PgoBatch(16)\n"
        "for k in range(16):\n"
        "    g = synth.make_pose_graph(300, 1200, seed=900 + k); bt.graphs[k].add_graph(g['nodes_pose'], g['nodes_fixed'], g['edges'])\n"
        "bt.optimize(6)\n"
        "h = hashlib.sha256()\n"
        "for k in range(16): h.update(bt.graphs[k].store()[0].tobytes())\n"
        "st = capi.stream_stats(0)\n"
        "print(h.hexdigest(), st['pairs_measured'], bt.n_batched)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for probe in ("0", "1"):
        env = dict(os.environ, UZL_STREAM_PROBE=probe, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(r.stdout.strip().split())
    assert res[0][1] == "0"                                   # nothing measured without the probe
    assert res[0][0] == res[1][0] and res[0][2] == res[1][2] == "16"
