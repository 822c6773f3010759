"""Shared driver of the edge-filter tests: plays the same sequence of addGraph-like rounds into two filters."""
import numpy as np

from uzliti_slam_amd import synth


def drifted(scn, r, rng):
    """node poses of round r: dead reckoning blended towards ground truth (as if the optimiser had run) + jitter"""
    a = min(1.0, 0.25 * r)
    P = scn["init"].copy()
    P[:, :, 3] = (1 - a) * scn["init"][:, :, 3] + a * scn["gt"][:, :, 3] + rng.normal(0, 0.002, (len(P), 3))
    P[:, :, :3] = scn["gt"][:, :, :3] if a >= 1.0 else scn["init"][:, :, :3]
    return P


def play(filters, scn, rounds=6, seed=1, remove_frac=0.03, check=None, calc=None):
    """Each round: the graph grows by a slice of edges, every present edge is (re-)added with the current poses in key
    order (g2o_optimizer.cpp:74-87), edges deleted from the graph are removed (:89-92), then calcValidEdges +
    validEdges (:96-97).  `check(round, stage)` is called after every stage."""
    rng = np.random.default_rng(seed)
    n = len(scn["edges"])
    present = []
    cut = np.linspace(0, n, rounds + 1).astype(int)
    for r in range(rounds):
        present += list(range(cut[r], cut[r + 1]))
        poses = drifted(scn, r, rng)
        batch = [synth.edge_with_poses(scn, k, poses) for k in present]
        for f in filters:
            f.add(batch)
        if check:
            check(r, "add")
        gone = [k for k in present if rng.random() < remove_frac]
        if gone:
            present = [k for k in present if k not in set(gone)]
            keys = np.array([scn["edges"][k]["key"] for k in gone], np.uint64)
            for f in filters:
                f.remove(keys)
            if check:
                check(r, "remove")
        for f in filters:
            (calc or (lambda x: x.calc_valid_edges()))(f)
        if check:
            check(r, "calc")


def assert_same_state(a, b, with_eval=True, tag=""):
    assert len(a) == len(b), (tag, len(a), len(b))
    for i, (x, y) in enumerate(zip(a, b)):
        for f in ("uid", "from_start_ns", "from_end_ns", "to_start_ns", "to_end_ns", "size", "consensus", "changed", "evaluations"):
            assert x[f] == y[f], (tag, i, f, x[f], y[f])
        assert np.array_equal(x["keys"], y["keys"]), (tag, i)
        assert np.array_equal(x["valid"], y["valid"]), (tag, i)
        if with_eval and "P" in x and "P" in y:
            assert x["P"].tobytes() == y["P"].tobytes() and x["Q"].tobytes() == y["Q"].tobytes(), (tag, i, "points differ")
            assert x["T"].tobytes() == y["T"].tobytes(), (tag, i, "transform differs")
            assert x["ransac_consensus"] == y["ransac_consensus"], (tag, i)


def replay_filter_fixture(z, filt, calc=None, state=None):
    """Plays tests/golden/filter_*.npz into a filter object (oracle.Filter, capi.Filter or np_reference.FilterRef);
    yields per round dict(evaluated, valid_keys, clusters)."""
    filt.set_sensors(z["sensors"])
    n = len(z["key"])
    for r in range(z["present"].shape[0]):
        P = z["poses"][r]
        batch = []
        for k in np.nonzero(z["present"][r])[0]:
            a, b = int(z["node_from"][k]), int(z["node_to"][k])
            batch.append(dict(key=int(z["key"][k]), matching_score=float(z["score"][k]), valid=int(z["valid"][k]),
                              sensor_from=int(z["sensor_from"][k]), sensor_to=int(z["sensor_to"][k]),
                              stamps_from=z["stamps"][a, :z["n_stamps"][a]].copy(), stamps_to=z["stamps"][b, :z["n_stamps"][b]].copy(),
                              transform=z["transform"][k], displacement_from=z["displacement_from"][k],
                              displacement_to=z["displacement_to"][k], pose_from=P[a], pose_to=P[b]))
        filt.add(batch)
        gone = z["key"][np.nonzero(z["removed"][r])[0]]
        if len(gone):
            filt.remove(gone)
        ev = calc(filt) if calc else filt.calc_valid_edges()
        yield dict(evaluated=ev, valid_keys=np.asarray(filt.valid_edges(), np.uint64),
                   clusters=(state(filt) if state else filt.clusters()))


def check_against_filter_fixture(z, got):
    for r, a in enumerate(got):
        assert a["evaluated"] == int(z[f"r{r}_evaluated"]), r
        assert np.array_equal(a["valid_keys"], z[f"r{r}_valid_keys"]), r
        info = np.array([[c["uid"], c["size"], c["consensus"], c["changed"], c["evaluations"], c["from_start_ns"], c["from_end_ns"],
                          c["to_start_ns"], c["to_end_ns"]] for c in a["clusters"]], np.int64).reshape(-1, 9)
        assert np.array_equal(info, z[f"r{r}_cluster_info"]), r
        assert np.array_equal(np.concatenate([c["keys"] for c in a["clusters"]] + [np.zeros(0, np.uint64)]), z[f"r{r}_cluster_keys"]), r
        assert np.array_equal(np.concatenate([c["valid"] for c in a["clusters"]] + [np.zeros(0, np.uint8)]), z[f"r{r}_cluster_valid"]), r
