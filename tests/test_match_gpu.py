"""GPU parity tests of the edge-estimation half: HIP path (through the C ABI) vs the CPU oracle.
Integer work (2-NN indices/distances, correspondence order, RANSAC votes, inlier sets) must be
bit-exact; the pose/mse/information are float results of the same operation sequence and must
also match bit for bit."""
import numpy as np
import pytest

from uzliti_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher(capi):
    m = capi.Match(ransac_threshold=0.1, ransac_iteration=100, ransac_break_percentage=0.6, seed=777)
    yield m
    m.close()


def _add(m, f):
    return m.add_frame(f["desc"], f["pos"], f["valid"], f["feature_type"], f["sensor_frame"])


@pytest.mark.parametrize("nq,nt,nbytes", [(1000, 1000, 32), (300, 300, 64), (257, 513, 32), (7, 9, 32),
                                          (64, 1, 32), (5, 0, 32), (100, 100, 20), (3000, 1500, 32),
                                          # tile boundaries of the matrix-core kernel: 32 train rows / 32 queries per MFMA tile,
                                          # 128 (W = 8) or 64 (W = 16) train rows per LDS chunk, 256 / 128 queries per workgroup
                                          (33, 31, 32), (129, 127, 32), (256, 128, 32), (255, 129, 64), (1, 2, 32), (128, 65, 64),
                                          # the matrix-core kernel carries 12 index bits in its sort key: train sets beyond 4096 rows are swept
                                          # in pieces whose winners are merged (ties across the seam: the lower index must still win)
                                          (70, 4096, 32), (70, 4097, 32), (200, 9000, 32), (65, 4100, 64)])
def test_knn2_bit_exact(matcher, oracle, nq, nt, nbytes):
    rng = np.random.default_rng(nq * 7919 + nt)
    q = rng.integers(0, 256, (nq, nbytes), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, nbytes), dtype=np.uint8)
    if nq >= 8 and nt >= 8:           # sparse and dense rows: |t| - 2 <t, q> (what the accumulators hold before |q| is added) goes negative
        t[3] = 0; t[4] = 255; q[2] = 255; q[3] = 0; q[4] = t[4]; q[5] = 0; q[5, 1] = 1
    if nt > 4200:
        t[4099] = t[2]; t[4095] = t[6]; t[4096] = t[6]; q[6] = t[6]      # duplicates on both sides of a sweep boundary
    if nt >= 8:                       # deliberate ties and exact duplicates
        t[5] = t[2]; t[7] = t[2]
        t[nt - 1] = t[2]              # the same tie again in the last tile / chunk: lower index must still win
        q[0] = t[2]
        q[1] = t[2]; q[1, 0] ^= 1
    ff = matcher.add_frame(t, np.zeros((3, nt)), np.ones(nt, np.uint8))
    ft = matcher.add_frame(q, np.zeros((3, nq)), np.ones(nq, np.uint8))
    got = matcher.knn2(ff, ft, nq)
    want = oracle.knn2(q, t)
    for g, w, name in zip(got, want, ("idx0", "d0", "idx1", "d1")):
        assert np.array_equal(g, w), name


def _oracle_pair(oracle, f, t, cfg, job_id):
    return oracle.estimate_edge([f], [t], ransac_threshold=cfg["thr"], ransac_iteration=cfg["it"],
                                break_percentage=cfg["bp"], do_prosac=cfg.get("prosac", True),
                                seed=cfg["seed"], job_id=job_id)


def _compare(res, diag, j, want):
    r = res[j]
    assert r["ok"] == want["ok"]
    assert r["n_matches"] == want["n_matches"]
    assert r["n_corr"] == want["n_corr"]
    m = want["n_corr"]
    if diag is not None:
        assert np.array_equal(diag["corr_query"][j, :m], want["corr_query"])
        assert np.array_equal(diag["corr_train"][j, :m], want["corr_train"])
        assert np.array_equal(diag["corr_dist"][j, :m], want["corr_dist"])
        assert np.array_equal(diag["mask"][j, :m], want["mask"]), "inlier set differs"
    assert r["iterations_run"] == want["iterations_run"]
    assert r["best_iteration"] == want["best_iteration"]
    assert r["consensus"] == want["consensus"]
    assert np.array_equal(r["T"].reshape(3, 4), want["T"]), "pose not bit-identical"
    assert r["mse"] == want["mse"]
    assert np.array_equal(r["information"].reshape(6, 6), want["information"])


@pytest.mark.parametrize("cfg", [dict(thr=0.1, it=100, bp=0.6, seed=777),
                                 dict(thr=0.1, it=500, bp=1.0, seed=1),
                                 dict(thr=0.05, it=300, bp=0.3, seed=5),
                                 dict(thr=0.2, it=64, bp=0.6, seed=9, prosac=False)])
def test_estimate_bit_exact_vs_oracle(matcher, oracle, cfg):
    matcher.set_config(ransac_threshold=cfg["thr"], ransac_iteration=cfg["it"], ransac_break_percentage=cfg["bp"],
                       seed=cfg["seed"], do_prosac=1 if cfg.get("prosac", True) else 0)
    pairs = synth.make_pairs(12, n_kp=400, seed=cfg["seed"] + 100)
    ids = [(_add(matcher, f), _add(matcher, t)) for f, t, _ in pairs]
    job_ids = [1000 + 3 * j for j in range(len(pairs))]
    res, diag = matcher.estimate(ids, job_ids=job_ids, max_corr=400)
    n_ok = 0
    for j, (f, t, T) in enumerate(pairs):
        want = _oracle_pair(oracle, f, t, cfg, job_ids[j])
        _compare(res, diag, j, want)
        assert res[j]["frame_from"] == ids[j][0] and res[j]["frame_to"] == ids[j][1]
        n_ok += int(want["consensus"] > 20)
        if want["consensus"] > 50:        # sanity: the estimate is the true motion
            assert np.abs(res[j]["T"].reshape(3, 4) - T).max() < 0.05
    assert n_ok >= len(pairs) // 2


def test_estimate_full_size_c3_shape(matcher, oracle):
    """BASELINE config 3 shape (1000 x 1000 x 256 bit, 500 hypotheses) on a few pairs, bit-exact."""
    cfg = dict(thr=0.1, it=500, bp=1.0, seed=777)
    matcher.set_config(ransac_threshold=0.1, ransac_iteration=500, ransac_break_percentage=1.0, seed=777, do_prosac=1)
    pairs = synth.make_pairs(4, n_kp=1000, seed=777)
    ids = [(_add(matcher, f), _add(matcher, t)) for f, t, _ in pairs]
    res, diag = matcher.estimate(ids, max_corr=1000)
    for j, (f, t, T) in enumerate(pairs):
        _compare(res, diag, j, _oracle_pair(oracle, f, t, cfg, j))


def test_edge_cases(matcher, oracle):
    matcher.set_config(ransac_threshold=0.1, ransac_iteration=50, ransac_break_percentage=0.6, seed=3, do_prosac=1)
    rng = np.random.default_rng(11)
    (f, t, _), = synth.make_pairs(1, n_kp=200, seed=5)
    cfg = dict(thr=0.1, it=50, bp=0.6, seed=3)
    cases = []
    # (a) too few keypoints on one side (< 7): no sensor pair -> ok = 0 (estimator.cpp:47,93)
    small = dict(desc=t["desc"][:6], pos=t["pos"][:, :6], valid=t["valid"][:6], feature_type=2, sensor_frame=0)
    cases.append(([f], [small]))
    # (b) different feature_type -> not matched
    other = dict(t); other["feature_type"] = 3
    cases.append(([f], [other]))
    # (c) different sensor frame -> not matched
    other2 = dict(t); other2["sensor_frame"] = 4
    cases.append(([f], [other2]))
    # (d) nothing valid in 3-D -> M = 0 -> ok = 0
    inval = dict(t); inval["valid"] = np.zeros_like(t["valid"])
    cases.append(([f], [inval]))
    # (e) two FeatureData per node: the pair with more ratio-test survivors wins, first wins ties
    noise = dict(desc=rng.integers(0, 256, t["desc"].shape, dtype=np.uint8), pos=t["pos"], valid=t["valid"],
                 feature_type=2, sensor_frame=0)
    cases.append(([f, noise], [noise, t]))
    cases.append(([f, f], [t, t]))
    # (f) pure noise descriptors: few matches, RANSAC finds no consensus of 3
    cases.append(([noise], [dict(noise, desc=rng.integers(0, 256, t["desc"].shape, dtype=np.uint8))]))
    jobs = []
    for fr, to in cases:
        jobs.append(([_add(matcher, x) for x in fr], [_add(matcher, x) for x in to]))
    res, diag = matcher.estimate(jobs, max_corr=200)
    for j, (fr, to) in enumerate(cases):
        want = oracle.estimate_edge(fr, to, ransac_threshold=0.1, ransac_iteration=50, break_percentage=0.6,
                                    do_prosac=True, seed=3, job_id=j)
        _compare(res, diag, j, want)
        if want["frame_from"] >= 0:
            assert res[j]["frame_from"] == jobs[j][0][want["frame_from"]]
            assert res[j]["frame_to"] == jobs[j][1][want["frame_to"]]
        else:
            assert res[j]["frame_from"] == -1
    assert [int(r["ok"]) for r in res[:4]] == [0, 0, 0, 0]
    assert res[4]["ok"] == 1 and res[5]["ok"] == 1
    # empty batch
    res0, _ = matcher.estimate([])
    assert len(res0) == 0


def test_ransac_points_matches_oracle(matcher, oracle):
    """estimateSVD twin used by TransformationFilter (transformation_filter.cpp:272-275): 200 its, no PROSAC."""
    rng = np.random.default_rng(3)
    probs = []
    for m in (3, 5, 10, 37, 100, 2, 0):
        P = rng.normal(size=(3, m)) * 2
        R = synth.quat_to_R(synth.quat_from_rotvec(rng.normal(size=3) * 0.3)); t = rng.normal(size=3)
        Q = R @ P + t[:, None] + rng.normal(0, 0.02, (3, m))
        if m >= 10:
            Q[:, ::4] += rng.normal(0, 2.0, Q[:, ::4].shape)
        probs.append((P, Q))
    got = matcher.ransac_points(probs, 0.3, 200, 0.6, do_prosac=False, job_ids=list(range(50, 50 + len(probs))))
    for b, (P, Q) in enumerate(probs):
        want = oracle.prosac(P, Q, 0.3, 200, 0.6, do_prosac=False, seed=matcher.cfg.seed, job_id=50 + b)
        assert got[b]["consensus"] == want["consensus"]
        assert got[b]["iterations_run"] == want["iterations_run"]
        assert np.array_equal(got[b]["mask"], want["mask"])
        assert np.array_equal(got[b]["T"], want["T"])
        assert got[b]["mse"] == want["mse"]


def test_large_frame_global_tile(matcher, oracle):
    """A frame too large for the LDS correspondence tile takes the HBM-scratch path; same result."""
    matcher.set_config(ransac_threshold=0.1, ransac_iteration=64, ransac_break_percentage=1.0, seed=2, do_prosac=1)
    (f, t, T), = synth.make_pairs(1, n_kp=3000, seed=21, outlier_frac=0.2)
    ids = [(_add(matcher, f), _add(matcher, t))]
    res, diag = matcher.estimate(ids, max_corr=3000)
    want = oracle.estimate_edge([f], [t], ransac_threshold=0.1, ransac_iteration=64, break_percentage=1.0,
                                do_prosac=True, seed=2, job_id=0)
    _compare(res, diag, 0, want)


def test_errors(capi, matcher):
    with pytest.raises(capi.UzlError) as e:
        matcher.estimate([(99999, 0)])
    assert e.value.status == capi.UZL_ERR_NOT_FOUND
    with pytest.raises(capi.UzlError) as e:
        matcher.add_frame(np.zeros((4, 30), np.uint8), np.zeros((3, 4)), np.ones(4, np.uint8))
    assert e.value.status == capi.UZL_ERR_BAD_ARG
    with pytest.raises(capi.UzlError):
        matcher.set_config(ransac_iteration=0)
    matcher.set_config(ransac_iteration=100)


def test_c3_full_batch_properties(capi, oracle):
    """BASELINE config 3 at its size: 512 node pairs x 1000 ORB-256 descriptors, 500 hypotheses, one batch.  The oracle needs ~2 ms per
    pair: all 512 are compared bit for bit; and properties: results do not depend on
    how the pairs are batched, the estimated motion is the one the frames were generated with, inliers are the planted ones."""
    cfg = dict(ransac_threshold=0.1, ransac_iteration=500, ransac_break_percentage=1.0, do_prosac=1, seed=777)
    pairs = synth.make_pairs(512, n_kp=1000, desc_bytes=32, seed=777)
    m = capi.Match(**cfg)
    ids = [(m.add_frame(f["desc"], f["pos"], f["valid"]), m.add_frame(t["desc"], t["pos"], t["valid"])) for f, t, _ in pairs]
    res, diag = m.estimate(ids, max_corr=1000)
    assert res["ok"].all() and (res["iterations_run"] == 500).all()
    # (a) batching does not matter: four batches of 128 with the same job ids give the same bytes
    parts = [m.estimate(ids[k:k + 128], job_ids=list(range(k, k + 128)))[0] for k in range(0, 512, 128)]
    again = np.concatenate(parts)
    for f in ("consensus", "n_corr", "n_matches", "best_iteration", "mse", "T", "information"):
        assert np.array_equal(res[f], again[f]), f
    # (b) ALL 512 pairs against the oracle, bit for bit (the CPU checker does ~370 pairs/s: 1.4 s)
    for j in range(512):
        f, t, _ = pairs[j]
        w = oracle.estimate_edge([f], [t], ransac_threshold=0.1, ransac_iteration=500, break_percentage=1.0, do_prosac=True, seed=777, job_id=j)
        k = w["n_corr"]
        assert res[j]["consensus"] == w["consensus"] and res[j]["n_corr"] == k and res[j]["best_iteration"] == w["best_iteration"]
        assert np.array_equal(diag["mask"][j, :k], w["mask"]) and np.array_equal(diag["corr_query"][j, :k], w["corr_query"])
        assert np.array_equal(res[j]["T"].reshape(3, 4), w["T"]) and res[j]["mse"] == w["mse"]
    # (c) the motion is recovered (1 cm point noise, ~480 inliers): translation within 1 cm, rotation within 0.2 degrees
    Tgt = np.array([p[2] for p in pairs])
    dt, dr = synth.pose_errors(res["T"].reshape(-1, 3, 4), Tgt)
    assert dt < 0.01 and dr < np.deg2rad(0.2), (dt, dr)
    # (d) consensus = correspondences whose residual under the returned transform is below the threshold (recount in numpy, 1e-9 band)
    for j in range(0, 512, 37):
        k = res[j]["n_corr"]
        f, t, _ = pairs[j]
        P = t["pos"][:, diag["corr_query"][j, :k]]; Q = f["pos"][:, diag["corr_train"][j, :k]]
        T = res[j]["T"].reshape(3, 4)
        d = np.linalg.norm(T[:, :3] @ P + T[:, 3:4] - Q, axis=0)
        assert np.array_equal(diag["mask"][j, :k][np.abs(d - 0.1) > 1e-9] != 0, (d < 0.1)[np.abs(d - 0.1) > 1e-9])
    m.close()


def test_deployed_operating_point_all_pairs(capi, oracle):
    """The estimator as the reference deploys it: BRISK-512 (64-byte descriptors), 300 keypoints per frame
    (feature_extraction_service_node.cpp:63-66), ransac_threshold 0.1 / 100 iterations (iti_slam_launch/yaml/slam.yaml:34-38), early exit
    at 60 % consensus (cfg/FeatureLinkEstimation.cfg:12): 512 pairs, every one against the oracle bit for bit - `iterations_run` included,
    i.e. the early exit fires in the same iteration."""
    cfg = dict(ransac_threshold=0.1, ransac_iteration=100, ransac_break_percentage=0.6, do_prosac=1, seed=777)
    pairs = synth.make_pairs(512, n_kp=300, desc_bytes=64, seed=4242)
    m = capi.Match(**cfg)
    ids = [(m.add_frame(f["desc"], f["pos"], f["valid"]), m.add_frame(t["desc"], t["pos"], t["valid"])) for f, t, _ in pairs]
    res, diag = m.estimate(ids, max_corr=300)
    assert res["ok"].all()
    assert (res["iterations_run"] < 100).mean() > 0.5          # the early exit is what this operating point is about
    for j in range(512):
        f, t, _ = pairs[j]
        w = oracle.estimate_edge([f], [t], ransac_threshold=0.1, ransac_iteration=100, break_percentage=0.6, do_prosac=True, seed=777, job_id=j)
        k = w["n_corr"]
        assert res[j]["consensus"] == w["consensus"] and res[j]["n_corr"] == k and res[j]["n_matches"] == w["n_matches"]
        assert res[j]["iterations_run"] == w["iterations_run"] and res[j]["best_iteration"] == w["best_iteration"]
        assert np.array_equal(diag["mask"][j, :k], w["mask"]) and np.array_equal(diag["corr_query"][j, :k], w["corr_query"])
        assert np.array_equal(diag["corr_train"][j, :k], w["corr_train"])
        assert np.array_equal(res[j]["T"].reshape(3, 4), w["T"]) and res[j]["mse"] == w["mse"]
        assert np.array_equal(res[j]["information"].reshape(6, 6), w["information"])
    Tgt = np.array([p[2] for p in pairs])
    dt, dr = synth.pose_errors(res["T"].reshape(-1, 3, 4), Tgt)
    assert dt < 0.03 and dr < np.deg2rad(0.6), (dt, dr)
    m.close()


def test_random_shapes_against_oracle():
    """Randomized sweep (tests/diag/stress_match.py): frame sizes 7 .. 2700, 256- and 512-bit descriptors, outlier / validity fractions,
    thresholds, 1 .. 1000 iterations, early exit, PROSAC on / off, duplicate descriptors (2-NN ties), smaller train than query sets,
    nodes with several FeatureData of mixed sensor frames / feature types - every edge bit-exact against the oracle."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "diag", "stress_match.py"), "50", "9"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ", 0 misses" in r.stdout


def test_bulk_add_frames_equals_single_adds(capi, oracle):
    """uzl_match_add_frames (one extent, threaded packing, one DMA per staging half): the frames read back byte for byte and the
    estimates equal those over frames added one by one."""
    from uzliti_slam_amd import wire as W
    pairs = synth.make_pairs(24, n_kp=700, seed=9)
    rng = np.random.default_rng(1)
    odd = []                                                 # ragged frames: empty, one keypoint, other descriptor widths
    for n, b in [(0, 32), (1, 32), (333, 64), (5, 4)]:
        odd.append((rng.integers(0, 256, (n, b), dtype=np.uint8), rng.normal(size=(3, n)), rng.integers(0, 2, n).astype(np.uint8)))
    frames = [(f["desc"], f["pos"], f["valid"]) for p in pairs for f in p[:2]] + odd
    a = capi.Match(ransac_iteration=100, seed=3)
    b = capi.Match(ransac_iteration=100, seed=3)
    ids_a = a.add_frames(capi.Match.pack_frames(frames))
    ids_b = [b.add_frame(*f) for f in frames]
    assert len(ids_a) == len(frames) and a.frame_count() == len(frames)
    for k, (d, p, v) in enumerate(frames):
        gd, gp, gv = W.get_frame(a, ids_a[k])
        assert np.array_equal(gd, d) and np.array_equal(gp, np.asarray(p, np.float64)) and np.array_equal(gv, v)
    ra, _ = a.estimate([(ids_a[2 * k], ids_a[2 * k + 1]) for k in range(len(pairs))])
    rb, _ = b.estimate([(ids_b[2 * k], ids_b[2 * k + 1]) for k in range(len(pairs))])
    assert np.array_equal(ra["consensus"], rb["consensus"]) and np.array_equal(ra["T"], rb["T"])
    a.close(); b.close()


def test_frame_store_reclaims_removed_frames(capi):
    """Frames come and go (the reference merges and deletes nodes all the time, graph_slam_node.cpp:665-777): ten times the arena's
    initial size goes through the store, a third of it alive at any moment - the arena must not grow beyond what is alive (plus
    fragmentation), and what is alive must stay intact."""
    from uzliti_slam_amd import wire as W
    rng = np.random.default_rng(5)
    m = capi.Match()
    cap0 = m.arena_bytes()["capacity"]
    live = {}
    total = 0
    k = 0
    peak = 0
    while total < 10 * cap0:
        n = int(rng.integers(200, 3000))
        fr = [(rng.integers(0, 256, (n, 32), dtype=np.uint8), rng.normal(size=(3, n)), rng.integers(0, 2, n).astype(np.uint8)) for _ in range(16)]
        if k % 2 == 0:
            ids = m.add_frames(capi.Match.pack_frames(fr))
        else:
            ids = [m.add_frame(*f) for f in fr]
        for i, f in zip(ids, fr):
            live[i] = f
        total += sum(f[0].nbytes + 24 * len(f[2]) + len(f[2]) for f in fr)
        # keep about 20 MB alive: drop random frames
        while sum(v[0].nbytes for v in live.values()) > 20e6:
            victim = list(live.keys())[int(rng.integers(0, len(live)))]
            m.remove_frame(victim); del live[victim]
        peak = max(peak, m.arena_bytes()["live"])
        k += 1
    ab = m.arena_bytes()
    assert ab["high_water"] <= 2.5 * peak and ab["capacity"] <= 2 * cap0, (ab, peak)     # 640 MB went through; the store holds what is alive (+ holes)
    assert ab["live"] <= ab["high_water"] and m.frame_count() == len(live)
    for i in list(live.keys())[:40]:
        gd, gp, gv = W.get_frame(m, i)
        assert np.array_equal(gd, live[i][0]) and np.array_equal(gp, np.asarray(live[i][1], np.float64)) and np.array_equal(gv, live[i][2])
    for i in list(live.keys()):
        m.remove_frame(i)
    assert m.arena_bytes()["high_water"] == 0 and m.arena_bytes()["live"] == 0
    m.close()
