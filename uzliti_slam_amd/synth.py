"""Synthetic workloads for the hot path: pose graphs and descriptor frames.

Pure data generation (numpy); the same arrays feed the HIP path, the oracle and the CPU baseline.
Graph: seed 12345; descriptors: seed 777 (overridable).  Shapes follow BASELINE.json's configs:
  C1 100 n / 300 e, C2 1k n / 5k e, C4 10k n / 50k e, C5 20k n;
  C3 512 pairs x 1000 ORB-256 descriptors.
"""
import numpy as np

EDGE_TYPE_ODOM = 104    # graph_slam_msgs/Edge TYPE_2D_WHEEL_ODOMETRY
EDGE_TYPE_3D_FULL = 1   # graph_slam_msgs/Edge TYPE_3D_FULL
EDGE_TYPE_2D_LASER = 105  # graph_slam_msgs/Edge TYPE_2D_LASER
FEATURE_ORB = 2


# ----------------------------------------------------------------------------- SE(3) helpers
def quat_mul(a, b):
    """Hamilton product, (w,x,y,z), broadcasting over leading dims."""
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def quat_from_rotvec(v):
    v = np.asarray(v, np.float64)
    th = np.linalg.norm(v, axis=-1, keepdims=True)
    half = 0.5 * th
    k = np.where(th > 1e-12, np.sin(half) / np.where(th > 1e-12, th, 1.0), 0.5)
    return np.concatenate([np.cos(half), k * v], axis=-1)


def quat_to_R(q):
    q = np.asarray(q, np.float64)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - z * w); R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w); R[..., 2, 1] = 2 * (y * z + x * w); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def se3(R, t):
    """(.,3,3),(.,3) -> (.,3,4)"""
    return np.concatenate([R, np.asarray(t)[..., None]], axis=-1)


def se3_mul(A, B):
    R = A[..., :3, :3] @ B[..., :3, :3]
    t = (A[..., :3, :3] @ B[..., :3, 3:4])[..., 0] + A[..., :3, 3]
    return se3(R, t)


def se3_inv(A):
    Rt = np.swapaxes(A[..., :3, :3], -1, -2)
    t = -(Rt @ A[..., :3, 3:4])[..., 0]
    return se3(Rt, t)


def se3_from_noise(dt, drot):
    return se3(quat_to_R(quat_from_rotvec(drot)), dt)


def rotation_angle(R):
    c = np.clip((np.trace(R, axis1=-2, axis2=-1) - 1.0) * 0.5, -1.0, 1.0)
    return np.arccos(c)


def pose_errors(A, B):
    """Max translation [m] and rotation [rad] difference between two (n,3,4) pose arrays."""
    A = np.asarray(A).reshape(-1, 3, 4); B = np.asarray(B).reshape(-1, 3, 4)
    dt = np.linalg.norm(A[:, :, 3] - B[:, :, 3], axis=1)
    dR = np.swapaxes(A[:, :, :3], 1, 2) @ B[:, :, :3]
    return float(dt.max(initial=0.0)), float(rotation_angle(dR).max(initial=0.0))


# ----------------------------------------------------------------------------- pose graphs
def make_pose_graph(n_nodes, n_edges, seed=12345, outlier_frac=0.05):
    """SURVEY §8d 'Synthetic graphs'.  Returns a dict in the reference's data model
    (SlamNode / SlamEdge fields as arrays), before G1 flattening:
      nodes_pose (N,12) initial poses = odometry dead reckoning, nodes_fixed (N) [node 0 fixed],
      gt_pose (N,12), edges {from,to,type,sensor_from,sensor_to,valid,transform,displacement_from,
      displacement_to,information}."""
    rng = np.random.default_rng(seed)
    N = int(n_nodes)
    n_loop = int(n_edges) - (N - 1)
    assert n_loop >= 0
    # --- ground-truth trajectory: 0.3 m steps, +-10 deg heading noise, kept inside a box so that
    # places are revisited (loop closures need pairs within 1.5 m with |i-j| > 20)
    side = max(4.0, np.sqrt(N / 3.4))
    half = 0.5 * side
    xy = np.zeros((N, 2)); yaw = np.zeros(N); z = np.zeros(N); roll = np.zeros(N); pitch = np.zeros(N)
    th = rng.uniform(-np.pi, np.pi)
    dth = rng.normal(0.0, np.deg2rad(10.0), N)
    dz = rng.normal(0.0, 0.005, N); dr = rng.normal(0.0, np.deg2rad(0.3), N); dp = rng.normal(0.0, np.deg2rad(0.3), N)
    for i in range(1, N):
        th = th + dth[i]
        nxt = xy[i - 1] + 0.3 * np.array([np.cos(th), np.sin(th)])
        if abs(nxt[0]) > half or abs(nxt[1]) > half:
            th = np.arctan2(-xy[i - 1, 1], -xy[i - 1, 0]) + dth[i]   # steer back towards the centre
            nxt = xy[i - 1] + 0.3 * np.array([np.cos(th), np.sin(th)])
        xy[i] = nxt; yaw[i] = th
        z[i] = 0.98 * z[i - 1] + dz[i]
        roll[i] = 0.95 * roll[i - 1] + dr[i]; pitch[i] = 0.95 * pitch[i - 1] + dp[i]
    yaw[0] = yaw[1] if N > 1 else 0.0
    qz = quat_from_rotvec(np.stack([np.zeros(N), np.zeros(N), yaw], 1))
    qy = quat_from_rotvec(np.stack([np.zeros(N), pitch, np.zeros(N)], 1))
    qx = quat_from_rotvec(np.stack([roll, np.zeros(N), np.zeros(N)], 1))
    gt = se3(quat_to_R(quat_mul(quat_mul(qz, qy), qx)), np.stack([xy[:, 0], xy[:, 1], z], 1))

    # --- odometry edges i -> i+1 (graph_slam_node.cpp:306-335): dt = 1
    rel = se3_mul(se3_inv(gt[:-1]), gt[1:])
    odo = se3_mul(rel, se3_from_noise(rng.normal(0, 0.02, (N - 1, 3)), rng.normal(0, 0.001, (N - 1, 3))))
    odom_info = np.zeros((6, 6))
    odom_info[:3, :3] = np.eye(3) / (0.02 ** 2)
    odom_info[3:, 3:] = np.eye(3) / (0.02 ** 2 * 0.05 ** 2)

    # --- loop closures: pairs within 1.5 m and |i-j| > 20
    cand = _close_pairs(gt[:, :, 3], 1.5, 20) if n_loop > 0 else np.zeros((0, 2), np.int64)
    radius = 1.5
    while len(cand) < n_loop and radius < 50:
        radius *= 1.5
        cand = _close_pairs(gt[:, :, 3], radius, 20 if N > 40 else 1)
    if n_loop == 0:
        sel = np.zeros(0, np.int64)
    elif len(cand) >= n_loop:
        sel = rng.choice(len(cand), size=n_loop, replace=False)
    else:                                    # tiny graphs: allow multi-edges
        sel = rng.choice(len(cand), size=n_loop, replace=True)
    sel.sort()
    pairs = cand[sel]
    # orient randomly (the reference estimates from close_node/older to current, either direction occurs)
    flip = rng.random(n_loop) < 0.5
    lf = np.where(flip, pairs[:, 1], pairs[:, 0]); lt = np.where(flip, pairs[:, 0], pairs[:, 1])
    lrel = se3_mul(se3_inv(gt[lf]), gt[lt])
    lz = se3_mul(lrel, se3_from_noise(rng.normal(0, 0.05, (n_loop, 3)), rng.normal(0, 0.01, (n_loop, 3))))
    outl = rng.random(n_loop) < outlier_frac
    gross = se3_from_noise(rng.uniform(-2, 2, (n_loop, 3)), rng.uniform(-0.5, 0.5, (n_loop, 3)))
    lz = np.where(outl[:, None, None], se3_mul(lrel, gross), lz)
    c = rng.uniform(20, 300, n_loop); m = rng.uniform(0.02, 0.08, n_loop)
    linfo = np.zeros((n_loop, 6, 6))
    s = 0.1 * c / m                                               # feature_transformation_estimator.cpp:133-137
    for k in range(3):
        linfo[:, k, k] = s; linfo[:, 3 + k, 3 + k] = 100.0 * s

    # --- initial poses: dead reckoning along the odometry chain
    init = np.empty_like(gt); init[0] = gt[0]
    for i in range(1, N):
        init[i] = se3_mul(init[i - 1], odo[i - 1])

    E = (N - 1) + n_loop
    I12 = np.tile(np.eye(3, 4).reshape(1, 12), (E, 1))
    edges = dict(
        **{"from": np.concatenate([np.arange(N - 1), lf]).astype(np.int32)},
        to=np.concatenate([np.arange(1, N), lt]).astype(np.int32),
        type=np.concatenate([np.full(N - 1, EDGE_TYPE_ODOM), np.full(n_loop, EDGE_TYPE_3D_FULL)]).astype(np.int32),
        sensor_from=np.full(E, -1, np.int32), sensor_to=np.full(E, -1, np.int32),
        valid=np.ones(E, np.int32),
        transform=np.concatenate([odo.reshape(-1, 12), lz.reshape(-1, 12)]),
        displacement_from=I12.copy(), displacement_to=I12.copy(),
        information=np.concatenate([np.tile(odom_info.reshape(1, 36), (N - 1, 1)), linfo.reshape(-1, 36)]),
        diff_time=np.concatenate([np.full(N - 1, 0.5), np.zeros(n_loop)]),      # SlamEdge::diff_time_ [s] of the odometry edges
    )
    fixed = np.zeros(N, np.uint8); fixed[0] = 1
    return dict(nodes_pose=init.reshape(N, 12), nodes_fixed=fixed, gt_pose=gt.reshape(N, 12), edges=edges,
                n_outliers=int(outl.sum()))


def _close_pairs(pos, radius, min_sep):
    """All (i<j) with ||p_i - p_j|| < radius and j - i > min_sep, sorted lexicographically."""
    from scipy.spatial import cKDTree
    pos = np.ascontiguousarray(pos, np.float64)
    pr = cKDTree(pos).query_pairs(radius, output_type="ndarray").astype(np.int64)
    if pr.size == 0:
        return np.zeros((0, 2), np.int64)
    pr = np.sort(pr, axis=1)
    pr = pr[(pr[:, 1] - pr[:, 0]) > min_sep]
    order = np.lexsort((pr[:, 1], pr[:, 0]))
    return pr[order]


# ----------------------------------------------------------------------------- descriptor frames
def make_pair(rng, n_kp=1000, desc_bytes=32, flip_p=0.08, outlier_frac=0.4, sigma=0.01, invalid_frac=0.1,
              max_t=1.0, max_rot_deg=20.0, T=None):
    """SURVEY §8d 'Synthetic descriptors': one node pair (frame `from`, frame `to`).
    Returns (frame_from, frame_to, T_from_to (3,4)) with frames as dicts
    {desc (n,bytes) u8, pos (3,n) f64, valid (n) u8, feature_type, sensor_frame}."""
    bits = desc_bytes * 8
    # landmarks in a 6 x 4 x 3 m box in front of camera A (z forward)
    lm = np.stack([rng.uniform(-3, 3, n_kp), rng.uniform(-2, 2, n_kp), rng.uniform(0.5, 3.5, n_kp)], 0)
    dbits = rng.integers(0, 2, (n_kp, bits), dtype=np.uint8)
    # relative motion: from_T_to (random inside the acceptance gate unless the caller gives it)
    if T is None:
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = np.deg2rad(rng.uniform(0, max_rot_deg))
        R = quat_to_R(quat_from_rotvec(ax * ang))
        tv = rng.normal(size=3); tv *= rng.uniform(0, max_t) / np.linalg.norm(tv)
        T = se3(R, tv)
    else:
        T = np.asarray(T, np.float64).reshape(3, 4)
        R = T[:, :3]; tv = T[:, 3]

    def noisy_bits():
        return dbits ^ (rng.random((n_kp, bits)) < flip_p).astype(np.uint8)

    pos_from = lm + rng.normal(0, sigma, lm.shape)
    desc_from = np.packbits(noisy_bits(), axis=1)
    # points seen from `to`: p_to = T^-1 * p_from  (T maps to-frame points into the from frame)
    pos_to = R.T @ (lm - tv[:, None]) + rng.normal(0, sigma, lm.shape)
    bits_to = noisy_bits()
    # 40 % of `to` rows replaced by fresh random descriptors + random 3-D points
    out = rng.random(n_kp) < outlier_frac
    n_out = int(out.sum())
    bits_to[out] = rng.integers(0, 2, (n_out, bits), dtype=np.uint8)
    pos_to[:, out] = np.stack([rng.uniform(-3, 3, n_out), rng.uniform(-2, 2, n_out), rng.uniform(0.5, 3.5, n_out)], 0)
    desc_to = np.packbits(bits_to, axis=1)
    # shuffle the row order of `to` so that trainIdx != queryIdx
    perm = rng.permutation(n_kp)
    desc_to = desc_to[perm]; pos_to = pos_to[:, perm]
    valid_from = (rng.random(n_kp) >= invalid_frac).astype(np.uint8)
    valid_to = (rng.random(n_kp) >= invalid_frac).astype(np.uint8)
    pos_from = pos_from.copy(); pos_to = pos_to.copy()
    pos_from[2, valid_from == 0] = -1.0          # z = -1 when invalid (feature_extraction_core.cpp:286-289)
    pos_to[2, valid_to == 0] = -1.0
    f = dict(desc=np.ascontiguousarray(desc_from), pos=np.ascontiguousarray(pos_from), valid=valid_from,
             feature_type=FEATURE_ORB, sensor_frame=0)
    t = dict(desc=np.ascontiguousarray(desc_to), pos=np.ascontiguousarray(pos_to), valid=valid_to,
             feature_type=FEATURE_ORB, sensor_frame=0)
    return f, t, T


def make_pairs(n_pairs, n_kp=1000, desc_bytes=32, seed=777, **kw):
    rng = np.random.default_rng(seed)
    return [make_pair(rng, n_kp=n_kp, desc_bytes=desc_bytes, **kw) for _ in range(n_pairs)]


def make_filter_scenario(n_nodes=400, n_loop=1200, seed=4242, outlier_frac=0.2, two_stamp_frac=0.1):
    """Synthetic input of the edge filter (TransformationFilter, transformation_filter.cpp:138-291): the loop
    closures of make_pose_graph as SlamEdge-like dicts, plus node stamps, drifting node poses and two sensors.
    Returns dict(edges=[...edge dicts without poses...], stamps=[int64 arrays per node], gt (N,3,4), init (N,3,4),
    sensors (2,12)).  `edge_with_poses(scn, k, poses)` makes the dict uzl_filter_add takes."""
    g = make_pose_graph(n_nodes, (n_nodes - 1) + n_loop, seed=seed, outlier_frac=outlier_frac)
    rng = np.random.default_rng(seed + 1)
    N = n_nodes
    t0 = 1_400_000_000 * 10**9
    stamps = []
    for i in range(N):
        t = t0 + int(0.5e9 * i) + int(rng.integers(0, 10**7))
        stamps.append(np.array([t, t + 250_000_000], np.int64) if rng.random() < two_stamp_frac else np.array([t], np.int64))
    sensors = np.stack([se3(quat_to_R(quat_from_rotvec(np.array([[0.0, 0.1, 0.0]])))[0], np.array([0.2, 0.0, 0.5])),
                        se3(quat_to_R(quat_from_rotvec(np.array([[0.0, 0.0, 1.5]])))[0], np.array([-0.1, 0.05, 0.4]))])
    E = g["edges"]
    loop = np.nonzero(E["type"] != EDGE_TYPE_ODOM)[0]
    ident = np.eye(3, 4)
    edges = []
    for n, k in enumerate(loop):
        sf, st = int(rng.integers(-1, 2)), int(rng.integers(-1, 2))
        df = se3_from_noise(rng.normal(0, 0.05, (1, 3)), rng.normal(0, 0.05, (1, 3)))[0] if rng.random() < 0.3 else ident
        dt = se3_from_noise(rng.normal(0, 0.05, (1, 3)), rng.normal(0, 0.05, (1, 3)))[0] if rng.random() < 0.3 else ident
        Sf = sensors[sf] if sf >= 0 else ident
        St = sensors[st] if st >= 0 else ident
        Z = E["transform"][k].reshape(3, 4)
        # node-to-node measurement Z = disp_from * S_from * T * S_to^-1 * disp_to^-1  (g2o_optimizer.cpp:281)
        T = se3_mul(se3_mul(se3_inv(se3_mul(df, Sf)), Z), se3_mul(dt, St))
        edges.append(dict(key=1000 + 7 * n, matching_score=float(rng.integers(20, 60)), valid=int(rng.random() < 0.15),
                          sensor_from=sf, sensor_to=st, node_from=int(E["from"][k]), node_to=int(E["to"][k]),
                          transform=T.reshape(12), displacement_from=np.asarray(df).reshape(12),
                          displacement_to=np.asarray(dt).reshape(12)))
    for n, k in enumerate(loop):
        edges[n]["graph_edge"] = int(k)
    return dict(edges=edges, stamps=stamps, gt=g["gt_pose"].reshape(N, 3, 4), init=g["nodes_pose"].reshape(N, 3, 4),
                sensors=sensors.reshape(2, 12), graph=g)


def edge_with_poses(scn, k, poses):
    """Edge k of a filter scenario as uzl_filter_add wants it, with the end nodes' current poses and stamps."""
    e = dict(scn["edges"][k])
    e["stamps_from"] = scn["stamps"][e["node_from"]]; e["stamps_to"] = scn["stamps"][e["node_to"]]
    e["pose_from"] = np.asarray(poses[e["node_from"]]).reshape(12); e["pose_to"] = np.asarray(poses[e["node_to"]]).reshape(12)
    return e


def make_slam_run(n_nodes=150, n_landmarks=1500, seed=2024, max_pairs=400, desc_bytes=32, flip_p=0.06, clutter=0.3,
                  sigma=0.01, invalid_frac=0.1, alias_frac=0.08):
    """A small end-to-end run in the shape of BASELINE config 5: a robot revisits places; every node carries one
    FeatureData frame of the world landmarks it sees (camera frame, metres), loop-closure candidates are node pairs
    within 1 m (what SlamGraph::getNodesWithinRadius would hand to the estimator, slam_graph.cpp:266-278).
    Returns dict(gt, init (N,3,4), stamps, odo edges (make_pose_graph layout), frames [dict], sensor (12),
    pairs [(from, to, frame_from, frame_to)] ordered by the later node = the order they would be produced online;
    for a fraction `alias_frac` of the pairs the `to` frame is the one of a node 5 steps away (a wrong but
    geometrically consistent match: the kind of edge the filter and the robust kernel exist for))."""
    g = make_pose_graph(n_nodes, n_nodes - 1, seed=seed)
    rng = np.random.default_rng(seed + 7)
    N = n_nodes
    gt = g["gt_pose"].reshape(N, 3, 4)
    # base -> camera: camera z = base x (forward), camera x = -base y, camera y = -base z
    S = se3(np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]]), np.array([0.1, 0.0, 0.3]))
    half = 0.5 * max(4.0, np.sqrt(N / 3.4)) + 2.5
    lm = np.stack([rng.uniform(-half, half, n_landmarks), rng.uniform(-half, half, n_landmarks), rng.uniform(-0.5, 2.5, n_landmarks)], 0)
    bits = desc_bytes * 8
    lbits = rng.integers(0, 2, (n_landmarks, bits), dtype=np.uint8)
    frames = []
    for i in range(N):
        C = se3_mul(gt[i], S)                                     # world <- camera
        pc = C[:, :3].T @ (lm - C[:, 3:4])
        vis = (pc[2] > 0.3) & (pc[2] < 4.0) & (np.abs(pc[0]) < pc[2]) & (np.abs(pc[1]) < 0.8 * pc[2])
        idx = np.nonzero(vis)[0]
        n_c = int(clutter * len(idx)) + 8
        d = np.concatenate([lbits[idx] ^ (rng.random((len(idx), bits)) < flip_p).astype(np.uint8),
                            rng.integers(0, 2, (n_c, bits), dtype=np.uint8)])
        p = np.concatenate([pc[:, idx] + rng.normal(0, sigma, (3, len(idx))),
                            np.stack([rng.uniform(-2, 2, n_c), rng.uniform(-1.5, 1.5, n_c), rng.uniform(0.3, 4.0, n_c)])], 1)
        perm = rng.permutation(d.shape[0])
        d = d[perm]; p = np.ascontiguousarray(p[:, perm])
        valid = (rng.random(d.shape[0]) >= invalid_frac).astype(np.uint8)
        p[2, valid == 0] = -1.0
        frames.append(dict(desc=np.ascontiguousarray(np.packbits(d, axis=1)), pos=p, valid=valid, feature_type=FEATURE_ORB, sensor_frame=0))
    t0 = 1_400_000_000 * 10**9
    stamps = [np.array([t0 + int(0.5e9 * i)], np.int64) for i in range(N)]
    pos = gt[:, :, 3]
    pairs = []
    for j in range(N):
        dd = np.linalg.norm(pos[:j] - pos[j], axis=1) if j else np.zeros(0)
        for i in np.nonzero(dd < 1.0)[0]:
            if j - i > 10 and rotation_angle(gt[i][:, :3].T @ gt[j][:, :3]) < np.deg2rad(50):
                pairs.append((int(i), int(j)) if rng.random() < 0.5 else (int(j), int(i)))
    if len(pairs) > max_pairs:
        keep = np.sort(rng.choice(len(pairs), max_pairs, replace=False))
        pairs = [pairs[k] for k in keep]
    full = []
    for a, b in pairs:
        fb = b
        if rng.random() < alias_frac:
            fb = b + 5 if b + 5 < N else b - 5
        full.append((a, b, a, fb))
    pairs = full
    return dict(gt=gt, init=g["nodes_pose"].reshape(N, 3, 4), fixed=g["nodes_fixed"], stamps=stamps, odo=g["edges"], frames=frames,
                sensor=S.reshape(12), pairs=pairs)


def make_online_run(n_nodes=20000, n_pairs=4096, n_kp=1000, seed=12345, desc_seed=777, alias_frac=0.05, desc_bytes=32,
                    max_pair_dist=0.9, max_pair_rot_deg=17.0, burst=16):
    """BASELINE config 5: `n_pairs` node-pair match jobs feeding a graph that grows to `n_nodes` nodes.
    The trajectory, odometry edges and dead-reckoned start poses are make_pose_graph's (seed 12345); the node pairs are places the
    robot revisits (closer than max_pair_dist, heading within max_pair_rot_deg: inside the acceptance gate, GraphSlam.cfg:19-20),
    ordered by their later node = the order the candidate producers would emit them online.  Every pair carries its own two
    FeatureData frames in the camera frame (make_pair's recipe, seed 777: 1000 landmarks, 8 % bit flips, 40 % clutter, 1 cm noise,
    10 % without depth) whose relative motion is the ground-truth relative pose of the two nodes; for a fraction `alias_frac` the
    frames come from a random motion instead (perceptual aliasing: a confident but wrong edge, what the gate, the filter and the
    Huber kernel exist for).  Camera = base (identity sensor transform).
    Returns dict(gt, init (N,3,4), fixed, odo (make_pose_graph edge arrays), stamps_ns (N) int64,
                 pair_from, pair_to, pair_later (P) int32, pair_alias (P) bool, frames: list of P (frame_from, frame_to))."""
    g = make_pose_graph(n_nodes, n_nodes - 1, seed=seed)
    N = n_nodes
    gt = g["gt_pose"].reshape(N, 3, 4)
    rng = np.random.default_rng(desc_seed)
    cand = _close_pairs(gt[:, :, 3], max_pair_dist, 20)
    if len(cand):
        rot = rotation_angle(np.swapaxes(gt[cand[:, 0], :, :3], 1, 2) @ gt[cand[:, 1], :, :3])
        cand = cand[rot < np.deg2rad(max_pair_rot_deg)]
    if len(cand) < n_pairs:
        raise ValueError("trajectory has only %d revisits for %d pairs" % (len(cand), n_pairs))
    # Revisits come in bursts: while the robot passes an old place, several consecutive new nodes each match several consecutive old
    # ones (the candidate producers emit every node within the radius, graph_slam_node.cpp:272-289) - and the edge filter only
    # validates clusters of >= 8 such edges (transformation_filter.cpp:233).  Draw seed revisits and take up to `burst` candidates
    # whose two nodes lie within +-6 nodes (3 s) of the seed's.
    key = cand[:, 0] * np.int64(N) + cand[:, 1]               # cand is sorted lexicographically, so key is ascending
    chosen = np.zeros(len(cand), bool)
    n_chosen = 0
    for seed_ix in rng.permutation(len(cand)):
        if n_chosen >= n_pairs:
            break
        if chosen[seed_ix]:
            continue
        i0, j0 = cand[seed_ix]
        near = []
        for i in range(max(0, i0 - 6), min(N, i0 + 7)):
            lo = np.searchsorted(key, i * np.int64(N) + max(0, j0 - 6)); hi = np.searchsorted(key, i * np.int64(N) + min(N - 1, j0 + 6), side="right")
            near.extend(range(lo, hi))
        near = np.array([k for k in near if not chosen[k]], np.int64)
        if len(near) > burst:
            near = rng.choice(near, size=burst, replace=False)
        near = near[:n_pairs - n_chosen]
        chosen[near] = True; n_chosen += len(near)
    if n_chosen < n_pairs:
        raise ValueError("could not draw %d pairs" % n_pairs)
    pr = cand[chosen]                                         # i < j, lexicographic
    order = np.lexsort((pr[:, 0], pr[:, 1]))                  # by later node j, then i
    pr = pr[order]
    flip = rng.random(n_pairs) < 0.5
    pf = np.where(flip, pr[:, 1], pr[:, 0]).astype(np.int32); pt = np.where(flip, pr[:, 0], pr[:, 1]).astype(np.int32)
    alias = rng.random(n_pairs) < alias_frac
    rel = se3_mul(se3_inv(gt[pf]), gt[pt])                    # from_T_to
    frames = []
    for k in range(n_pairs):
        f, t, _ = make_pair(rng, n_kp=n_kp, desc_bytes=desc_bytes, T=None if alias[k] else rel[k])
        frames.append((f, t))
    t0 = 1_400_000_000 * 10**9
    stamps = (t0 + (0.5e9 * np.arange(N)).astype(np.int64)).astype(np.int64)
    return dict(gt=gt, init=g["nodes_pose"].reshape(N, 3, 4), fixed=g["nodes_fixed"], odo=g["edges"], stamps_ns=stamps,
                pair_from=pf, pair_to=pt, pair_later=pr[:, 1].astype(np.int32), pair_alias=alias, frames=frames)


def permute_graph(g, perm):
    """The same pose graph with its nodes renumbered: new index of old node i = perm[i] (edges keep their order).  What a merged /
    global-scope graph looks like to the optimizer: std::map order is no longer trajectory order (graph_slam_node.cpp:401-576)."""
    perm = np.asarray(perm)
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    e = {k: np.array(v) for k, v in g["edges"].items()}
    e["from"] = perm[e["from"]].astype(np.int32); e["to"] = perm[e["to"]].astype(np.int32)
    out = dict(g)
    out["nodes_pose"] = np.asarray(g["nodes_pose"])[inv]; out["nodes_fixed"] = np.asarray(g["nodes_fixed"])[inv]
    out["gt_pose"] = np.asarray(g["gt_pose"])[inv]; out["edges"] = e
    return out


def interleave_sessions(ga, gb, n_cross=40, seed=3):
    """Two recorded sessions in one graph, node ids interleaved (a0, b0, a1, b1, ...: ids start with the stamp, two robots
    recording at the same time), joined by `n_cross` inter-session loop closures; session b's node 0 is not fixed."""
    rng = np.random.default_rng(seed)
    na, nb = len(ga["nodes_fixed"]), len(gb["nodes_fixed"])
    n = min(na, nb)
    ia = np.concatenate([2 * np.arange(n), 2 * n + np.arange(na - n)]); ib = np.concatenate([2 * np.arange(n) + 1, 2 * n + np.arange(nb - n)])
    if na > n:
        ib = ib
    N = na + nb
    pose = np.zeros((N, 12)); fixed = np.zeros(N, np.uint8); gt = np.zeros((N, 12))
    # session b lives 3 m to the side of session a (same world frame)
    off = np.eye(3, 4); off[1, 3] = 3.0
    def shift(P):
        return se3_mul(off, np.asarray(P).reshape(-1, 3, 4)).reshape(-1, 12)
    pose[ia] = ga["nodes_pose"]; pose[ib] = shift(gb["nodes_pose"]); gt[ia] = ga["gt_pose"]; gt[ib] = shift(gb["gt_pose"])
    fixed[ia] = ga["nodes_fixed"]
    ea, eb = ga["edges"], gb["edges"]
    e = {k: np.concatenate([np.asarray(ea[k]), np.asarray(eb[k])]) for k in ea}
    e["from"] = np.concatenate([ia[ea["from"]], ib[eb["from"]]]).astype(np.int32)
    e["to"] = np.concatenate([ia[ea["to"]], ib[eb["to"]]]).astype(np.int32)
    # inter-session closures between nodes that are close in the world
    G = gt.reshape(-1, 3, 4)
    from scipy.spatial import cKDTree
    d, j = cKDTree(G[ib][:, :, 3]).query(G[ia][:, :, 3])
    cand = np.argsort(d)[: max(n_cross * 4, n_cross)]
    sel = rng.choice(cand, size=min(n_cross, len(cand)), replace=False)
    f = ia[sel]; t = ib[j[sel]]
    Z = se3_mul(se3_mul(se3_inv(G[f]), G[t]), se3_from_noise(rng.normal(0, 0.05, (len(sel), 3)), rng.normal(0, 0.01, (len(sel), 3))))
    info = np.zeros((len(sel), 6, 6))
    for k in range(3):
        info[:, k, k] = 500.0; info[:, 3 + k, 3 + k] = 50000.0
    I12 = np.tile(np.eye(3, 4).reshape(1, 12), (len(sel), 1))
    extra = {"from": f.astype(np.int32), "to": t.astype(np.int32), "type": np.full(len(sel), EDGE_TYPE_3D_FULL, np.int32),
             "sensor_from": np.full(len(sel), -1, np.int32), "sensor_to": np.full(len(sel), -1, np.int32), "valid": np.ones(len(sel), np.int32),
             "transform": Z.reshape(-1, 12), "displacement_from": I12, "displacement_to": I12.copy(), "information": info.reshape(-1, 36),
             "diff_time": np.zeros(len(sel))}
    e = {k: np.concatenate([e[k], extra[k]]) for k in e}
    return dict(nodes_pose=pose, nodes_fixed=fixed, gt_pose=gt, edges=e)


def drop_odometry(g, keep_every=0):
    """The graph without its odometry chain (only feature edges; `keep_every` > 0 keeps every k-th odometry edge): what remains when
    odometry was never recorded as TYPE_2D_WHEEL_ODOMETRY edges.  Components without a fixed node get their gauge from setFixedNodes."""
    e = g["edges"]
    odo = np.asarray(e["type"]) == EDGE_TYPE_ODOM
    keep = ~odo
    if keep_every > 0:
        idx = np.nonzero(odo)[0]
        keep[idx[::keep_every]] = True
    out = dict(g)
    out["edges"] = {k: np.asarray(v)[keep] for k, v in e.items()}
    return out
