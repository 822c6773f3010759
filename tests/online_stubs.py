"""CPU-checker stand-ins for the four handles uzliti_slam_amd/online.py drives (capi.Match / Gate / Filter / Pgo), built on oracle/.
Test infrastructure: the GPU test replays a whole online run through them and compares; the CPU multi-rank test runs the driver's
sharding / gathering logic on them without a GPU."""
import numpy as np


class OMatch:
    def __init__(self, oracle, ransac_iteration=200, seed=777):
        self.O, self.fr, self.it, self.seed = oracle, [], ransac_iteration, seed

    def add_frame(self, d, p, v):
        self.fr.append(dict(desc=d, pos=p, valid=v, feature_type=2, sensor_frame=0))
        return len(self.fr) - 1

    def launch_raw(self, jobs, fids):
        self.jobs, self.fids = jobs.copy(), fids.copy()

    def collect(self, out):
        for i, j in enumerate(self.jobs):
            e = self.O.estimate_edge([self.fr[self.fids[j["from_begin"]]]], [self.fr[self.fids[j["to_begin"]]]], ransac_threshold=0.1,
                                     ransac_iteration=self.it, break_percentage=0.6, do_prosac=True, seed=self.seed, job_id=int(j["job_id"]))
            out[i]["job_id"] = j["job_id"]; out[i]["ok"] = e["ok"]; out[i]["consensus"] = e["consensus"]
            out[i]["T"] = np.asarray(e["T"]).reshape(12); out[i]["information"] = np.asarray(e["information"]).reshape(36); out[i]["mse"] = e["mse"]
        return out

    def close(self):
        pass


class OPgo:
    def __init__(self, oracle):
        self.O = oracle

    def add_graph(self, poses, fixed, edges):
        self.g = (np.array(poses), np.array(fixed), {k: np.array(v) for k, v in edges.items()})

    def optimize(self, its):
        fl = self.O.flatten_graph(*self.g)
        fx, _ = self.O.set_fixed_nodes(fl["fixed"], fl["ij"])
        self.P, so = self.O.pgo_optimize(fl["poses"], fx, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=its)
        so = dict(so); so.update(status=0, n_edges=len(fl["ij"]), pcg_iterations=0)
        return so

    def store(self):
        return self.P.reshape(-1, 12), None, None

    def close(self):
        pass


class OFilter:
    def __init__(self, oracle, stamps_ns, **cfg):
        self.f = oracle.Filter(**cfg); self.st = stamps_ns

    def set_sensors(self, x):
        self.f.set_sensors(x)

    def add_packed(self, fe):
        st = self.st; base = st.ctypes.data
        self.f.add([dict(key=int(r["key"]), matching_score=float(r["matching_score"]), valid=int(r["valid"]), sensor_from=-1, sensor_to=-1,
                         stamps_from=st[(int(r["stamps_from_ns"]) - base) // 8:][:1], stamps_to=st[(int(r["stamps_to_ns"]) - base) // 8:][:1],
                         transform=r["transform"], displacement_from=r["displacement_from"], displacement_to=r["displacement_to"],
                         pose_from=r["pose_from"], pose_to=r["pose_to"]) for r in fe])

    def calc_valid_edges(self):
        return self.f.calc_valid_edges()

    def valid_edges(self):
        return np.asarray(self.f.valid_edges())

    def close(self):
        pass


def oracle_online(oracle, run, ransac_iteration=200, seed=777, gate_cfg=None, **kw):
    """uzliti_slam_amd.online.OnlineSlam with its four handles replaced by the CPU checker's: the product class is driven unchanged,
    only the handle factory is overridden here, under tests/."""
    from uzliti_slam_amd import online

    class OracleOnlineSlam(online.OnlineSlam):
        def _open_handles(self, device, mc, gate_cfg_, filter_cfg, pgo_cfg):
            self.matcher = OMatch(oracle, ransac_iteration, seed)
            if self.is_solver:
                class OGate(oracle.Gate):           # the checker always runs the reference's search
                    def check(self, cand, want_dist=True):
                        return oracle.Gate.check(self, cand)
                self.gate = OGate(**(gate_cfg_ or {}))
                self.filt = OFilter(oracle, run["stamps_ns"], seed=seed)
                self.pgo = OPgo(oracle)

    return OracleOnlineSlam(run, gate_cfg=gate_cfg, **kw)
