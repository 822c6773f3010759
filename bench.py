#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X back end (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

Primary metric   : SE(3) edges optimised / s   on BASELINE config 2 (1k nodes / 5k edges, 20 LM iterations)
`secondary`      : node pairs matched / s      on BASELINE config 3 (512 pairs x 1000 ORB-256, 500 hypotheses)
`batched`        : 16 independent config-2 graphs per GPU in one batch call (uzl_pgo_batch_*: shared launches, two launch sequences from 12 graphs on)
`c4_1gpu`        : the north-star line: 10k nodes / 50k edges on ONE GPU against the CPU path (1 thread and all cores)   [N = 1]
`online_c5`      : BASELINE config 5: 4096 pair jobs feeding a graph that grows to 20k nodes, re-optimised every 256 edges
`formats`        : Feature records -> frame arena for the 1024 frames of config 3 (HBM-bound byte shuffle)           [rank 0]
`rooflines`      : every kernel SURVEY section 8(d) names, each against the roof that bounds it

A "step" is one pass of the hot path over one batch with the inputs already resident in HBM:
  primary   step = uzl_pgo_reset + uzl_pgo_optimize(20)   (graph resident, poses restored on the device)
  secondary step = uzl_match_estimate over the resident frames of 512 node pairs
The timed solves run with uzl_pgo_cfg::pass_history = 1: nothing an earlier optimize of the same graph learned sizes a pass of the
device-resident LM loop (the reference re-optimises a graph that has changed, graph_slam_node.cpp:1138-1150, not the identical problem);
the figure WITH that memory is reported beside it as `repeat_identical`, the first solve of a fresh structure as `first_solve_ms`.

Output: the full record goes to stderr and to gpurun_out/bench_full.json; the LAST line on stdout - the only one - is the compact record
(`compact_record`, < 6 KB, no prose) the driver parses.
With --gpus N > 1 (one rank per GPU under torch.distributed.run; when RANK is not set this script starts the ranks itself, before
anything touches the GPU) every rank solves its own independent graph / its own shard of node pairs: the path partitions into
independent units, so there is no data-path collective and scaling is weak; torch.distributed is used only for the barrier and
the max-over-ranks of the timed region.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GOPS = 256 * 4 * 32 * 2.4   # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz = 78 643 G lane-ops/s (32-bit VALU)
MFMA_I8_PEAK_TOPS = 5000.0            # MI355X_MICROARCH.md: int8 MFMA = 2 x the ~2.5 PFLOP/s dense bf16 rate
F64_PEAK_TFLOPS = 78.6                # MI355X_MICROARCH.md: f64 vector = f64 matrix = 78.6 TFLOP/s
TRAFFIC_JSON = os.path.join("profiles", "traffic.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=1000)
    ap.add_argument("--edges", type=int, default=5000)
    ap.add_argument("--lm-iters", type=int, default=20)
    ap.add_argument("--pairs", type=int, default=512)
    ap.add_argument("--keypoints", type=int, default=1000)
    ap.add_argument("--hypotheses", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-formats", action="store_true")
    ap.add_argument("--no-deployed", action="store_true", help="skip `secondary.deployed` (BRISK-512, 300 keypoints, 100 iterations, early exit)")
    ap.add_argument("--no-c4", action="store_true", help="skip the 10k/50k one-GPU block (N = 1 only)")
    ap.add_argument("--no-online", action="store_true", help="skip the BASELINE config 5 block")
    ap.add_argument("--no-batched", action="store_true", help="skip the batched multi-graph block")
    ap.add_argument("--batch", type=int, default=16, help="graphs per batch of the batched block")
    ap.add_argument("--batch-queue", type=int, default=256, help="graphs in the queue of the batched block's refill measurement (0 = skip)")
    ap.add_argument("--online-nodes", type=int, default=20000)
    ap.add_argument("--online-pairs", type=int, default=4096)
    ap.add_argument("--sharded", action="store_true", help="(default for N > 1; kept for old command lines)")
    ap.add_argument("--no-sharded", action="store_true",
                    help="skip `sharded_c4` (N > 1: ONE 10000-node/50000-edge graph sharded over all ranks, BASELINE config 4, native RCCL "
                         "all-reduce per PCG iteration) and `c4_1gpu.sharded_world1` (N = 1: the same path with a one-rank communicator)")
    ap.add_argument("--sharded-world1-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--sharded-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--online-cpu-seconds", type=float, default=25.0, help="budget of the CPU replay of config 5 (online_c5.cpu_baseline)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each CPU-baseline sample of the primary / secondary block")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the N > 1 code paths on a box with fewer GPUs than ranks: process group over gloo, rank r on device "
                         "r %% device_count (RCCL refuses two ranks on one device); the numbers mean nothing")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run as a child process (nothing in this
    process has touched the GPU yet) and relay: rank 0 of the child prints the JSON line to the inherited stdout."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class Dist:
    """barrier + max-reduce over ranks; a no-op at world size 1 (then torch is never imported)."""

    def __init__(self, n_gpus, rehearse_gloo=False):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.torch = None
        self.dist = None
        self.rehearsal = False
        if self.world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: one rank per GPU over RCCL
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if rehearse_gloo:
                self.local_rank = self.local_rank % max(torch.cuda.device_count(), 1)
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("gloo")
                self.rehearsal = True
            else:
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            self.torch = torch
            self.dist = dist
        if self.world != max(n_gpus, 1) and self.rank == 0:
            print(f"[bench] --gpus {n_gpus} but WORLD_SIZE={self.world}: using WORLD_SIZE", file=sys.stderr)

    def sync(self):
        if self.torch is not None:
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.torch is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def _red(self, v, op):
        if self.torch is None:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device="cpu" if self.rehearsal else "cuda")
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, v):
        return self._red(v, self.dist.ReduceOp.MAX) if self.torch is not None else v

    def sum(self, v):
        return self._red(v, self.dist.ReduceOp.SUM) if self.torch is not None else v

    def close(self):
        if self.torch is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed(dist, fn, steps):
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    dist.sync()
    dt = time.perf_counter() - t0
    dist.barrier()
    return dist.max(dt)


def _traffic_file():
    try:
        return json.load(open(os.path.join(ROOT, TRAFFIC_JSON)))
    except Exception:
        return {}


def traffic_of(key, applies):
    """PMC-measured HBM bytes per launch, collected by profiles/collect.sh on the default workloads (separate --pmc passes); this
    run does not measure it - `traffic_source` in the JSON says where it comes from."""
    return _traffic_file().get(key) if applies else None


def rocprof_us(key, applies):
    """the kernel's average dispatch duration in the committed rocprofv3 --kernel-trace --stats summary of the same collection
    (bench.py's own `avg_launch_us` is measured live in THIS run from the dispatch's begin / end timestamps - hipExtLaunchKernelGGL
    start / stop events on the launching stream -, which is what rocprofv3 reports; `timing` says so per kernel)"""
    return _traffic_file().get(key + "_rocprof_avg_us") if applies else None


def traffic_source():
    m = _traffic_file().get("_meta") or {}
    return "%s@%s (%s, %s)" % (TRAFFIC_JSON, m.get("commit", "unknown"), m.get("tag", "?"), m.get("collected_utc", "?"))


PHASES_JSON = os.path.join(ROOT, "profiles", "r03_estimate_phases.json")


def estimate_phases(applies):
    """Where a workgroup of estimate_kernel spends its time: in-kernel stamps of the diagnostic build (tests/diag/stamps_match.sh) on the
    default secondary workload, committed under profiles/; not measured in this run.  Each part with the bound that holds for it."""
    if not applies or not os.path.exists(PHASES_JSON):
        return None
    d = json.load(open(PHASES_JSON))
    us = d["phase_us_per_workgroup"]; tot = d["total_us_per_workgroup"]
    short = ["select", "sort", "gather", "poses", "votes", "bookkeeping", "winner_mask", "refit", "recount_mse"]
    bound = {"select": "LDS latency: ballot-ordered compaction, two workgroup barriers per 256 queries",
             "sort": "LDS latency: 55 compare-exchange stages at n = 1024, one LDS round trip each (52 inside a wave's quarter, 3 with barriers)",
             "gather": "HBM / L2 latency: two dependent gathers (key -> train index -> point)",
             "poses": "instruction latency of one lane: Eigen's two-sided Jacobi SVD in float, 3 - 6 sweeps of dependent divides and square roots; "
                      "256 hypotheses per round run it side by side, the chain length is what counts",
             "votes": "the CU's f64 units, shared by matrix and vector instructions (both peak at 78.6 TFLOP/s): per wave and 16-point step 12 "
                      "v_mfma_f64_16x16x4_f64 (64 cycles each) + 112 f64 vector instructions (4 cycles each) = ~1220 cycles, two waves per SIMD; the "
                      "measured ~2870 cycles per step of both waves is 85 % of that.  The priced 27 flop per (hypothesis, point) over this phase alone "
                      "is `vote_phase_frac_of_f64_peak` (the MFMA's fourth k-slot carries the translation, subtractions and compares count one flop)",
             "bookkeeping": "two ballots and three barriers per round",
             "winner_mask": "one lane recomputes the winning pose (same Jacobi chain), then one pass over the points",
             "refit": "sequential by definition (running mean / covariance in inlier order, float): ~490 dependent three-operation steps on one wave, "
                      "then the Jacobi chain once more on one lane",
             "recount_mse": "one lane adds the distances in index order (the reference's summation order)"}
    names = list(us.keys())
    return dict(source="profiles/r03_estimate_phases.json (" + d["source"] + ")", us_per_workgroup={k: round(us[n], 2) for k, n in zip(short, names)},
                fraction={k: round(us[n] / tot, 4) for k, n in zip(short, names)}, bound=bound)


def roof(kernel, bound, achieved, peak, unit, **extra):
    d = dict(kernel=kernel, bound=bound, achieved=round(achieved, 3), peak=peak, unit=unit, frac=round(achieved / peak, 5) if peak else None)
    d.update(extra)
    return d


# ---------------------------------------------------------------------------------------------------------------------- pose graph
def pgo_block(capi, synth, dist, dev, a, nodes, edges, steps, warmup, seed, xy=False, repeat=True):
    """resident-graph solve loop (pass_history = 1: no pass sized from an earlier optimize of the same graph), then the same loop with
    the memory on (`repeat_identical`)"""
    g = synth.make_pose_graph(nodes, edges, seed=seed)
    pgo = capi.Pgo(device=dev, iterations=a.lm_iters, optimize_xy_only=1 if xy else 0, pass_history=1)
    t0 = time.perf_counter()
    pgo.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])            # H2D + flattening kernels, outside the timed region
    h2d_ms = 1e3 * (time.perf_counter() - t0)
    work = {"edges": 0}
    # the first optimize after add_graph: gauge, block-CSR, hierarchy, no captured segment, no history - what a fresh local-scope graph costs
    t0 = time.perf_counter()
    st_first = pgo.optimize(a.lm_iters)
    first_ms = 1e3 * (time.perf_counter() - t0)

    def step():
        pgo.reset()
        st = pgo.optimize(a.lm_iters)
        work["edges"] += st["n_edges"] * st["iterations_done"]
        work["last"] = st

    for _ in range(warmup):
        step()
    work["edges"] = 0
    t = timed(dist, step, steps)
    t0 = time.perf_counter()
    poses = pgo.store()[0]
    d2h_ms = 1e3 * (time.perf_counter() - t0)
    out = dict(g=g, pgo=pgo, t=t, edges=work["edges"], st=dict(work["last"]), h2d_ms=h2d_ms, d2h_ms=d2h_ms, poses=poses,
               first=dict(first_solve_ms=round(first_ms, 3), structure_ms=round(st_first["structure_ms"], 3), lm_passes=st_first["lm_passes"]))
    if repeat:
        pgo.set_config(pass_history=0)
        for _ in range(2):                                                    # one solve to learn the counts, one to use them
            step()
        e0 = work["edges"]
        n_rep = max(2, steps // 2)
        t_rep = timed(dist, step, n_rep)
        out["repeat"] = dict(value=round(dist.sum(float(work["edges"] - e0)) / t_rep, 1), ms_per_step=round(1e3 * t_rep / n_rep, 4), lm_passes=work["last"]["lm_passes"])
        pgo.set_config(pass_history=1)
        # the reference's timer tick on a graph nothing was added to (graph_slam_node.cpp:1138-1150 after :1248-1282 wrote the poses back):
        # optimize again FROM the solved poses - twenty more LM iterations, each a few PCG iterations long
        pgo.reset(); pgo.optimize(a.lm_iters)
        calls = []
        for _ in range(3):                                                    # (the third usually finds nothing left: rho = 0, Terminate after one trial)
            dist.barrier(); t0 = time.perf_counter()
            stc_ = pgo.optimize(a.lm_iters)
            dist.sync(); t_c = dist.max(time.perf_counter() - t0)
            calls.append(dict(ms=round(1e3 * t_c, 4), lm_iterations_done=stc_["iterations_done"], pcg_iterations=stc_["pcg_iterations"], lm_passes=stc_["lm_passes"]))
        out["continued"] = dict(calls=calls, ms_first_call=calls[0]["ms"], chi2_final=stc_["chi2_final"])
        pgo.reset()
    return out


def pgo_profile(pgo, a):
    pgo.set_profiling(True)
    pgo.reset(); st = pgo.optimize(a.lm_iters)
    kt = pgo.kernel_times()
    pgo.set_profiling(False)
    return st, kt


def pgo_rooflines(pgo, st, st_prof, kt, nodes, edges, is_default):
    nb = st["n_vertices"] - int(pgo.get_fixed().sum())
    E = st["n_edges"]
    out = []
    spmv = kt.get("pcg_spmv", dict(ms=0.0, launches=1))
    alg = 288.0 * (nb + E) + 96.0 * nb        # H once (symmetric) + read p + write Ap  
    # launches after the device-side `done` flag are ~0.7 us no-ops that move nothing: bytes are counted for the launches that did
    # work (= PCG iterations of the profiled solve) over the kernel's whole measured time
    active = min(int(st_prof["pcg_iterations"]), int(spmv["launches"])) or 1
    ach = alg * active / (spmv["ms"] * 1e-3) / 1e9 if spmv["ms"] > 0 else 0.0
    agg4 = nb > 2048
    out.append(roof("ml_spmv_kernel<%d>" % (4 if agg4 else 1) if pgo.cfg.preconditioner else "pcg_spmv_kernel", "hbm", ach, HBM_PEAK_GBS, "GB/s",
                    traffic=traffic_of("pcg_spmv_bytes_per_launch" if not agg4 else "pcg_spmv4_bytes_per_launch", is_default),
                    algorithmic_bytes_per_launch=alg, avg_launch_us=round(1e3 * spmv["ms"] / max(spmv["launches"], 1), 3),
                    rocprof_avg_us=rocprof_us("pcg_spmv4" if agg4 else "pcg_spmv", is_default), timing="dispatch_timestamps",
                    launches=spmv["launches"], active_launches=active,
                    note="working set (H = %.1f MB) is L2 / Infinity-Cache resident; launch-latency bound at this size" % (288e-6 * (nb + 2 * E))))
    lin = kt.get("linearize")
    if lin and lin["ms"] > 0:
        alg_l = 632.0 * E + 336.0 * st["n_vertices"]                      # B_lin = 632 E + 336 N
        out.append(roof("hessian_kernel", "hbm", alg_l * lin["launches"] / (lin["ms"] * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s",
                        traffic=traffic_of("c4_hessian_bytes_per_launch" if agg4 else "hessian_bytes_per_launch", is_default), algorithmic_bytes_per_launch=alg_l,
                        avg_launch_us=round(1e3 * lin["ms"] / lin["launches"], 3), launches=lin["launches"],
                        rocprof_avg_us=rocprof_us("c4_hessian" if agg4 else "hessian", is_default), timing="dispatch_timestamps",
                        note="the sparse Hessian build as ONE row-gather kernel (rounds 1-3: linearize_kernel + assemble_kernel): a lane per slot reads its slot-major "
                             "record, recomputes its edge's Jacobians and leaves the H_ac block and its share of H_aa | b in LDS; the workgroup writes the blocks out "
                             "contiguously, lane (row, r) adds the shares in slot order - no atomics, nothing but H itself in HBM"))
    gm = kt.get("ml_ns_gemm")
    if gm and gm["ms"] > 0:
        n1 = (nb + 7) // 8
        n_c = n1 if not agg4 else (n1 + 3) // 4
        # the product is symmetric: the kernel computes the 64 x 64 tiles on and above the diagonal and mirrors them; flops = what it executes
        n6 = int(6 * n_c); gt = (n6 + 63) // 64
        ext = [min(64, n6 - 64 * i) for i in range(gt)]
        upper = sum(ext[i] * ext[j] for i in range(gt) for j in range(i, gt))
        flop = 2.0 * n6 * upper
        out.append(roof("ml_ns_gemm_kernel" if agg4 else "ml_ns_gemm32_kernel", "mfma", flop * gm["launches"] / (gm["ms"] * 1e-3) / 1e12, F64_PEAK_TFLOPS, "TFLOP/s (f64 matrix cores)",
                        traffic=traffic_of("c4_ns_gemm_bytes_per_launch" if agg4 else "ns_gemm32_bytes_per_launch", is_default),
                        operand_bytes_per_launch=3.0 * 8.0 * n6 * n6, flop_per_launch=flop, n=n6, avg_launch_us=round(1e3 * gm["ms"] / gm["launches"], 3), launches=gm["launches"],
                        rocprof_avg_us=rocprof_us("c4_ns_gemm" if agg4 else "ns_gemm32", is_default), timing="dispatch_timestamps",
                        note="the block-GEMM of the path: X' = 2X - X(AX), v_mfma_f64_16x16x4_f64, 64 x 64 tiles on and above the diagonal (the product is symmetric: "
                             "%.0f %% of 2 n^3); %d launches per solve" % (100.0 * flop / (2.0 * n6 ** 3), gm["launches"])))
    return out


def cpu_pgo(O, g, a, seconds, threads_list, max_solves=64):
    """the CPU path on this host: C restatement of g2o LM + block sparse direct Cholesky (`-O3 -march=native -fopenmp`, built here)"""
    fl = O.flatten_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
    fixed, _ = O.set_fixed_nodes(fl["fixed"], fl["ij"])
    out = {}
    for th in threads_list:
        t0 = time.perf_counter(); n_solves = 0; cpu_edges = 0
        while True:
            P, so = O.pgo_optimize(fl["poses"], fixed, fl["ij"], fl["meas"], fl["info"], fl["robust"], iterations=a.lm_iters, native_threads=th)
            if "poses" not in out:
                out["poses"], out["stats"] = P, dict(so)                    # kept: the parity block compares the GPU solve of the same graph with it
            n_solves += 1; cpu_edges += so["n_edges"] * so["iterations_done"]
            if time.perf_counter() - t0 > seconds or n_solves >= max_solves:
                break
        dt = time.perf_counter() - t0
        out[th] = dict(value=round(cpu_edges / dt, 1), seconds_per_solve=round(dt / n_solves, 4), solves=n_solves, seconds=round(dt, 2),
                       cholesky_share=round(so["t_numeric_ms"] / max(so["t_total_ms"], 1e-9), 3))
    return out


PARITY_T, PARITY_R = 1e-3, 1e-4          # BASELINE.json north_star: within 1e-3 m / 1e-4 rad of the CPU path after the same iteration count


def parity_block(synth, poses_gpu, st, cb):
    """GPU solve against the CPU checker's direct solve of the same graph, same LM iteration count (G2oOptimizer::optimizeImpl,
    g2o_optimizer.cpp:137-149).  The CPU solve is the one cpu_baseline times; a miss makes the run exit non-zero."""
    so = cb["stats"]
    dt, dr = synth.pose_errors(np.asarray(poses_gpu).reshape(-1, 3, 4), np.asarray(cb["poses"]).reshape(-1, 3, 4))
    chi2_rel = abs(st["chi2_final"] - so["chi2_final"]) / max(abs(so["chi2_final"]), 1e-300)
    return dict(dt_m=float(dt), dr_rad=float(dr), chi2_rel=float(chi2_rel), lm_iterations_equal=bool(st["iterations_done"] == so["iterations_done"]),
                lm_trials_equal=bool(st["lm_trials"] == so["lm_trials"]), lm_trials=[int(st["lm_trials"]), int(so["lm_trials"])],
                bar=dict(dt_m=PARITY_T, dr_rad=PARITY_R), ok=bool(dt < PARITY_T and dr < PARITY_R),
                against="oracle/ (C restatement of g2o LM + sparse direct Cholesky), the solve cpu_baseline times; parity unpinned (DESIGN.md section 2)")


def effective_cpus():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a one-GPU job a share of
    its host, e.g. 16 of 256 hardware threads; threads beyond the quota only time-slice)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


# ---------------------------------------------------------------------------------------------------------------------- deployed estimator point
def deployed_block(capi, synth, dist, dev, a):
    """The operating point the reference deploys: BRISK-512 descriptors (64 bytes), 300 keypoints per frame
    (feature_extraction_service_node.cpp:63-66), ransac_threshold 0.1 and 100 PROSAC iterations (iti_slam_launch/yaml/slam.yaml:34-38),
    early exit at 60 % consensus (cfg/FeatureLinkEstimation.cfg:12).  Same 512 node pairs per GPU as config 3; every pair is compared
    with the CPU checker on rank 0."""
    n_pairs, n_kp, hyp, brk = a.pairs, 300, 100, 0.6
    pairs = synth.make_pairs(n_pairs, n_kp=n_kp, desc_bytes=64, seed=4242 + dist.rank)
    m = capi.Match(device=dev, ransac_threshold=0.1, ransac_iteration=hyp, ransac_break_percentage=brk, do_prosac=1, seed=777)
    ids = [(m.add_frame(f["desc"], f["pos"], f["valid"]), m.add_frame(t["desc"], t["pos"], t["valid"])) for f, t, _ in pairs]
    jobs, fids = capi.Match._jobs(ids, None)
    res = np.zeros(n_pairs, capi.EDGE_RESULT_DTYPE)

    def step():
        m.launch_raw(jobs, fids)
        m.collect(res)
    for _ in range(a.warmup):
        step()
    t = timed(dist, step, a.steps)
    m.set_profiling(True); step(); mk = m.kernel_times(); m.set_profiling(False)
    out = dict(workload="%d node pairs x %d BRISK-512 descriptors per frame, 2-NN Hamming + ratio test + <= %d PROSAC iterations, early exit at %.0f %% consensus "
                        "(slam.yaml:34-38, feature_extraction_service_node.cpp:63-66)" % (n_pairs, n_kp, hyp, 100 * brk),
               value=round(dist.sum(float(n_pairs * a.steps)) / t, 1), unit="pairs/s", ms_per_step=round(1e3 * t / a.steps, 4),
               kernels_ms={k: round(v["ms"], 4) for k, v in mk.items()}, mean_correspondences=float(res["n_corr"].mean()),
               mean_iterations_run=float(res["iterations_run"].mean()), mean_consensus=float(res["consensus"].mean()), ok_fraction=float(res["ok"].mean()))
    knn_ms = mk.get("knn2", dict(ms=0.0))["ms"]; est_ms = mk.get("estimate", dict(ms=0.0))["ms"]
    if knn_ms > 0:
        out["knn2_frac_of_int8_peak"] = round(2.0 * n_pairs * n_kp * n_kp * 512.0 / (knn_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS, 4)
    if est_ms > 0:
        # the votes actually cast: iterations_run x M x 27 flop per pair (the early exit stops most jobs long before iteration 100)
        flop = 27.0 * float((res["iterations_run"].astype(np.float64) * res["n_corr"]).sum())
        out["estimate_vote_frac_of_f64_peak"] = round(flop / (est_ms * 1e-3) / 1e12 / F64_PEAK_TFLOPS, 5)
        out["estimate_bound"] = ("a job casts ~%.0f x %.0f votes (early exit): the vote phase is a few microseconds; what remains is one workgroup per job walking "
                                 "select / sort / gather / one-lane float Jacobi SVD per hypothesis round / sequential refit - latency chains, not f64 throughput "
                                 "(phase split at config 3: rooflines[estimate_kernel].phases)" % (res["iterations_run"].mean(), res["n_corr"].mean()))
    if dist.rank == 0 and dist.world == 1 and not a.no_cpu_baseline:
        import oracle as O
        kw = dict(ransac_threshold=0.1, ransac_iteration=hyp, break_percentage=brk, do_prosac=True, seed=777)
        bad = 0
        t0 = time.perf_counter()
        for j, (f, tt, _) in enumerate(pairs):                                  # all pairs, field by field (the oracle's portable build)
            w = O.estimate_edge([f], [tt], job_id=j, **kw)
            same = (res[j]["consensus"] == w["consensus"] and res[j]["n_corr"] == w["n_corr"] and res[j]["best_iteration"] == w["best_iteration"]
                    and res[j]["iterations_run"] == w["iterations_run"] and np.array_equal(res[j]["T"].reshape(3, 4), w["T"]) and res[j]["mse"] == w["mse"])
            bad += 0 if same else 1
        out["parity"] = dict(pairs_checked=n_pairs, pairs_differing=bad, fields="consensus, n_corr, best_iteration, iterations_run, T, mse (array_equal)",
                             seconds=round(time.perf_counter() - t0, 2))
        out["parity_ok"] = bad == 0
        fp = [(f, tt) for f, tt, _ in pairs]
        chunks = [O.PreparedPairs(fp[k:k + 32]) for k in range(0, len(fp), 32)]
        n1 = 0; t0 = time.perf_counter()
        while time.perf_counter() - t0 < min(a.cpu_seconds, 5.0) and n1 < 65536:
            ch = chunks[(n1 // 32) % len(chunks)]
            O.estimate_edge_batch(ch, job_id0=n1, threads=1, **kw); n1 += ch.n
        dtm = time.perf_counter() - t0
        out["cpu_baseline"] = dict(value=round(n1 / dtm, 2), unit="pairs/s", cores=1, kind="port",
                                   sample="%d node pairs drawn cyclically from the same %d, %.1f s; gcc -O3 -march=native on this host" % (n1, n_pairs, dtm))
    m.close()
    return out


# ---------------------------------------------------------------------------------------------------------------------- formats
def bench_formats(capi, dev, pairs, n_kp):
    """FeatureData::fromMsg for every frame of the secondary workload in ONE launch: serialised graph_slam_msgs/Feature records
    (41 + 4 D bytes per keypoint, one float32 per descriptor byte) -> descriptor rows / positions / flags in the frame arena.
    Pure byte shuffle: HBM bound; algorithmic bytes = records in + arrays out.  Records are built on the host with numpy
    (layout only, untimed); the timed quantity is the kernel (HIP events on the estimator's stream)."""
    from uzliti_slam_amd import wire as W
    frames = [f for p in pairs for f in p[:2]]
    D = frames[0]["desc"].shape[1]
    dt = np.dtype([("u", "<i4"), ("v", "<i4"), ("is_3d", "u1"), ("keypoint_strength", "<f4"), ("count", "<u4"),
                   ("descriptor", "<f4", (D,)), ("keypoint_position", "<f8", (3,))], align=False)
    keep = W._Keep()
    sens = (W.WireSensor * len(frames))()
    n_kp_total = 0
    for k, f in enumerate(frames):
        n = len(f["desc"])
        rec = np.zeros(n, dt)
        rec["is_3d"] = f["valid"]; rec["keypoint_strength"] = -1.0; rec["count"] = D
        rec["descriptor"] = f["desc"]; rec["keypoint_position"] = np.asarray(f["pos"]).T
        sens[k].sensor_type = 1; sens[k].descriptor_type = 2; sens[k].n_features = n; sens[k].desc_len = D; sens[k].uniform = 1
        sens[k].records = keep.span(rec.tobytes())
        n_kp_total += n
    m = capi.Match(device=dev)
    m.set_profiling(True)
    best = None
    for _ in range(3):
        ids, _ = W.add_frames_wire(m, sens, len(frames))
        ms = m.kernel_times().get("wire_unpack", dict(ms=0.0))["ms"]
        best = ms if best is None or (0 < ms < best) else best
        for i in ids:
            m.remove_frame(i)
    ids, _ = W.add_frames_wire(m, sens, len(frames))                        # spot check against the arrays the records were made from
    gd, gp, gv = W.get_frame(m, ids[-1])
    ok = bool(np.array_equal(gd, frames[-1]["desc"]) and np.array_equal(gp, np.asarray(frames[-1]["pos"], np.float64)) and np.array_equal(gv, np.asarray(frames[-1]["valid"], np.uint8)))
    m.close()
    alg = float(n_kp_total) * ((41 + 4 * D) + (D + 25))
    ach = alg / (best * 1e-3) / 1e9 if best and best > 0 else 0.0
    return dict(kernel="wire_unpack_kernel", workload="%d frames x %d keypoints, ORB-256: Feature records -> frame arena, one launch" % (len(frames), n_kp),
                ms=round(best or 0.0, 4),
                roofline=roof("wire_unpack_kernel", "hbm", ach, HBM_PEAK_GBS, "GB/s", algorithmic_bytes=alg,
                              traffic=traffic_of("wire_unpack_bytes_per_launch", len(frames) == 1024 and n_kp == 1000), traffic_source=traffic_source()),
                matches_source_arrays=ok)


# ---------------------------------------------------------------------------------------------------------------------- config 4, sharded
def sharded_block(capi, synth, dist, dev, a, rank, world, budget_s=20.0):
    """BASELINE config 4: ONE 10k-node / 50k-edge graph, edges partitioned over the ranks, vertices replicated, native RCCL all-reduce of
    [A p | restricted A p | p.Ap partials] per PCG iteration on the solver's stream.  Strong scaling; never `value`."""
    g4 = synth.make_pose_graph(10000, 50000, seed=12345)
    p4 = capi.Pgo(device=dev)
    box = [capi.rccl_unique_id() if rank == 0 else None]
    if world > 1:
        dist.dist.broadcast_object_list(box, src=0)
    p4.set_shard_rccl(rank, world, box[0])
    p4.add_graph(g4["nodes_pose"], g4["nodes_fixed"], g4["edges"])

    def c4_step():
        p4.reset()
        return p4.optimize(a.lm_iters)
    t0 = time.perf_counter()
    st4 = c4_step()                                                            # warm-up: structure, communicator, first launches
    first_s = time.perf_counter() - t0
    n4 = int(max(1, min(max(1, a.steps // 5), budget_s / max(first_s, 1e-3))))  # capped at ~budget_s seconds of solves
    if world > 1:                                                              # every rank must time the same number of solves
        n4 = int(dist._red(float(n4), dist.dist.ReduceOp.MIN))
    acc = dict(ex_ms=0.0, ex_calls=0)

    def timed_step():
        st_ = c4_step()
        acc["ex_ms"] += st_["exchange_ms"]; acc["ex_calls"] += st_["exchange_calls"]
    t4 = timed(dist, timed_step, n4)
    out = dict(metric="SE(3) edges optimized/sec, one 10k-node / 50k-edge graph sharded over all ranks (BASELINE config 4)", unit="edges/s", scaling="strong",
               value=round(st4["n_edges"] * st4["iterations_done"] * n4 / t4, 1), n_ranks=world, rccl_ranks_seen=p4.rccl_ranks(), n_eliminated=st4["n_eliminated"],
               ms_per_solve=round(1e3 * t4 / n4, 3), solves_timed=n4, pcg_iterations_per_solve=st4["pcg_iterations"], lm_trials_per_solve=st4["lm_trials"],
               exchange="native RCCL (communicator owned by the handle): 1 all-reduce per PCG iteration + 3 per LM trial, on the solver's stream",
               exchange_calls_per_solve=acc["ex_calls"] // n4,
               exchange_ms_per_solve=round(acc["ex_ms"] / n4, 3), exchange_ms_note="host time inside the ncclAllReduce calls of rank 0 (enqueue cost; the collective itself runs on the stream)",
               chi2_final=st4["chi2_final"])
    p4.close()
    return out


def sharded_world1_child(a):
    """`c4_1gpu.sharded_world1`, run in a process of its own so that a stalled RCCL bootstrap is a missing block, not a hung bench: the
    10k/50k graph through uzl_pgo_set_shard_rccl(0, 1): every exchange step of the sharded path with a one-rank communicator - what the
    eager launches and the RCCL call per PCG iteration cost on top of the hipGraph-captured solve."""
    from uzliti_slam_amd import capi, synth

    class One:                      # the Dist interface at world 1
        rank = 0; world = 1; dist = None
        def barrier(self): pass
        def sync(self): pass
        def max(self, v): return v
    print(json.dumps(sharded_block(capi, synth, One(), 0, a, 0, 1)))


# ---------------------------------------------------------------------------------------------------------------------- output
COMPACT_LIMIT = 6000


def _finite(o):
    """strict JSON: NaN / Infinity become null"""
    if isinstance(o, float):
        return o if o == o and o not in (float("inf"), float("-inf")) else None
    if isinstance(o, dict):
        return {k: _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, (np.floating,)):
        return _finite(float(o))
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.bool_,)):
        return bool(o)
    return o


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None} if isinstance(d, dict) else None


ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch", "flop_per_launch", "avg_launch_us", "timing",
             "rocprof_avg_us", "launches", "active_launches")
CPU_KEYS = ("value", "unit", "cores", "kind", "nproc", "cpu", "sample", "seconds_per_solve")


def _roof(r):
    r = _pick(r, ROOF_KEYS)
    if r is not None and "traffic" not in r:
        r["traffic"] = None
    if r and isinstance(r.get("unit"), str):
        r["unit"] = r["unit"].split(" ")[0]
    return r


def _cpu(c):
    c = _pick(c, CPU_KEYS)
    if c and isinstance(c.get("sample"), str) and len(c["sample"]) > 110:
        c["sample"] = c["sample"][:107] + "..."
    return c


def compact_record(out):
    """The record the driver parses: the contract's keys, the headline's `roofline` and `cpu_baseline`, the parity verdict and one short
    summary per block - numbers only, no prose, < COMPACT_LIMIT bytes (the full record travels on stderr and in gpurun_out/)."""
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    c["config"] = _pick(cfg, ("workload", "system_edges", "lm_iterations_done", "lm_trials_per_solve", "pcg_iterations_per_solve", "pcg_tol"))
    c["timing"] = "pass_history=1"
    c["first_solve_ms"] = out.get("first_solve_ms")
    c["repeat_identical"] = _pick(out.get("repeat_identical"), ("value", "ms_per_step"))
    cf = out.get("continued_from_solution") or {}
    c["continued_from_solution"] = [[x.get("ms"), x.get("lm_iterations_done"), x.get("pcg_iterations")] for x in cf.get("calls", [])] or None
    c["roofline"] = _roof(out.get("roofline"))
    c["rooflines"] = [_pick(r, ("kernel", "bound", "frac", "avg_launch_us", "rocprof_avg_us", "traffic")) for r in (out.get("rooflines") or [])][:6]
    c["cpu_baseline"] = _cpu(out.get("cpu_baseline"))
    c["parity"] = _pick(out.get("parity"), ("ok", "dt_m", "dr_rad", "lm_trials_equal"))
    c["lm_overhead_ms"] = out.get("lm_overhead_ms")
    c["streams"] = _pick(out.get("streams"), ("pooled", "leased", "pairs_measured", "fallbacks", "probe_ms"))
    x = out.get("xy_only")
    if x:
        c["xy_only"] = _pick(x, ("value", "ms_per_solve", "pcg_iterations_per_solve"))
    sec = out.get("secondary")
    if sec:
        d = _pick(sec, ("metric", "value", "unit", "ms_per_step"))
        d["roofline"] = _pick(sec.get("roofline"), ("kernel", "bound", "achieved", "peak", "frac", "traffic", "ms"))
        d["cpu_baseline"] = _pick(sec.get("cpu_baseline"), ("value", "unit", "cores", "kind"))
        d["upload_inclusive"] = (sec.get("upload_inclusive") or {}).get("value")
        d["kernels_ms"] = sec.get("kernels_ms")
        dep = sec.get("deployed")
        if dep:
            d["deployed"] = _pick(dep, ("value", "ms_per_step", "kernels_ms", "parity_ok"))
            d["deployed"]["cpu_baseline"] = (dep.get("cpu_baseline") or {}).get("value")
        c["secondary"] = d
    b = out.get("batched")
    if b:
        d = _pick(b, ("value", "unit", "graphs", "graphs_batched", "ms_per_batch", "vs_single_graph", "batch_create_ms"))
        d["repeat_identical"] = (b.get("repeat_identical") or {}).get("value")
        d["rooflines"] = [_pick(r, ("kernel", "frac", "avg_launch_us")) for r in (b.get("rooflines") or [])]
        for k in ("small_graphs", "chain_like"):
            if b.get(k):
                d[k] = _pick(b[k], ("value", "ms_per_batch", "one_graph_alone", "vs_one_graph_alone"))
                if b[k].get("repeat_identical"):
                    d[k]["repeat_identical"] = b[k]["repeat_identical"].get("value")
        q = b.get("queue") or {}
        d["queue"] = {k: v.get("value") for k, v in q.items() if isinstance(v, dict) and "value" in v}
        c["batched"] = d
    f = out.get("formats")
    if f:
        c["formats"] = dict(ms=f.get("ms"), frac=(f.get("roofline") or {}).get("frac"), traffic=(f.get("roofline") or {}).get("traffic"))
    c4 = out.get("c4_1gpu")
    if c4:
        d = _pick(c4, ("value", "unit", "ms_per_solve", "pcg_iterations_per_solve", "speedup_vs_cpu_1_thread", "speedup_vs_cpu_all_cores"))
        d["first_solve_ms"] = (c4.get("first_solve") or {}).get("first_solve_ms")
        d["repeat_identical"] = (c4.get("repeat_identical") or {}).get("value")
        d["roofline"] = _roof(c4.get("roofline"))
        d["rooflines"] = [_pick(r, ("kernel", "bound", "frac", "avg_launch_us", "rocprof_avg_us", "traffic")) for r in (c4.get("rooflines") or [])]
        d["cpu_baseline"] = _pick(c4.get("cpu_baseline"), ("value", "unit", "cores", "kind", "seconds_per_solve"))
        d["parity"] = _pick(c4.get("parity"), ("ok", "dt_m", "dr_rad", "lm_trials_equal"))
        w1 = c4.get("sharded_world1") or {}
        d["sharded_world1"] = _pick(w1, ("ms_per_solve", "vs_graph_captured_solve", "rccl_ranks_seen", "exchange_calls_per_solve"))
        c["c4_1gpu"] = d
    o5 = out.get("online_c5")
    if o5:
        d = _pick(o5, ("wall_s", "pairs_per_s", "edges_per_s", "solves", "optimize_ms_per_solve", "structure_ms_per_solve", "add_graph_ms_per_solve",
                       "pcg_iterations", "lm_iterations", "not_converged", "ate_online_m", "seconds"))
        d["roofline"] = _pick(o5.get("roofline"), ("kernel", "bound", "achieved", "peak", "frac", "avg_launch_us"))
        cb = o5.get("cpu_baseline") or {}
        d["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "gpu_same_prefix_s", "speedup_on_prefix"))
        d["parity"] = _pick(cb.get("pose_difference_at_that_point"), ("ok", "dt_m", "dr_rad"))
        c["online_c5"] = d
    s4 = out.get("sharded_c4")
    if s4:
        c["sharded_c4"] = _pick(s4, ("value", "unit", "scaling", "n_ranks", "rccl_ranks_seen", "ms_per_solve", "exchange_calls_per_solve", "pcg_iterations_per_solve"))
        if "error" in s4:
            c["sharded_c4"]["error"] = str(s4["error"])[-200:]
    c = _finite(c)
    line = json.dumps(c, separators=(",", ":"), allow_nan=False)
    for drop in ("streams", "formats", "xy_only", "rooflines"):            # (never needed so far: the record is ~4 KB)
        if len(line) < COMPACT_LIMIT:
            break
        c.pop(drop, None)
        line = json.dumps(c, separators=(",", ":"), allow_nan=False)
    return line


def emit(out):
    full = json.dumps(_finite(out), allow_nan=False)
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_full.json"), "w") as f:
            f.write(full + "\n")
    except OSError:
        pass
    sys.stderr.write(full + "\n")
    sys.stderr.flush()
    print(compact_record(out), flush=True)


def sharded_child(a):
    """`sharded_c4` in processes of its own (rank 0 of the bench starts `torch.distributed.run ... bench.py --sharded-child` and waits with a
    time limit): the handle-owned RCCL communicator over N ranks has never run on hardware - no multi-GPU box in five rounds - and a stalled
    bootstrap must cost this one block, not the whole record.  Rendezvous and the barrier / max of the timed region over gloo; the
    data path is the library's own ncclAllReduce."""
    d = Dist(a.gpus, rehearse_gloo=True)
    from uzliti_slam_amd import capi, synth
    capi.lib()
    out = sharded_block(capi, synth, d, d.local_rank, a, d.rank, d.world)
    if d.rank == 0:
        print(json.dumps(_finite(out), allow_nan=False), flush=True)
    d.close()


def run_sharded_child(a, world, timeout_s=240):
    """rank 0 of the bench: the N-rank sharded block as a child job; returns its record or {"error": ...}"""
    import signal
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), "--sharded-child", "--gpus", str(world), "--steps", str(a.steps), "--lm-iters", str(a.lm_iters)]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("NCCL_SOCKET_IFNAME", "lo")
    try:
        pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)      # its own process group: killed as one
        try:
            so, se = pr.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(pr.pid, signal.SIGKILL)
            pr.communicate()
            return dict(error="the %d-rank sharded solve did not finish within %d s (killed)" % (world, timeout_s))
        lines = [ln for ln in so.strip().splitlines() if ln.startswith("{")]
        if pr.returncode != 0 or not lines:
            return dict(error=("rc %d: " % pr.returncode) + (se or so)[-400:])
        return json.loads(lines[-1])
    except Exception as ex:
        return dict(error=repr(ex)[:400])


def main():
    a = parse()
    if a.sharded_world1_child:
        return sharded_world1_child(a)
    if a.sharded_child:
        return sharded_child(a)
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a))
    dist = Dist(a.gpus, a.rehearse_gloo)
    from uzliti_slam_amd import capi, synth
    from uzliti_slam_amd import dist as ud
    capi.lib()
    dev = dist.local_rank
    is_c2 = (a.nodes, a.edges) == (1000, 5000)
    parity_fail = []

    # ------------------------------------------------------------------ primary: pose-graph solve, one independent graph per rank
    B = pgo_block(capi, synth, dist, dev, a, a.nodes, a.edges, a.steps, a.warmup, ud.replica_seed(12345, dist.rank))
    pgo, g, st, t_pgo = B["pgo"], B["g"], B["st"], B["t"]
    value = dist.sum(float(B["edges"])) / t_pgo
    st_prof, kt = pgo_profile(pgo, a)
    rooflines = pgo_rooflines(pgo, st, st_prof, kt, a.nodes, a.edges, is_c2)
    roofline = dict(rooflines[0]); roofline["traffic_source"] = traffic_source()
    kernels_ms = {k: round(v["ms"], 4) for k, v in sorted(kt.items(), key=lambda x: -x[1]["ms"])}
    cfg_name = {(1000, 5000): "BASELINE config 2", (10000, 50000): "BASELINE config 4 size on one GPU", (100, 300): "BASELINE config 1"}.get((a.nodes, a.edges), "custom size")
    # What a solve costs besides its PCG iterations: the same graph solved so loosely that every solve stops at its first look; the slope
    # between the two is the cost of a PCG iteration, the rest - linearise, set-up / rebuild of the hierarchy, evaluation, the host's look
    # at the loop's state - is the LM overhead (tests/diag/lm_overhead.py is the same measurement for any size)
    lm_overhead = None
    if dist.rank == 0:
        pl = capi.Pgo(device=dev, iterations=a.lm_iters, pcg_tol=1e-3, pass_history=1)
        pl.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"]); pl.optimize(a.lm_iters)
        best = 1e30
        for _ in range(5):
            pl.reset(); t0 = time.perf_counter(); stl_ = pl.optimize(a.lm_iters); best = min(best, time.perf_counter() - t0)
        pl.close()
        ms_def = 1e3 * t_pgo / a.steps
        d_it = st["pcg_iterations"] - stl_["pcg_iterations"]
        if d_it > 0:
            us_it = 1e3 * (ms_def - 1e3 * best) / d_it
            lm_overhead = dict(lm_overhead_ms=round(ms_def - 1e-3 * us_it * st["pcg_iterations"], 3), us_per_pcg_iteration=round(us_it, 3),
                               us_per_lm_trial=round(1e3 * (ms_def - 1e-3 * us_it * st["pcg_iterations"]) / max(st["lm_trials"], 1), 1),
                               loose_solve=dict(pcg_tol=1e-3, ms=round(1e3 * best, 3), pcg_iterations=stl_["pcg_iterations"], lm_trials=stl_["lm_trials"]),
                               host_looks_per_solve=st.get("lm_passes", 0),
                               note="lm_overhead_ms = ms_per_step - pcg_iterations x the slope; the LM loop's decisions run on the device (csrc/pgo_lm_kernels.hip), the "
                                    "host looks at the state once per pass (`host_looks_per_solve`)")
    # What a NEW graph costs on a live optimizer (the reference keeps one G2oOptimizer for the life of the node and rebuilds its g2o graph
    # for every optimisation, g2o_optimizer.cpp:57): add_graph of another graph of the same size + the first optimize - structure built,
    # nothing captured, no history.  `fresh_handle` is the same on a handle that has never solved (device buffers allocated on the way);
    # B["first"] the process's very first solve (code-object loading, first pinned allocations).
    pf = capi.Pgo(device=dev, iterations=a.lm_iters, pass_history=1)
    first_warm = {}
    for tag, sd in (("fresh_handle", 77), ("live_handle", 78)):
        gf = synth.make_pose_graph(a.nodes, a.edges, seed=ud.replica_seed(12345, dist.rank) + sd)
        t0 = time.perf_counter(); pf.add_graph(gf["nodes_pose"], gf["nodes_fixed"], gf["edges"]); t1 = time.perf_counter()
        stf = pf.optimize(a.lm_iters); t2 = time.perf_counter()
        first_warm[tag] = dict(first_solve_ms=round(1e3 * (t2 - t1), 3), add_graph_ms=round(1e3 * (t1 - t0), 3), structure_ms=round(stf["structure_ms"], 3),
                               lm_passes=stf["lm_passes"], pcg_iterations=stf["pcg_iterations"])
    pf.close()
    first_warm = dict(first_warm["live_handle"], fresh_handle=first_warm["fresh_handle"], first_solve_of_the_process_ms=B["first"]["first_solve_ms"])
    # the deployed operating point (iti_slam_launch/yaml/slam.yaml:50-53): optimize_xy_only = true, same graph
    Bxy = pgo_block(capi, synth, dist, dev, a, a.nodes, a.edges, max(2, a.steps // 2), 1, ud.replica_seed(12345, dist.rank), xy=True, repeat=False)
    xy_only = dict(value=round(dist.sum(float(Bxy["edges"])) / Bxy["t"], 1), unit="edges/s", ms_per_solve=round(1e3 * Bxy["t"] / max(2, a.steps // 2), 4),
                   lm_iterations_done=Bxy["st"]["iterations_done"], pcg_iterations_per_solve=Bxy["st"]["pcg_iterations"],
                   chi2_final=Bxy["st"]["chi2_final"], note="optimize_xy_only = true (poses and measurements projected to x, y, yaw; g2o_optimizer.cpp:164-170)")
    Bxy["pgo"].close()

    # ------------------------------------------------------------------ secondary: match + RANSAC
    secondary = None
    matcher = None
    pairs = None
    if not a.no_secondary:
        per_rank = a.pairs                                                     # weak scaling: fixed work per GPU
        pairs = synth.make_pairs(per_rank, n_kp=a.keypoints, desc_bytes=32, seed=777 + dist.rank)
        matcher = capi.Match(device=dev, ransac_threshold=0.1, ransac_iteration=a.hypotheses,
                             ransac_break_percentage=1.0, do_prosac=1, seed=777)

        def upload():
            return [(matcher.add_frame(f["desc"], f["pos"], f["valid"]), matcher.add_frame(t["desc"], t["pos"], t["valid"])) for f, t, _ in pairs]
        t0 = time.perf_counter()
        ids = upload()
        upload_ms = 1e3 * (time.perf_counter() - t0)
        jobs, fids = capi.Match._jobs(ids, None)
        res = np.zeros(per_rank, capi.EDGE_RESULT_DTYPE)

        def match_step():
            matcher.launch_raw(jobs, fids)
            matcher.collect(res)

        for _ in range(a.warmup):
            match_step()
        t_match = timed(dist, match_step, a.steps)
        pairs_total = dist.sum(float(per_rank * a.steps))
        # SURVEY section 8(d) defines pairs/s as submit -> results INCLUDING the H2D of the descriptors: one pass that uploads every
        # frame (pageable host arrays, one add_frame per FeatureData as the adapter does) and estimates; never `value`
        for fa, fb in ids:
            matcher.remove_frame(fa); matcher.remove_frame(fb)
        dist.barrier(); t0 = time.perf_counter()
        ids2 = upload(); jobs2, fids2 = capi.Match._jobs(ids2, None)
        matcher.launch_raw(jobs2, fids2); matcher.collect(res)
        t_incl_single = dist.max(time.perf_counter() - t0)
        for fa, fb in ids2:
            matcher.remove_frame(fa); matcher.remove_frame(fb)
        # the bulk entry point: uzl_match_add_frames over an array of uzl_frame (what the adapter's batching worker holds; building
        # that array from numpy objects is Python marshalling a C++ caller does not have, so it is done before the clock starts)
        packed = capi.Match.pack_frames([(x["desc"], x["pos"], x["valid"]) for f, t, _ in pairs for x in (f, t)])
        for fid_ in matcher.add_frames(packed):                        # warm-up: pinned staging allocated, pages touched
            matcher.remove_frame(fid_)
        dist.barrier(); t0 = time.perf_counter()
        flat = matcher.add_frames(packed)
        t_h2d_bulk = time.perf_counter() - t0
        ids3 = [(flat[2 * k], flat[2 * k + 1]) for k in range(per_rank)]
        jobs3, fids3 = capi.Match._jobs(ids3, None)
        matcher.launch_raw(jobs3, fids3); matcher.collect(res)
        t_incl = dist.max(time.perf_counter() - t0)
        jobs, fids = jobs3, fids3
        matcher.set_profiling(True)
        match_step()
        mk = matcher.kernel_times()
        matcher.set_profiling(False)
        knn_ms = mk.get("knn2", dict(ms=0.0))["ms"]; est_ms = mk.get("estimate", dict(ms=0.0))["ms"]
        is_c3 = (per_rank, a.keypoints) == (512, 1000)
        # Hamming matrix as an int8 GEMM: d = |t| + |q| - 2 <t,q>, <t,q> over D = 256 expanded bit positions
        ops = 2.0 * per_rank * a.keypoints * a.keypoints * 256.0
        ach = ops / (knn_ms * 1e-3) / 1e12 if knn_ms > 0 else 0.0
        knn_roof = roof("knn2_mfma_kernel<8, 2>", "mfma", ach, MFMA_I8_PEAK_TOPS, "TOP/s (int8, dense)", traffic=traffic_of("knn2_bytes_per_launch", is_c3),
                        traffic_source=traffic_source(), ms=round(knn_ms, 4),
                        note="v_mfma_i32_32x32x32_i8 over 0/1-expanded 256-bit descriptors (2 x 1000 x 1000 x 256 ops per pair); `peak` = 2 x the "
                             "~2.5 PFLOP/s dense bf16 rate (MI355X_MICROARCH.md); the vector ALU that folds each 32 x 32 tile into the per-query "
                             "top-2 (2 instructions per distance: the matrix cores emit the sort key) issues beside the matrix pipe")
        # RANSAC scoring: SURVEY section 8(d): hypotheses x M x 27 flop per pair (transform 18, difference 3, squared norm 5, compare 1)
        flop = 27.0 * a.hypotheses * float(res["n_corr"].sum())
        ach_e = flop / (est_ms * 1e-3) / 1e12 if est_ms > 0 else 0.0
        est_roof = roof("estimate_kernel", "f64", ach_e, F64_PEAK_TFLOPS, "TFLOP/s (f64; transform on the matrix cores, norm / compare on the vector ALU)",
                        traffic=None, ms=round(est_ms, 4), mean_correspondences=float(res["n_corr"].mean()),
                        note="27 x hypotheses x M flop per pair; the kernel also sorts, samples, fits 500 float poses, refits and scores each job, "
                             "all inside one workgroup with the point tile in LDS - `phases` says where a workgroup's time goes and what bounds each part")
        ph = estimate_phases(is_c3)
        if ph:
            est_roof["phases"] = ph
            vf = ph["fraction"].get("votes")
            if vf:
                est_roof["vote_phase_frac_of_f64_peak"] = round(est_roof["frac"] / vf, 4)
        rooflines += [knn_roof, est_roof]
        secondary = dict(metric="node-pairs matched/sec", value=round(pairs_total / t_match, 1), unit="pairs/s",
                         ms_per_step=round(1e3 * t_match / a.steps, 4),
                         config=dict(workload="BASELINE config 3: %d node pairs x %d ORB-256 descriptors per frame, "
                                              "2-NN Hamming + %d-hypothesis PROSAC, early exit off" % (per_rank, a.keypoints, a.hypotheses)),
                         upload_inclusive=dict(value=round(dist.sum(float(per_rank)) / t_incl, 1), unit="pairs/s", ms=round(1e3 * t_incl, 3),
                                               add_frames_call_ms=round(1e3 * t_h2d_bulk, 3), h2d_mbytes=round(2e-6 * per_rank * a.keypoints * (32 + 24 + 1), 1),
                                               add_frames_enqueue_gbytes_per_s=round(2e-9 * per_rank * a.keypoints * (32 + 24 + 1) / max(t_h2d_bulk, 1e-9), 2),
                                               add_frames_note="an ENQUEUE rate, not transfer bandwidth: the call returns when the frames are packed and the last DMA (up to 32 MB) is enqueued; `ms` (upload + one estimate) runs until the results are back",
                                               one_add_frame_per_frame=dict(value=round(dist.sum(float(per_rank)) / t_incl_single, 1), unit="pairs/s", ms=round(1e3 * t_incl_single, 3),
                                                                            note="uzl_match_add_frame once per frame through ctypes, as round 2 measured it"),
                                               note="uzl_match_add_frames of all %d frames from pageable host memory (threads pack into pinned staging, one DMA per 32 MB) + one "
                                                    "estimate (SURVEY 8d's definition); in the running system a frame is uploaded once per node and reused by every pair it takes part in" % (2 * per_rank)),
                         mean_consensus=float(res["consensus"].mean()), ok_fraction=float(res["ok"].mean()),
                         kernels_ms={k: round(v["ms"], 4) for k, v in mk.items()},
                         roofline=knn_roof)

    # ------------------------------------------------------------------ the estimator's DEPLOYED operating point
    if secondary is not None and not a.no_deployed:
        secondary["deployed"] = deployed_block(capi, synth, dist, dev, a)
        if secondary["deployed"].get("parity_ok") is False:
            parity_fail.append(("secondary.deployed", secondary["deployed"]["parity"]))

    # ------------------------------------------------------------------ formats: Feature records -> frame arena
    formats = None
    if matcher is not None and dist.rank == 0 and not a.no_formats:
        formats = bench_formats(capi, dev, pairs, a.keypoints)

    # ------------------------------------------------------------------ batched: B independent config-2 graphs, one batch call
    batched = None
    if not a.no_batched and is_c2:
        nB = a.batch
        tb0 = time.perf_counter()
        bt = capi.PgoBatch(nB, device=dev, iterations=a.lm_iters, pass_history=1)
        batch_create_ms = 1e3 * (time.perf_counter() - tb0)
        for k in range(nB):
            gk = synth.make_pose_graph(a.nodes, a.edges, seed=ud.replica_seed(12345, dist.rank) + 1000 * k)
            bt.graphs[k].add_graph(gk["nodes_pose"], gk["nodes_fixed"], gk["edges"])
        wb = {"edges": 0}

        def batch_step():
            for p_ in bt.graphs:
                p_.reset()
            for st_ in bt.optimize(a.lm_iters):
                wb["edges"] += st_["n_edges"] * st_["iterations_done"]

        batch_step(); wb["edges"] = 0
        nsteps_b = max(3, a.steps // 2)
        t_b = timed(dist, batch_step, nsteps_b)
        vb = dist.sum(float(wb["edges"])) / t_b
        for p_ in bt.graphs:                                                  # the same batch with the per-trial memory on (`repeat_identical`)
            p_.set_config(pass_history=0)
        batch_step(); batch_step(); eb0 = wb["edges"]
        t_br = timed(dist, batch_step, nsteps_b)
        vb_rep = dist.sum(float(wb["edges"] - eb0)) / t_br
        for p_ in bt.graphs:
            p_.set_config(pass_history=1)
        # the same PCG kernel bodies with the chip full: their rate against the HBM roof (the working set streams from the Infinity Cache)
        bt.set_profiling(True)
        for p_ in bt.graphs:
            p_.reset()
        stp = bt.optimize(a.lm_iters)
        ktb = bt.kernel_times()
        bt.set_profiling(False)
        b_roofs = []
        nbf = st["n_vertices"] - int(pgo.get_fixed().sum())
        sp_b = ktb.get("ml_spmv_batch")
        if sp_b and sp_b["ms"] > 0:
            # a launch does work for the graphs still iterating: sum of the graphs' PCG iterations x per-graph bytes over the kernel's time
            work = sum(x["pcg_iterations"] for x in stp)
            alg1 = 288.0 * (nbf + st["n_edges"]) + 96.0 * nbf
            ach_b = alg1 * work / (sp_b["ms"] * 1e-3) / 1e9
            b_roofs.append(roof("ml_spmv_batch_kernel", "hbm", ach_b, HBM_PEAK_GBS, "GB/s", traffic=None, avg_launch_us=round(1e3 * sp_b["ms"] / sp_b["launches"], 3),
                                launches=sp_b["launches"], algorithmic_bytes_per_graph_iteration=alg1))
        cg_b = ktb.get("ml_cg_comp_batch")
        if cg_b and cg_b["ms"] > 0:
            n1b = (nbf + 7) // 8
            alg_y = 4.0 * (6.0 * n1b) ** 2 + 8.0 * 48 * 48 * n1b + 10 * 48.0 * nbf      # dense level-1 operator (f32 copy) + sibling blocks + vectors, per graph iteration
            ach_c = alg_y * sum(x["pcg_iterations"] for x in stp) / (cg_b["ms"] * 1e-3) / 1e9
            b_roofs.append(roof("ml_cg_comp_batch_kernel<5>", "hbm", ach_c, HBM_PEAK_GBS, "GB/s", traffic=None, avg_launch_us=round(1e3 * cg_b["ms"] / cg_b["launches"], 3),
                                launches=cg_b["launches"], algorithmic_bytes_per_graph_iteration=alg_y,
                                note="every workgroup streams its 6 rows of the graph's dense level-1 operator (f32 copy, 2.25 MB per graph) and its 48 x 48 smoother block"))
        batched = dict(metric="SE(3) edges optimized/sec, %d independent config-2 graphs per GPU in one batch call (uzl_pgo_batch_*)" % nB,
                       rooflines=b_roofs,
                       value=round(vb, 1), unit="edges/s", graphs=nB, graphs_batched=bt.n_batched, ms_per_batch=round(1e3 * t_b / nsteps_b, 3),
                       repeat_identical=dict(value=round(vb_rep, 1), ms_per_batch=round(1e3 * t_br / nsteps_b, 3)), batch_create_ms=round(batch_create_ms, 3),
                       ms_per_graph=round(1e3 * t_b / nsteps_b / nB, 4), vs_single_graph=round(vb / value, 2),
                       note="every graph's poses are bit-identical to its own uzl_pgo_optimize (tests/test_batch_gpu.py); the single-graph figure is `value`")
        bt.close()
        # a queue of config-2 graphs through 16 / 64 resident slots: a finished graph hands its slot to the next one (uzl_pgo_batch_set_resident)
        nQ = a.batch_queue
        if nQ > 0:
            bq = capi.PgoBatch(nQ, device=dev, iterations=a.lm_iters, pass_history=1)
            for k in range(nQ):
                gk = synth.make_pose_graph(a.nodes, a.edges, seed=ud.replica_seed(12345, dist.rank) + 1000 * k)
                bq.graphs[k].add_graph(gk["nodes_pose"], gk["nodes_fixed"], gk["edges"])
            batched["queue"] = dict(graphs=nQ, workload="%d config-2 graphs queued, R resident at a time" % nQ,
                                    policy="cohorts: the R resident graphs advance in step, the slots are refilled when all of them are through "
                                           "(refilling slot by slot as graphs finish was built and measured slower: DESIGN.md section 7)")
            for R_ in (16, 64):
                if R_ > nQ:
                    continue
                bq.set_resident(R_)
                for p_ in bq.graphs:
                    p_.reset()
                bq.optimize(a.lm_iters)                                   # warm-up: structures, capture for this slot count
                for p_ in bq.graphs:
                    p_.reset()
                dist.barrier(); t0 = time.perf_counter()
                sts = bq.optimize(a.lm_iters)
                dist.sync(); tq = dist.max(time.perf_counter() - t0)
                eq = dist.sum(float(sum(x["n_edges"] * x["iterations_done"] for x in sts)))
                batched["queue"]["resident_%d" % R_] = dict(value=round(eq / tq, 1), unit="edges/s", ms=round(1e3 * tq, 2), graphs_batched=bq.n_batched,
                                                             vs_single_graph=round(eq / tq / value, 2),
                                                             lm_trials_min_max=[min(x["lm_trials"] for x in sts), max(x["lm_trials"] for x in sts)])
            bq.close()
        # the regime batching is for: many small graphs (BASELINE config 1 size: 100 nodes / 300 edges - local scopes, per-robot graphs)
        nS = 64
        g1 = [synth.make_pose_graph(100, 300, seed=ud.replica_seed(777, dist.rank) + 1000 * k) for k in range(nS)]
        one = capi.Pgo(device=dev, iterations=a.lm_iters, pass_history=1)
        one.add_graph(g1[0]["nodes_pose"], g1[0]["nodes_fixed"], g1[0]["edges"]); one.optimize(a.lm_iters)
        t0 = time.perf_counter(); e1 = 0
        for _ in range(5):
            one.reset(); st_ = one.optimize(a.lm_iters); e1 += st_["n_edges"] * st_["iterations_done"]
        t_one = time.perf_counter() - t0
        one.close()
        bs = capi.PgoBatch(nS, device=dev, iterations=a.lm_iters, pass_history=1)
        for k in range(nS):
            bs.graphs[k].add_graph(g1[k]["nodes_pose"], g1[k]["nodes_fixed"], g1[k]["edges"])
        bs.optimize(a.lm_iters)
        t0 = time.perf_counter(); eS = 0
        for _ in range(5):
            for p_ in bs.graphs:
                p_.reset()
            for st_ in bs.optimize(a.lm_iters):
                eS += st_["n_edges"] * st_["iterations_done"]
        t_S = time.perf_counter() - t0
        batched["small_graphs"] = dict(workload="%d graphs of BASELINE config 1 size (100 nodes / 300 edges), %d LM iterations" % (nS, a.lm_iters),
                                       value=round(eS / t_S, 1), unit="edges/s", graphs_batched=bs.n_batched, ms_per_batch=round(1e3 * t_S / 5, 3),
                                       one_graph_alone=round(e1 / t_one, 1), vs_one_graph_alone=round((eS / t_S) / (e1 / t_one), 1))
        bs.close()
        # graphs of the reference's own shape - an odometry chain plus a few loop closures (graph_slam_node.cpp:578-663, local scopes): their
        # chain interiors are Schur-eliminated and they batch on their reduced systems (round 3 sent such a batch one by one through the single path)
        nC = 16
        gc = [synth.make_pose_graph(1500, 1530, seed=ud.replica_seed(4040, dist.rank) + k) for k in range(nC)]
        one = capi.Pgo(device=dev, iterations=a.lm_iters, pass_history=1)
        one.add_graph(gc[0]["nodes_pose"], gc[0]["nodes_fixed"], gc[0]["edges"]); one.optimize(a.lm_iters)
        t0 = time.perf_counter(); e1 = 0
        for _ in range(5):
            one.reset(); st_ = one.optimize(a.lm_iters); e1 += st_["n_edges"] * st_["iterations_done"]
        t_one = time.perf_counter() - t0
        one.close()
        bc = capi.PgoBatch(nC, device=dev, iterations=a.lm_iters, pass_history=1)
        for k in range(nC):
            bc.graphs[k].add_graph(gc[k]["nodes_pose"], gc[k]["nodes_fixed"], gc[k]["edges"])
        bc.optimize(a.lm_iters)

        def chain_rounds(n_):
            t0_ = time.perf_counter(); e_ = 0; st_ = None
            for _ in range(n_):
                for p_ in bc.graphs:
                    p_.reset()
                st_ = bc.optimize(a.lm_iters)
                e_ += sum(x["n_edges"] * x["iterations_done"] for x in st_)
            return time.perf_counter() - t0_, e_, st_
        t_C, eC, stc = chain_rounds(5)
        for p_ in bc.graphs:
            p_.set_config(pass_history=0)
        chain_rounds(2)
        t_Cr, eCr, _ = chain_rounds(5)
        batched["chain_like"] = dict(workload="%d chain-like graphs (1500 nodes / 1530 edges: an odometry chain + 31 loop closures), %d LM iterations" % (nC, a.lm_iters),
                                     value=round(eC / t_C, 1), unit="edges/s", graphs_batched=bc.n_batched, ms_per_batch=round(1e3 * t_C / 5, 3),
                                     repeat_identical=dict(value=round(eCr / t_Cr, 1), ms_per_batch=round(1e3 * t_Cr / 5, 3)),
                                     vertices_schur_eliminated=[int(x["n_eliminated"]) for x in stc][:4] + ["..."],
                                     one_graph_alone=round(e1 / t_one, 1), vs_one_graph_alone=round((eC / t_C) / (e1 / t_one), 1))
        bc.close()

    # ------------------------------------------------------------------ north star: 10k / 50k on ONE GPU (N = 1 only)
    c4 = None
    if dist.world == 1 and not a.no_c4 and is_c2:
        steps4 = max(3, a.steps // 3)
        B4 = pgo_block(capi, synth, dist, dev, a, 10000, 50000, steps4, 1, 12345)
        st4p, kt4 = pgo_profile(B4["pgo"], a)
        r4 = pgo_rooflines(B4["pgo"], B4["st"], st4p, kt4, 10000, 50000, True)
        for r in r4:
            r["traffic_source"] = traffic_source()
        c4 = dict(metric="SE(3) edges optimized/sec, 10k nodes / 50k edges, one GPU", value=round(B4["edges"] / B4["t"], 1), unit="edges/s",
                  ms_per_solve=round(1e3 * B4["t"] / steps4, 3), solves_timed=steps4, h2d_ms=round(B4["h2d_ms"], 3), d2h_ms=round(B4["d2h_ms"], 3),
                  first_solve=B4["first"], repeat_identical=B4.get("repeat"),
                  lm_iterations_done=B4["st"]["iterations_done"], pcg_iterations_per_solve=B4["st"]["pcg_iterations"],
                  chi2_initial=B4["st"]["chi2_initial"], chi2_final=B4["st"]["chi2_final"], roofline=r4[0], rooflines=r4,
                  kernels_ms_per_solve={k: round(v["ms"], 4) for k, v in sorted(kt4.items(), key=lambda x: -x[1]["ms"])})
        if not a.no_cpu_baseline:
            import oracle as O
            ncpu = effective_cpus()
            nth = min(ncpu, 16)                                                # OpenMP over 50k edges: more threads only add fork/join cost
            cb = cpu_pgo(O, B4["g"], a, 1.0, [1, nth], max_solves=1)           # one solve each: ~20 s per solve on one core
            c4["cpu_baseline"] = dict(value=cb[1]["value"], unit="edges/s", cores=1, kind="port", seconds_per_solve=cb[1]["seconds_per_solve"],
                                      all_cores=dict(value=cb[nth]["value"], cores=nth, seconds_per_solve=cb[nth]["seconds_per_solve"],
                                                     note="OpenMP over the edges as in a g2o built with it (%d of the host's %d cores: the edge loops are %.0f %% of the "
                                                          "solve, more threads only add fork/join cost); the sparse Cholesky (%.0f %%) is serial, as CSparse is"
                                                          % (nth, ncpu, 100 * (1 - cb[1]["cholesky_share"]), 100 * cb[1]["cholesky_share"])),
                                      nproc=ncpu, hardware_threads=os.cpu_count(), cpu=cpu_model(), sample="1 solve per thread count of the same graph, %d LM iterations" % a.lm_iters,
                                      build="gcc -O3 -march=native -fopenmp on this host")
            c4["parity"] = parity_block(synth, B4["poses"], B4["st"], cb)
            if not c4["parity"]["ok"]:
                parity_fail.append(("c4_1gpu", c4["parity"]))
            c4["speedup_vs_cpu_1_thread"] = round(cb[1]["seconds_per_solve"] / (B4["t"] / steps4), 1)
            c4["speedup_vs_cpu_all_cores"] = round(cb[nth]["seconds_per_solve"] / (B4["t"] / steps4), 1)
        B4["pgo"].close()
        if not a.no_sharded:
            env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("NCCL_SOCKET_IFNAME", "lo")
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--sharded-world1-child", "--steps", str(a.steps), "--lm-iters", str(a.lm_iters)],
                                   env=env, capture_output=True, text=True, timeout=240)
                w1 = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else dict(error=(r.stderr or r.stdout)[-400:])
            except Exception as ex:                                            # timeout / no JSON: reported, not fatal
                w1 = dict(error=repr(ex)[:400])
            if "ms_per_solve" in w1:
                w1["vs_graph_captured_solve"] = round(w1["ms_per_solve"] / c4["ms_per_solve"], 3)
                w1["note"] = ("same graph, same kernels, world_size 1: the ratio is the cost of the sharded path's structure on one GPU "
                              "(RCCL call per PCG iteration, block-diagonal level-0 smoother, synchronous rebuilds)")
            c4["sharded_world1"] = w1

    # ------------------------------------------------------------------ BASELINE config 5: match jobs feeding a growing graph
    online_c5 = None
    if not a.no_online and is_c2:
        from uzliti_slam_amd import online
        run = synth.make_online_run(a.online_nodes, a.online_pairs, n_kp=a.keypoints)
        o = online.OnlineSlam(run, device=dev, rank=dist.rank, world=dist.world, tdist=dist.dist, match_batch=512)
        o.keep_poses_per_solve = 1000                   # the CPU replay (cpu_baseline below) is compared with the poses at its last interval
        o.upload_frames()
        dist.barrier(); t0 = time.perf_counter()
        o.run_all()
        dist.sync(); wall = dist.max(time.perf_counter() - t0)
        if dist.rank == 0:
            s = o.summary(wall)
            gt = run["gt"]
            online_c5 = dict(metric="BASELINE config 5: %d node-pair jobs -> acceptance gate -> edge filter -> graph growing to %d nodes, re-optimised every 256 edges"
                                    % (a.online_pairs, a.online_nodes),
                             wall_s=round(wall, 3), pairs_per_s=round(s["pairs_per_s_wall"], 1), edges_per_s=round(s["edges_per_s_wall"], 1),
                             edges_per_s_inside_optimize=round(s["edges_per_s_solver"], 1), solves=s["n_solves"],
                             add_graph_ms_per_solve=round(s["add_graph_ms_per_solve"], 3), structure_ms_per_solve=round(s["structure_ms_per_solve"], 3),
                             optimize_ms_per_solve=round(s["optimize_ms_per_solve"], 3), seconds=s["seconds"], feature_edges_accepted=s["feature_edges_accepted"],
                             feature_edges_valid=s["feature_edges_valid"], pcg_iterations=s["pcg_iterations"], lm_iterations=s["lm_iterations"],
                             vertices_schur_eliminated_last_solve=o.solves[-1].get("n_eliminated", 0) if o.solves else 0,
                             solves_reduced=s.get("solves_reduced"), solves_strong_aggregates=s.get("solves_strong_aggregates"),
                             not_converged=s["not_converged"],
                             ate_dead_reckoning_m=round(float(np.linalg.norm(run["init"][:, :, 3] - gt[:, :, 3], axis=1).mean()), 3),
                             ate_online_m=round(float(np.linalg.norm(o.poses[:, :, 3] - gt[:, :, 3], axis=1).mean()), 3),
                             parallelism="pair jobs sharded over %d rank(s) per batch of 512, results gathered in job order; gate, filter and solver on rank 0, "
                                         "the next batch's matching in flight during the solve" % dist.world,
                             note="rebuild per re-optimise = add_graph (upload + flattening) + structure (gauge, block-CSR, Schur plan, hierarchy, graph capture)")
            # ---- roofline of the kernel this config adds: the Schur elimination of the chain interiors (HBM-bound by construction:
            #      per eliminated vertex it reads 3 blocks of 288 B + b, writes u | W | T = 624 B; a run is a chain of <= 24 dependent steps)
            pl = capi.Pgo(device=dev, iterations=a.lm_iters)
            pl.add_graph(*o.last_input)
            pl.optimize(a.lm_iters)
            pl.set_profiling(True); pl.add_graph(*o.last_input); stl = pl.optimize(a.lm_iters); ktl = pl.kernel_times(); pl.close()
            el = ktl.get("schur_eliminate")
            if el and el["ms"] > 0 and stl["n_eliminated"] > 0:
                alg_e = float(stl["n_eliminated"]) * (3 * 288.0 + 48.0 + 624.0)
                online_c5["roofline"] = roof("schur_eliminate_kernel", "hbm", alg_e * el["launches"] / (el["ms"] * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s", traffic=None,
                                             algorithmic_bytes_per_launch=alg_e, avg_launch_us=round(1e3 * el["ms"] / el["launches"], 3), launches=el["launches"],
                                             vertices_eliminated=stl["n_eliminated"],
                                             note="last re-optimisation of the run (%d vertices, %d edges): one wave per run of <= 24 chain interiors, a chain of dependent 6x6 "
                                                  "inversions and products - latency-bound, not bandwidth-bound, at this size" % (stl["n_vertices"], stl["n_edges"]))
                online_c5["last_solve_kernels_ms"] = {k: round(v["ms"], 4) for k, v in sorted(ktl.items(), key=lambda x: -x[1]["ms"])[:8]}
            # ---- the CPU path on the same run: tests/online_stubs.py drives the SAME schedule through the CPU checker's estimator, gate,
            #      filter and solver (one thread, as the reference's plugins run).  Budget-capped; compared on the common prefix of intervals.
            if dist.world == 1 and not a.no_cpu_baseline:
                import oracle as O
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from online_stubs import oracle_online
                c = oracle_online(O, run, ransac_iteration=500, match_batch=512)
                c.upload_frames()
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < a.online_cpu_seconds and c.step():
                    pass
                k = len(c.solves)
                if k >= 1 and len(o.solves) >= k:
                    cpu_s = c.solves[k - 1]["wall_s"]; gpu_s = o.solves[k - 1]["wall_s"]
                    same = bool(c.solves[k - 1]["n_nodes"] == o.solves[k - 1]["n_nodes"] and c.solves[k - 1]["n_edges"] == o.solves[k - 1]["n_edges"])
                    dtc, drc = synth.pose_errors(o.poses_at_solve[k - 1], c.poses[:c.solves[k - 1]["n_nodes"]]) if same and len(o.poses_at_solve) >= k else (None, None)
                    online_c5["cpu_baseline"] = dict(
                        kind="port", cores=1, unit="s", value=round(cpu_s, 3),
                        sample="the first %d of %d re-optimisation intervals of the same run (graph grown to %d nodes), %.1f s budget; oracle = CPU checker's estimator "
                               "(-O3, portable build), gate, filter and LM + sparse Cholesky, one thread" % (k, len(o.solves), c.solves[k - 1]["n_nodes"], a.online_cpu_seconds),
                        gpu_same_prefix_s=round(gpu_s, 3), speedup_on_prefix=round(cpu_s / max(gpu_s, 1e-9), 1), same_graph_at_that_point=same,
                        pose_difference_at_that_point=dict(dt_m=dtc, dr_rad=drc),
                        extrapolated_full_run_s=round(cpu_s * len(o.solves) / k, 1),
                        extrapolation="linear in the number of intervals: a LOWER bound (later intervals hold larger graphs and cost more)",
                        nproc=effective_cpus(), cpu=cpu_model())
                    if dtc is not None:
                        ok5 = bool(dtc < PARITY_T and drc < PARITY_R)
                        online_c5["cpu_baseline"]["pose_difference_at_that_point"].update(ok=ok5, bar=dict(dt_m=PARITY_T, dr_rad=PARITY_R),
                            note="every interval starts from the previous interval's result: the difference is what twenty LM iterations leave of the earlier "
                                 "ones' plus this interval's; tests/diag/c5_tolerance.py shows it follows the linear solver's accuracy (default stop test against "
                                 "a tightly solved run: median 4e-6 m, largest 2.2e-4 m over the 94 intervals)")
                        if not ok5:
                            parity_fail.append(("online_c5", online_c5["cpu_baseline"]["pose_difference_at_that_point"]))
                c.close()
        o.close()

    # ------------------------------------------------------------------ optional: config 4, one graph sharded over the ranks
    sharded_c4 = None
    if dist.world > 1 and not a.no_sharded and (not dist.rehearsal or os.environ.get("UZL_BENCH_FORCE_SHARDED") == "1"):      # (the env switch: rehearsal of this block's control flow on one GPU)
        # (a child job with a time limit: see sharded_child; the other ranks wait at the barrier and leave their GPUs to it)
        # the other ranks wait on the rendezvous store (a CPU wait: an NCCL barrier would park a spinning kernel on every GPU the child uses)
        store = None
        try:
            from torch.distributed import distributed_c10d as _c10d
            store = _c10d._get_default_store()
        except Exception:
            store = None
        if dist.rank == 0:
            sharded_c4 = run_sharded_child(a, dist.world)
            if store is not None:
                store.set("uzl_sharded_child_done", "1")
        elif store is not None:
            import datetime
            try:
                store.wait(["uzl_sharded_child_done"], datetime.timedelta(seconds=400))
            except Exception:
                pass
        dist.barrier()

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1 only)
    cpu = None
    parity = None
    if dist.rank == 0 and dist.world == 1 and not a.no_cpu_baseline:
        import oracle as O
        ncpu = effective_cpus()
        nth = min(ncpu, 8)
        cb = cpu_pgo(O, g, a, a.cpu_seconds, [1, nth])
        parity = parity_block(synth, B["poses"], st, cb)
        if not parity["ok"]:
            parity_fail.append(("primary", parity))
        cpu = dict(value=cb[1]["value"], unit="edges/s", cores=1, kind="port",
                   sample="%d solve(s) of the same %d-node/%d-edge graph, %d LM iterations each, %.1f s; "
                          "oracle = C restatement of g2o LM + block sparse direct Cholesky (reference binaries not buildable here), gcc -O3 -march=native -fopenmp on this host"
                          % (cb[1]["solves"], a.nodes, a.edges, a.lm_iters, cb[1]["seconds"]),
                   all_cores=dict(value=cb[nth]["value"], cores=nth, note="OpenMP over the edges (g2o's own parallelism; %d of %d cores - 5000 edges do not feed more); "
                                                                           "the Cholesky (%.0f %% of the solve) is serial as CSparse is" % (nth, ncpu, 100 * cb[1]["cholesky_share"])),
                   nproc=ncpu, hardware_threads=os.cpu_count(), cpu=cpu_model())
        if secondary is not None:
            kw = dict(ransac_threshold=0.1, ransac_iteration=a.hypotheses, break_percentage=1.0, do_prosac=True, seed=777)
            fp = [(f, t) for f, t, _ in pairs]
            chunks = [O.PreparedPairs(fp[k:k + 32]) for k in range(0, len(fp), 32)]      # ctypes structs built outside the timed loops
            n1 = 0; t0 = time.perf_counter()                         # one estimator thread, as one plugin instance runs
            while time.perf_counter() - t0 < a.cpu_seconds and n1 < 16384:
                ch = chunks[(n1 // 32) % len(chunks)]
                O.estimate_edge_batch(ch, job_id0=n1, threads=1, **kw); n1 += ch.n
            dtm = time.perf_counter() - t0
            # all cores: one estimator thread per core over independent pairs (OpenMP inside the native build), bounded sample
            n_all = 16384                                                    # a few seconds on a many-core host
            big = O.PreparedPairs([fp[k % len(fp)] for k in range(n_all)])
            O.estimate_edge_batch(O.PreparedPairs(fp[:ncpu]), job_id0=0, threads=ncpu, **kw)      # thread pool up
            dta = 1e30
            for _ in range(2):                                               # best of two: the first run after a 1-thread region starts slowly
                t0 = time.perf_counter()
                O.estimate_edge_batch(big, job_id0=0, threads=ncpu, **kw)
                dta = min(dta, time.perf_counter() - t0)
            secondary["cpu_baseline"] = dict(value=round(n1 / dtm, 2), unit="pairs/s", cores=1, kind="port",
                                             sample="%d node pairs drawn cyclically from the same %d, %.1f s; gcc -O3 -march=native on this host" % (n1, len(pairs), dtm),
                                             all_cores=dict(value=round(n_all / dta, 2), cores=ncpu, sample="%d pairs over %d OpenMP threads (one estimator per core), %.1f s" % (n_all, ncpu, dta)))

    if dist.rank == 0:
        out = dict(
            metric="SE(3) edges optimized/sec (node-pairs matched/sec in `secondary`)",
            value=round(value, 1), unit="edges/s", n_gpus=dist.world, steps=a.steps, warmup=a.warmup,
            ms_per_step=round(1e3 * t_pgo / a.steps, 4), higher_is_better=True, scaling="weak", vs_baseline=None,
            dtype="f64", data="synthetic" if not dist.rehearsal else "synthetic (REHEARSAL over gloo, ranks sharing devices: not a measurement)",
            config=dict(workload="%s: %d-node / %d-edge SE(3) pose graph, %d LM iterations, Huber(1) on loop closures; "
                                 "one independent graph per GPU" % (cfg_name, a.nodes, a.edges, a.lm_iters),
                        system_edges=st["n_edges"], lm_iterations_done=st["iterations_done"], lm_trials_per_solve=st["lm_trials"],
                        pcg_iterations_per_solve=st["pcg_iterations"], preconditioner_builds_per_solve=st["precond_builds"], pcg_tol=pgo.cfg.pcg_tol,
                        preconditioner=("multilevel, 8-vertex rigid-body aggregates (small graphs: dense level-1 operator, multiplicative cycle + 2 Newton-Schulz steps on the f64 matrix cores; "
                                        "the PCG kernels apply an f32 COPY of that dense coarse operator (ml_cmat32_kernel) with f64 accumulation - preconditioner only: x, r, p, A p and every "
                                        "reduction are f64, the converged solution does not depend on it)" if pgo.cfg.preconditioner else "block-Jacobi"),
                        chi2_initial=st["chi2_initial"], chi2_final=st["chi2_final"]),
            h2d_ms=round(B["h2d_ms"], 3), d2h_ms=round(B["d2h_ms"], 3),
            timing="uzl_pgo_cfg::pass_history = 1: no pass of the LM loop is sized from an earlier optimize of the same graph",
            first_solve_ms=first_warm["first_solve_ms"], first_solve=first_warm, repeat_identical=B.get("repeat"), continued_from_solution=B.get("continued"),
            streams=capi.stream_stats(dev),
            roofline=roofline, rooflines=rooflines, traffic_source=traffic_source() + " (rocprofv3 --pmc passes of profiles/collect.sh on the default workloads; not measured in this run)",
            kernels_ms_per_solve=kernels_ms, kernels_ms_note="profiled solve: eager launches of the by-value instantiations of the kernel bodies (host-driven loop); "
                                                            "the timed solves run the same bodies as slot twins (ml_spmv_lm_kernel, ...) under the device-resident loop",
            lm_overhead_ms=(lm_overhead or {}).get("lm_overhead_ms"), lm_overhead=lm_overhead,
            cpu_baseline=cpu, parity=parity, xy_only=xy_only, secondary=secondary)
        for k, v in (("batched", batched), ("formats", formats), ("c4_1gpu", c4), ("online_c5", online_c5), ("sharded_c4", sharded_c4)):
            if v is not None:
                out[k] = v
        emit(out)
    pgo.close()
    if matcher is not None:
        matcher.close()
    dist.close()
    if parity_fail:
        print("[bench] PARITY MISS: %s" % json.dumps(parity_fail), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
