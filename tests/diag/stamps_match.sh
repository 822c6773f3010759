#!/bin/bash
# Diagnostic: where estimate_kernel spends its time (s_memrealtime stamps of thread 0 of every eighth workgroup, -DUZL_STAMPS build of
# match_kernels.hip), on bench.py's secondary workload (BASELINE config 3: 512 pairs x 1000 keypoints x 500 hypotheses).
#   bash tests/diag/stamps_match.sh [out.json]     (run on the GPU box, from the repo root, after `make -C uzliti_slam_amd/csrc`)
set -e
OUT=${1:-gpurun_out/estimate_phases.json}
cd uzliti_slam_amd/csrc
HIPCC=/opt/rocm/bin/hipcc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off"
mkdir -p ../../build/stb
$HIPCC $FL -DUZL_STAMPS -mllvm -amdgpu-mfma-vgpr-form -c match_kernels.hip -o ../../build/stb/match_kernels.o
OBJS=$(ls *.o | grep -v match_kernels.o)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../build/stb/libuzl_stamps_match.so ../../build/stb/match_kernels.o $OBJS
cd ../..
mkdir -p $(dirname $OUT)
UZL_LIB=$PWD/build/stb/libuzl_stamps_match.so python - $OUT <<'PY'
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from uzliti_slam_amd import capi, synth
L = capi.lib()
n_pairs, n_kp, hyp = 512, 1000, 500
pairs = synth.make_pairs(n_pairs, n_kp=n_kp, desc_bytes=32, seed=777)
m = capi.Match(device=0, ransac_threshold=0.1, ransac_iteration=hyp, ransac_break_percentage=1.0, do_prosac=1, seed=777)
ids = [(m.add_frame(f["desc"], f["pos"], f["valid"]), m.add_frame(t["desc"], t["pos"], t["valid"])) for f, t, _ in pairs]
jobs, fids = capi.Match._jobs(ids, None)
res = np.zeros(n_pairs, capi.EDGE_RESULT_DTYPE)
out = (C.c_ulonglong * 32)()
for _ in range(3):
    m.launch_raw(jobs, fids); m.collect(res)
L.uzl_debug_read_mstamps(out, 1)
reps = 10
for _ in range(reps):
    m.launch_raw(jobs, fids); m.collect(res)
L.uzl_debug_read_mstamps(out, 0)
names = ["sensor-pair selection + ratio / valid compaction", "bitonic sort of (distance, queryIdx)", "gather of the 3-D points into the LDS tile",
         "hypotheses: sample + float pose (running covariance, Jacobi SVD)", "hypotheses: votes (f64 matrix cores + vector ALU)", "hypotheses: early-exit bookkeeping",
         "winning hypothesis + its inlier mask", "refit: ordered compaction + sequential running covariance + SVD", "recount + mse (index-order sum)"]
nb = max(out[31], 1)
order = [0, 1, 2, 8, 9, 10, 4, 5, 6]
out[10] += out[3]          # (what is left between the last round's bookkeeping and the end of the phase)
us = [out[i] / nb / 100.0 for i in order]
tot = sum(us)
print("estimate_kernel: %d workgroups stamped over %d launches; mean correspondences %.1f, mean consensus %.1f" % (nb, reps, res["n_corr"].mean(), res["consensus"].mean()))
for nm, u in zip(names, us):
    print("  %-78s %7.2f us  %5.1f %%" % (nm, u, 100 * u / tot))
print("  total per workgroup %.2f us" % tot)
json.dump({"workload": "%d pairs x %d keypoints x %d hypotheses" % (n_pairs, n_kp, hyp), "workgroups_stamped": int(nb), "mean_correspondences": float(res["n_corr"].mean()),
           "mean_consensus": float(res["consensus"].mean()), "phase_us_per_workgroup": dict(zip(names, us)), "total_us_per_workgroup": tot,
           "source": "tests/diag/stamps_match.sh (s_memrealtime stamps, thread 0 of every eighth workgroup)"}, open(sys.argv[1], "w"), indent=1)
PY
