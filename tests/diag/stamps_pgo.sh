#!/bin/bash
# Diagnostic: where ml_cg / ml_spmv spend their time (s_memrealtime stamps of workgroup 0, -DUZL_STAMPS build of pgo_ml_kernels.hip).
#   bash tests/diag/stamps_pgo.sh 20000 100000        (run on the GPU box, from the repo root, after `make -C uzliti_slam_amd/csrc`)
set -e
N=${1:-10000}; E=${2:-50000}
cd uzliti_slam_amd/csrc
HIPCC=/opt/rocm/bin/hipcc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt"
mkdir -p ../../build/stb
$HIPCC $FL -DUZL_STAMPS -c pgo_ml_kernels.hip -o ../../build/stb/pgo_ml_kernels.o
OBJS=$(ls *.o | grep -v pgo_ml_kernels.o)
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../../build/stb/libuzl_stamps.so ../../build/stb/pgo_ml_kernels.o $OBJS
cd ../..
UZL_LIB=$PWD/build/stb/libuzl_stamps.so python - $N $E <<'PY'
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
from uzliti_slam_amd import capi, synth
L = capi.lib()
n, e = int(sys.argv[1]), int(sys.argv[2])
g = synth.make_pose_graph(n, e)
p = capi.Pgo()
p.add_graph(g["nodes_pose"], g["nodes_fixed"], g["edges"])
p.optimize(20)
out = (C.c_ulonglong * 64)()
L.uzl_debug_read_stamps(out, 1)
p.reset(); st = p.optimize(20)
L.uzl_debug_read_stamps(out, 0)
cg = ["entry", "prefetch issue", "p.Ap reduction", "gather-level residual -> LDS", "coarse product / walk", "x, r, zJ, w", "exact r1", "y1, z", "block sum + stores"]
sp = ["prefetch issue", "r.z reduction", "row products + folds", "restriction + stores"]
nc, ns = max(out[31], 1), max(out[47], 1)
print("n %d e %d pcg %d; ml_cg launches stamped %d, ml_spmv %d (100 MHz ticks -> us)" % (n, e, st["pcg_iterations"], nc, ns))
for i, nm in enumerate(cg):
    print("  ml_cg   %-32s %7.2f us" % (nm, out[i] / nc / 100.0))
print("  ml_cg   total %.2f us" % (sum(out[i] for i in range(9)) / nc / 100.0))
for i, nm in enumerate(sp):
    print("  ml_spmv %-32s %7.2f us" % (nm, out[16 + i] / ns / 100.0))
print("  ml_spmv total %.2f us" % (sum(out[16 + i] for i in range(4)) / ns / 100.0))
PY
