#!/bin/bash
# builds build/diag/ns_gemm_lab from tests/diag/ns_gemm_lab.hip + the library's own GEMM kernel (cut out of csrc/pgo_ml_kernels.hip)
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/diag
S=uzliti_slam_amd/csrc/pgo_ml_kernels.hip
a=$(grep -n '^constexpr int kGemmTile = 64, kGemmK = 64;' $S | cut -d: -f1)
b=$(grep -n '^// The same product for ONE small graph' $S | cut -d: -f1)
sed -n "${a},$((b-1))p" $S > build/diag/ns_gemm_lab_old.inc
cp tests/diag/ns_gemm_lab_new.inc build/diag/ns_gemm_lab_new.inc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Ibuild/diag tests/diag/ns_gemm_lab.hip -o build/diag/ns_gemm_lab
