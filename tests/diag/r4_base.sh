set -e
mkdir -p gpurun_out/r4
python tests/diag/c2_repeat.py > gpurun_out/r4/base_c2_repeat.log 2>&1
python tests/diag/lm_overhead.py 100:300 1000:5000 10000:50000 > gpurun_out/r4/base_lm_overhead.log 2>&1
UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so UZL_PHASES=1 UZL_PHASES_EACH=1 python tests/diag/c2_repeat.py > gpurun_out/r4/base_phases.log 2>&1
cat gpurun_out/r4/base_c2_repeat.log gpurun_out/r4/base_lm_overhead.log
tail -8 gpurun_out/r4/base_phases.log
