mkdir -p gpurun_out/r4
export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so LOOPS=0 REPS=4 UZL_PHASES=1 UZL_PHASES_EACH=1
echo "--- default"; python tests/diag/lm_passes.py 1000:5000 2>&1 | tail -32
echo "--- UZL_LM_NO_RUN_AHEAD=1"; UZL_LM_NO_RUN_AHEAD=1 python tests/diag/lm_passes.py 1000:5000 2>&1 | tail -32
