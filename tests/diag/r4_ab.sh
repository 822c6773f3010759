export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so LOOPS=0 REPS=10
echo "--- default"; python tests/diag/lm_passes.py 1000:5000 100:300 10000:50000 3000:3300 2>&1 | grep lm_loop
echo "--- UZL_LM_RUN_AHEAD=1"; UZL_LM_RUN_AHEAD=1 python tests/diag/lm_passes.py 1000:5000 100:300 10000:50000 3000:3300 2>&1 | grep lm_loop
echo "--- host loop"; LOOPS=1 python tests/diag/lm_passes.py 1000:5000 100:300 10000:50000 3000:3300 | grep lm_loop
