export UZL_LIB=uzliti_slam_amd/libuzl_mi355x_diag.so LOOPS=0 REPS=3 UZL_PHASES=1 UZL_PHASES_EACH=1
echo "--- graphs"; python tests/diag/lm_passes.py 1000:5000 2>&1 | tail -30 | head -8
echo "--- eager"; UZL_NO_GRAPH=1 python tests/diag/lm_passes.py 1000:5000 2>&1 | tail -30 | head -8
