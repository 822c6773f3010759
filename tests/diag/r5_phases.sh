#!/bin/bash
# GPU time between the segment boundaries of the passes of one solve (diagnostic build, UZL_PHASES=1)
export UZL_LIB=$PWD/uzliti_slam_amd/libuzl_mi355x_diag.so UZL_PHASES=1
python3 tests/diag/pass_log.py ${1:-1000} ${2:-5000} 1 2>&1 | grep -E "segments|diag\]"
