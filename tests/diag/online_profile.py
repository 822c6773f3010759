#!/usr/bin/env python3
"""Diagnostic: cProfile of the online run's driver (uzliti_slam_amd/online.py, BASELINE config 5): where the host side of the loop
spends its time beside the library calls.   python tests/diag/online_profile.py"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uzliti_slam_amd import online, synth   # noqa: E402

run = synth.make_online_run(20000, 4096, n_kp=1000)
o = online.OnlineSlam(run, match_batch=512)
o.upload_frames()
o.run_all()                                  # warm
o.close()
o = online.OnlineSlam(run, match_batch=512)
o.upload_frames()
pr = cProfile.Profile()
pr.enable()
wall = o.run_all()
pr.disable()
print("wall %.3f s (under cProfile)" % wall)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
