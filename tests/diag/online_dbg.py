import os, sys
sys.path.insert(0, os.getcwd())
from uzliti_slam_amd import online, synth
run = synth.make_online_run(8000, 1600, n_kp=300)
o = online.OnlineSlam(run, match_batch=512, log=lambda m: print(m, file=sys.stderr), match_cfg=dict(ransac_iteration=100))
o.upload_frames()
o.run_all()
