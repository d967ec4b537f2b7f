"""Set-up phases of a local-window call (MAVBA_SETUP_TIMING) - debug harness."""
import os, sys
os.environ["MAVBA_SETUP_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
import mavmap_amd
from mavmap_amd import synth, _abi as A
p = synth.make_scene(num_images=10, num_points=2500, track_len=4, models=[A.MODEL_OPENCV], seed=3, refine_camera_params=False)
p.pose_const[:2] = A.CONST_POSE
p.intr_const[:] = 1
opts = dict(max_num_iterations=100, function_tolerance=1e-4, gradient_tolerance=1e-8)
for rep in range(4):
    print("---- call", rep, file=sys.stderr)
    mavmap_amd.bundle_adjustment(p.copy(), opts)
