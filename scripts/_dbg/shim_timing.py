"""Host time of the drop-in shim's FeatureManager walk at a bench configuration, against the recording mock
(no GPU needed), or - second argument `real` - the whole drop-in call on the GPU: walk + mavba_solve + write-back into the maps.
Usage: shim_timing.py [C2|C3] [real]"""
import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["MAVBA_SETUP_TIMING"] = "1"
import numpy as np
from mavmap_amd import synth
from tests import test_shim as T
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
p = synth.make_config(cfg)
L = T._build(real=len(sys.argv) > 2 and sys.argv[2] == "real")
L.shim_last_ba_seconds.restype = __import__("ctypes").c_double
fm = T.Scene(p)
free = list(range(2, p.num_images)); fixed = [0]; fixed_x = [1]
for rep in range(3):
    t = time.perf_counter()
    rc, *_ = T.run(L, fm, free, fixed, fixed_x, refine_camera_params=1)
    dt = time.perf_counter() - t
    print("rc %d: python call %.1f ms, bundle_adjustment() inside the driver %.1f ms" % (rc, 1e3 * dt, 1e3 * L.shim_last_ba_seconds()))
