import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["MAVBA_SETUP_TIMING"] = "1"
import numpy as np, mavmap_amd
from mavmap_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
p = synth.make_config(cfg)
opts = dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10)
for rep in range(3):
    t = time.time()
    s = mavmap_amd.Session(p, opts)
    print("create %.1f ms" % (1e3 * (time.time() - t)), file=sys.stderr)
    s.close()
for rep in range(3):
    q = p.copy()
    print("---- mavba_solve call", rep, file=sys.stderr)
    t = time.time(); cost, res = mavmap_amd.bundle_adjustment(q, opts); dt = time.time() - t
print("mavba_solve end to end %.1f ms: setup %.1f, solve %.1f, iterations %d" % (1e3 * dt, 1e3 * res["setup_seconds"], 1e3 * res["solve_seconds"], res["num_successful_steps"] + res["num_unsuccessful_steps"]), file=sys.stderr)
