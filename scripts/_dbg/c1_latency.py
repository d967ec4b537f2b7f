"""Latency of small local-BA solves (C1 windows): wall time per mavba_solve call incl. set-up, vs the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
import numpy as np
from mavmap_amd import synth, api
from tests import oracle_lib as O
full = synth.make_config("C1", 1.0)
wins = [synth.local_ba_window(full, s, 8) for s in range(0, 3)]
opts = dict(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10)
api.bundle_adjustment(wins[0].copy(), opts)  # warm-up (context, module load)
for w in wins:
    t = time.time(); _, r = api.bundle_adjustment(w.copy(), opts); dt = time.time() - t
    its = r["num_successful_steps"] + r["num_unsuccessful_steps"]
    O.set_threads(8)
    t = time.time(); ro, _ = O.solve(w.copy(), O.options(**opts), jac_mode=1); dto = time.time() - t
    print("window: %d img %d pts %d obs | gpu %.1f ms total (setup %.1f ms, solve %.1f ms, %d it, %.3f ms/it) | cpu oracle 8 thr %.1f ms (%d it)" % (
        w.num_images, w.num_points, w.num_obs, dt * 1e3, r["setup_seconds"] * 1e3, r["solve_seconds"] * 1e3, its, r["solve_seconds"] * 1e3 / max(its, 1), dto * 1e3,
        ro["num_successful_steps"] + ro["num_unsuccessful_steps"]))
