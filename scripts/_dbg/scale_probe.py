"""A scene beyond the BASELINE configs on one GPU (C5 scaled): set-up time, device memory, LM iterations - debug harness.
   python scripts/_dbg/scale_probe.py 2.5     (5 000 images / 2.5 M points / ~26 M observations)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
import numpy as np, torch
import mavmap_amd
from mavmap_amd import synth
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
t = time.perf_counter(); p = synth.make_config("C5", scale=scale); print("scene: %d images %d points %d obs, generated in %.1f s" % (p.num_images, p.num_points, p.num_obs, time.perf_counter() - t), flush=True)
free0 = torch.cuda.mem_get_info()[0]
opts = dict(max_num_iterations=8, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8)
t = time.perf_counter()
with mavmap_amd.Session(p, opts) as s:
    print("session create %.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
    info = s.info()
    print({k: info[k] for k in ("matrix_dim", "reduced_dim", "nd_parts", "chain_steps", "envelope_tiles", "num_clusters", "schur_terms") if k in info})
    t = time.perf_counter(); res = s.solve(); dt = time.perf_counter() - t
    it = res["num_successful_steps"] + res["num_unsuccessful_steps"]
    print("solve: %d iterations in %.1f ms (%.2f ms / iteration), cost %.6g -> %.6g, termination %s" % (it, 1e3 * dt, 1e3 * dt / max(it, 1), res["initial_cost"], res["final_cost"], res["termination"]))
    print("device memory in use by the session: %.2f GB" % ((free0 - torch.cuda.mem_get_info()[0]) / 1e9))
    nb = info["matrix_dim"] // 64
    print("reduced system: tile store 2 x %.2f GB (a dense (n + 64) n array would be 2 x %.2f GB)" % (info["envelope_tiles"] * 32768 / 1e9 if "envelope_tiles" in info else float("nan"), (info["matrix_dim"] + 64) * info["matrix_dim"] * 8 / 1e9))
