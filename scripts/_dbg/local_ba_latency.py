"""Latency of a local-BA sized call (10 images, 2 of them fixed) through mavba_solve, and where a session spends it."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, mavmap_amd
from mavmap_amd import synth
from mavmap_amd import _abi as A
p = synth.make_scene(num_images=10, num_points=2500, track_len=4, models=[A.MODEL_OPENCV], seed=3, refine_camera_params=False)
p.pose_const[:2] = A.CONST_POSE
p.intr_const[:] = 1
opts = dict(max_num_iterations=100, function_tolerance=1e-4, gradient_tolerance=1e-8)
print("images %d points %d obs %d" % (p.num_images, p.num_points, p.num_obs))
ts, su, so = [], [], []
for rep in range(60):
    q = p.copy()
    t = time.perf_counter(); cost, res = mavmap_amd.bundle_adjustment(q, opts); ts.append(time.perf_counter() - t)
    su.append(res["setup_seconds"]); so.append(res["solve_seconds"])
it = res["num_successful_steps"] + res["num_unsuccessful_steps"]
print("mavba_solve: median %.3f ms, min %.3f ms, iterations %d, setup median %.3f ms (min %.3f), solve median %.3f ms" % (1e3 * np.median(ts[5:]), 1e3 * min(ts), it, 1e3 * np.median(su[5:]), 1e3 * min(su), 1e3 * np.median(so[5:])))
# the reference's default for local BA refines the intrinsics as well (mapper.cc:883, local-ba-refine-camera-params = true)
pr = synth.make_scene(num_images=10, num_points=2500, track_len=4, models=[A.MODEL_OPENCV], seed=3, refine_camera_params=True)
pr.pose_const[:2] = A.CONST_POSE
ts, su, so = [], [], []
for rep in range(60):
    q = pr.copy()
    t = time.perf_counter(); cost, res = mavmap_amd.bundle_adjustment(q, opts); ts.append(time.perf_counter() - t)
    su.append(res["setup_seconds"]); so.append(res["solve_seconds"])
it = res["num_successful_steps"] + res["num_unsuccessful_steps"]
print("mavba_solve, free intrinsics: median %.3f ms, min %.3f ms, iterations %d, setup median %.3f ms (min %.3f), solve median %.3f ms (%.1f us / iteration)" % (
    1e3 * np.median(ts[5:]), 1e3 * min(ts), it, 1e3 * np.median(su[5:]), 1e3 * min(su), 1e3 * np.median(so[5:]), 1e6 * np.median(so[5:]) / max(it, 1)))
with mavmap_amd.Session(p, dict(opts, profile_kernels=1)) as s:
    t = time.perf_counter(); s.iterate(1000); dt = time.perf_counter() - t
    print("session solve with event timers %.3f ms" % (1e3 * dt))
    st = s.kernel_stats()
    tot = 0.0
    for k, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"]):
        print("  %-18s n=%4d avg=%8.4f ms total=%7.3f" % (k, v["launches"], v["total_ms"] / max(v["launches"], 1), v["total_ms"])); tot += v["total_ms"]
    print("  sum of kernel times %.3f ms" % tot)
with mavmap_amd.Session(p, opts) as s:
    print({k: v for k, v in s.info().items() if k in ("matrix_dim", "reduced_dim", "nd_parts", "chain_steps", "num_clusters", "clustered_points", "schur_terms")})
