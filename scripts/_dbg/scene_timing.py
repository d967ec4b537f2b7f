"""Host time of mavba_scene_flatten for a global BA (no GPU needed). Usage: scene_timing.py [C2|C3]"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from mavmap_amd import api, synth
from mavmap_amd import _abi as A
import ctypes as C
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
p = synth.make_config(cfg)
L = api.load()
sc = api.Scene()
t0 = time.time()
for c in range(p.num_cameras):
    sc.set_camera(c + 1, int(p.camera_model[c]), p.intrinsics[c])
for i in range(p.num_images):
    sc.set_image(i + 1, int(p.image_camera[i]) + 1, p.poses[i, :3], p.poses[i, 3:])
dp = C.POINTER(C.c_double)
pts = np.ascontiguousarray(p.points)
for k in range(p.num_points):
    L.mavba_scene_set_point3d(sc._h, k + 1, C.cast(pts.ctypes.data + 24 * k, dp))
order = np.argsort(p.obs_image, kind="stable")
uv = np.ascontiguousarray(p.obs_uv[order]); oi = p.obs_image[order]; op = p.obs_point[order]
for o in range(p.num_obs):
    L.mavba_scene_add_point2d(sc._h, int(oi[o]) + 1, o + 1, C.cast(uv.ctypes.data + 16 * o, dp))
    L.mavba_scene_link(sc._h, o + 1, int(op[o]) + 1)
print("fill %.1f s" % (time.time() - t0))
free = np.arange(3, p.num_images + 1); fixed = np.array([1]); fx = np.array([2])
args, hold = sc._lists(free, fixed, fx, (), None)
so = sc._scene_options(refine_camera_params=True)
P = A.CProblem()
for rep in range(5):
    t0 = time.perf_counter()
    rc = L.mavba_scene_flatten(sc._h, *args, C.byref(so), C.byref(P), None, None, None)
    t1 = time.perf_counter()
    print("flatten %.2f ms rc=%d NI=%d NP=%d NO=%d" % ((t1 - t0) * 1e3, rc, P.num_images, P.num_points, P.num_obs))

# host route vs device-resident route of mavba_scene_bundle_adjust (same scene, fresh copy of the parameters each time)
for route in ("host", "device"):
    os.environ["MAVBA_SCENE"] = route
    for rep in range(3):
        t = time.time()
        cost, res = sc.bundle_adjustment(free, fixed, fx, dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10), refine_camera_params=1)
        dt = time.time() - t
        print("route %-6s call %d: %.1f ms end to end (setup %.1f ms, solve %.1f ms, %d iterations, cost %.6f)" % (
            route, rep, 1e3 * dt, 1e3 * res["setup_seconds"], 1e3 * res["solve_seconds"], res["num_successful_steps"] + res["num_unsuccessful_steps"], cost), file=sys.stderr)
