import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
import numpy as np
from mavmap_amd import synth, api, _abi as A
from tests import oracle_lib as O
rng = np.random.default_rng(0)
for n in (100, 1000):
    X = rng.uniform(-5, 5, (n, 3)) + np.array([0, 0, 20.0])
    rvec = np.array([0.05, -0.03, 0.02]); tvec = np.array([0.3, -0.2, 0.5])
    intr = np.zeros(9); intr[:4] = [600, 600, 376, 240]
    R = synth.rodrigues(rvec[None])[0]
    Xc = X @ R.T + tvec
    uv = np.stack([600 * Xc[:, 0] / Xc[:, 2] + 376, 600 * Xc[:, 1] / Xc[:, 2] + 240], 1) + rng.normal(0, 0.5, (n, 2))
    r0 = rvec + 0.02; t0 = tvec + 0.3
    cp = np.array([600, 600, 376, 240, A.MODEL_PINHOLE], float)
    api.pose_refinement(r0.copy(), t0.copy(), cp, uv, X)
    ts = []
    for _ in range(5):
        t = time.time(); out = api.pose_refinement(r0.copy(), t0.copy(), cp, uv, X); ts.append(time.time() - t)
    print(n, "points: gpu pose_refinement %.2f ms" % (1e3 * min(ts)), {k: out[1][k] for k in ("num_successful_steps", "num_unsuccessful_steps", "setup_seconds", "solve_seconds")})
    # batches: N inlier sets in ONE launch (mavba_pose_refine_batch)
    for count in (16, 256):
        items = [dict(rvec=r0.copy(), tvec=t0.copy(), camera_params=cp, points2D=uv, points3D=X,
                      inlier_mask=(rng.random(n) > 0.2).astype(np.uint8)) for _ in range(count)]
        api.pose_refinement_batch([dict(it, rvec=it["rvec"].copy(), tvec=it["tvec"].copy()) for it in items])
        t = time.time(); out = api.pose_refinement_batch(items); dt = time.time() - t
        print(n, "points: batch of %d in %.2f ms = %.1f us per refinement" % (count, 1e3 * dt, 1e6 * dt / count))
    os.environ["MAVBA_POSE_REFINE_SESSION"] = "1"
    t = time.time(); out = api.pose_refinement(r0.copy(), t0.copy(), cp, uv, X); dt = time.time() - t
    print(n, "points: general session path %.2f ms" % (1e3 * dt))
    del os.environ["MAVBA_POSE_REFINE_SESSION"]
