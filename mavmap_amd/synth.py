"""Seeded synthetic MAV-style scenes for the BASELINE.json configs.

Cameras fly a serpentine nadir grid at altitude 50; points lie on a gently
rolling ground; every point is observed by its L nearest cameras that actually
see it. World frame = frame of image 0 (the reference sets the first image's
pose to exactly zero, reference src/sfm/sequential_mapper.cc:270,332), image 0
is FIXED and image 1 FIXED_X as in the reference's global BA
(reference src/sfm/sequential_mapper.cc:1095-1097).

This is a build-owned generator (the reference has no synthetic data and no BA
test); numpy only.
"""
import numpy as np

from . import _abi as A
from .problem import BAProblem

PINHOLE_PARAMS = np.array([600.0, 600.0, 376.0, 240.0, 0, 0, 0, 0, 0])
OPENCV_PARAMS = np.array([600.0, 600.0, 376.0, 240.0, -0.1, 0.02, 1e-3, -1e-3, 0])
CATA_PARAMS = np.array([600.0, 600.0, 376.0, 240.0, -0.1, 0.02, 1e-3, -1e-3, 0.3])
IMAGE_W, IMAGE_H = 752.0, 480.0
ALTITUDE = 50.0

CONFIGS = {
    # name: images, points, track length, camera models (alternating by image), extras
    "C1": dict(num_images=10, num_points=2000, track_len=4, models=[A.MODEL_PINHOLE]),
    "C2": dict(num_images=100, num_points=30000, track_len=10, models=[A.MODEL_PINHOLE]),
    "C3": dict(num_images=500, num_points=200000, track_len=10,
               models=[A.MODEL_PINHOLE, A.MODEL_OPENCV]),
    "C5": dict(num_images=2000, num_points=1000000, track_len=10,
               models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], rot_priors=True, long_track_frac=0.05,
               long_track_len=30),
}
CONFIG_SEED = {"C1": 1001, "C2": 1002, "C3": 1003, "C4": 1003, "C5": 1005}


def rodrigues(rvec):
    """(N,3) angle-axis -> (N,3,3) rotation matrices."""
    rvec = np.atleast_2d(rvec)
    th = np.linalg.norm(rvec, axis=1)
    K = np.zeros((len(rvec), 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -rvec[:, 2], rvec[:, 1]
    K[:, 1, 0], K[:, 1, 2] = rvec[:, 2], -rvec[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -rvec[:, 1], rvec[:, 0]
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0, np.sin(ths) / ths)
    b = np.where(small, 0.5, (1 - np.cos(ths)) / ths ** 2)
    return np.eye(3)[None] + a[:, None, None] * K + b[:, None, None] * (K @ K)


def log_so3(R):
    """(3,3) rotation matrix -> angle-axis (robust near pi)."""
    c = np.clip((np.trace(R) - 1) / 2, -1, 1)
    th = np.arccos(c)
    if th < 1e-10:
        return np.zeros(3)
    if np.pi - th < 1e-6:
        # near pi: axis from the largest diagonal element of (R + I)/2
        M = (R + np.eye(3)) / 2
        k = int(np.argmax(np.diag(M)))
        ax = M[:, k] / np.sqrt(M[k, k])
        w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        if np.dot(w, ax) < 0:
            ax = -ax
        return th * ax / np.linalg.norm(ax)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2 * np.sin(th))
    return th * w


def project(model, params, Xc):
    """World2image on camera-frame points (N,3) -> (N,2); numpy restatement used by the generator."""
    x, y, z = Xc[:, 0], Xc[:, 1], Xc[:, 2]
    if model == A.MODEL_CATA:
        z = z + params[8] * np.sqrt(x * x + y * y + z * z)
    u, v = x / z, y / z
    if model != A.MODEL_PINHOLE:
        k1, k2, p1, p2 = params[4:8]
        u2, v2, uv = u * u, v * v, u * v
        r2 = u2 + v2
        rad = k1 * r2 + k2 * r2 * r2
        du = u * rad + 2 * p1 * uv + p2 * (r2 + 2 * u2)
        dv = v * rad + 2 * p2 * uv + p1 * (r2 + 2 * v2)
        u, v = u + du, v + dv
    return np.stack([params[0] * u + params[2], params[1] * v + params[3]], axis=1)


def _camera_grid(num_images, spacing):
    cols = int(np.ceil(np.sqrt(num_images * 1.5)))
    rows = int(np.ceil(num_images / cols))
    centres, yaw = [], []
    for r in range(rows):
        cs = range(cols) if r % 2 == 0 else range(cols - 1, -1, -1)
        for c in cs:
            if len(centres) == num_images:
                break
            centres.append((c * spacing, r * spacing * 1.25, ALTITUDE))
            yaw.append(0.0 if r % 2 == 0 else np.pi)
    return np.array(centres), np.array(yaw)


def make_scene(num_images, num_points, track_len, models, seed, rot_priors=False,
               long_track_frac=0.0, long_track_len=0, noise_px=0.5, outlier_frac=0.01,
               perturb=True, spacing=11.0, refine_camera_params=True, image_camera=None):
    """Build one global-BA problem. Returns a BAProblem with `truth` filled."""
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(seed)
    centres, yaw = _camera_grid(num_images, spacing)
    centres = centres + rng.normal(0, 0.3, centres.shape)
    NI = num_images
    NC = len(models)
    if image_camera is None:
        image_camera = np.arange(NI) % NC  # round-robin over the cameras
    image_camera = np.asarray(image_camera, np.int32)
    assert image_camera.shape == (NI,) and image_camera.min() >= 0 and image_camera.max() < NC
    camera_model = np.array(models, np.int32)
    intr_true = np.stack([{A.MODEL_PINHOLE: PINHOLE_PARAMS, A.MODEL_OPENCV: OPENCV_PARAMS,
                           A.MODEL_CATA: CATA_PARAMS}[m] for m in models]).copy()

    # nadir attitude: camera x = world x, y = -world y, z = down; yaw flips on odd rows; +-5 deg jitter
    R_wc = np.zeros((NI, 3, 3))
    base = np.diag([1.0, -1.0, -1.0])
    for i in range(NI):
        cy, sy = np.cos(yaw[i]), np.sin(yaw[i])
        Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
        jit = rodrigues(rng.uniform(-1, 1, 3) * np.deg2rad(5.0) / np.sqrt(3))[0]
        R_wc[i] = jit @ base @ Rz.T
    t_wc = -np.einsum("nij,nj->ni", R_wc, centres)

    # ground points under the flight area, relief sigma = 2
    lo = centres[:, :2].min(0) + 5.0
    hi = centres[:, :2].max(0) - 5.0
    if np.any(hi <= lo):
        lo, hi = centres[:, :2].min(0) - 5.0, centres[:, :2].max(0) + 5.0
    tree = cKDTree(centres[:, :2])
    pts, obs_img_l, obs_pt_l = [], [], []
    track_len = min(track_len, NI)
    long_track_len = min(long_track_len, NI)
    done = 0
    stalled = 0
    kq = min(NI, max(4 * max(track_len, long_track_len), 48))
    while done < num_points:
        before = done
        if stalled > 50:
            raise RuntimeError("scene generator: no point is visible in enough cameras "
                               f"(images={NI}, track_len={track_len}, spacing={spacing})")
        m = min(max(2 * (num_points - done), 1024), 400000)
        xy = rng.uniform(lo, hi, (m, 2))
        z = 2.0 * np.sin(xy[:, 0] / 17.0) * np.cos(xy[:, 1] / 23.0) + rng.normal(0, 0.5, m)
        X = np.column_stack([xy, z])
        is_long = rng.random(m) < long_track_frac if long_track_frac > 0 else np.zeros(m, bool)
        _, nn = tree.query(xy, k=kq)
        nn = nn.reshape(m, -1)
        vis = np.zeros(nn.shape, bool)
        cnt = np.zeros(m, np.int64)
        need = np.where(is_long, max(long_track_len, track_len), track_len)
        rows = np.arange(m)
        for j in range(nn.shape[1]):
            # columns are sorted by distance: rows that already see enough cameras drop out
            rows = rows[cnt[rows] < need[rows]]
            if len(rows) == 0:
                break
            ci = nn[rows, j]
            Xc = np.einsum("nij,nj->ni", R_wc[ci], X[rows]) + t_wc[ci]
            ok = Xc[:, 2] > 1.0
            for c in range(NC):
                sel = ok & (image_camera[ci] == c)
                if not sel.any():
                    continue
                uv = project(camera_model[c], intr_true[c], Xc[sel])
                inside = (uv[:, 0] > 2) & (uv[:, 0] < IMAGE_W - 2) & (uv[:, 1] > 2) & (uv[:, 1] < IMAGE_H - 2)
                hit = rows[np.nonzero(sel)[0][inside]]
                vis[hit, j] = True
                cnt[hit] += 1
        accept = np.nonzero(cnt >= track_len)[0][: num_points - done]
        if len(accept):
            # columns of `nn` are sorted by distance: a stable sort on "not visible" lists the visible
            # cameras of every row first, nearest first
            first_vis = np.argsort(~vis[accept], axis=1, kind="stable")
            normal = ~is_long[accept]
            rows_n = accept[normal]
            cams_n = np.sort(np.take_along_axis(nn[rows_n], first_vis[normal][:, :track_len], axis=1), axis=1)
            ids = done + np.arange(len(accept))
            obs_img_l.append(cams_n.ravel())
            obs_pt_l.append(np.repeat(ids[normal], track_len))
            # long "loop-closure" tracks: as many views as exist (up to long_track_len), spread over
            # the visible set instead of the nearest cameras
            for r, pid in zip(accept[~normal], ids[~normal]):
                cand = nn[r, vis[r]]
                L = min(long_track_len, len(cand))
                chosen = np.sort(cand[np.linspace(0, len(cand) - 1, L).astype(int)]) if L > track_len else np.sort(cand[:track_len])
                obs_img_l.append(chosen)
                obs_pt_l.append(np.full(len(chosen), pid))
            pts.append(X[accept])
            done += len(accept)
        stalled = stalled + 1 if done == before else 0
    X_true = np.concatenate(pts)
    obs_img = np.concatenate(obs_img_l).astype(np.int32)
    obs_pt = np.concatenate(obs_pt_l).astype(np.int32)

    # move the world frame into image 0's frame -> pose 0 == (0, 0) exactly
    R0, t0 = R_wc[0].copy(), t_wc[0].copy()
    X_true = X_true @ R0.T + t0
    R_new = np.einsum("nij,kj->nik", R_wc, R0)          # R_i R0^T
    t_new = t_wc - np.einsum("nij,j->ni", R_new, t0)
    rvec_true = np.array([log_so3(R) for R in R_new])
    rvec_true[0] = 0.0
    t_new[0] = 0.0
    poses_true = np.hstack([rvec_true, t_new])

    # reference residual order: FREE images, FIXED images, FIXED_X images (bundle_adjustment.cc:511-533)
    pose_const = np.zeros(NI, np.uint8)
    pose_const[0] = A.CONST_POSE
    if NI > 1:
        pose_const[1] = A.CONST_TX
    img_rank = np.empty(NI, np.int64)
    order_imgs = [i for i in range(NI) if pose_const[i] == 0] + [0] + ([1] if NI > 1 else [])
    img_rank[order_imgs] = np.arange(NI)
    order = np.lexsort((obs_pt, img_rank[obs_img]))
    obs_img, obs_pt = obs_img[order], obs_pt[order]

    # observations = projection of the truth + noise (+ gross outliers)
    R_all = rodrigues(rvec_true)
    Xc = np.einsum("nij,nj->ni", R_all[obs_img], X_true[obs_pt]) + t_new[obs_img]
    uv = np.zeros((len(obs_img), 2))
    for c in range(NC):
        sel = image_camera[obs_img] == c
        uv[sel] = project(camera_model[c], intr_true[c], Xc[sel])
    uv_clean = uv.copy()
    uv += rng.normal(0, noise_px, uv.shape)
    n_out = int(round(outlier_frac * len(uv)))
    if n_out:
        oi = rng.choice(len(uv), n_out, replace=False)
        uv[oi] += rng.uniform(-50, 50, (n_out, 2))

    poses = poses_true.copy()
    points = X_true.copy()
    intr = intr_true.copy()
    if perturb:
        dr = rng.normal(0, 0.01, (NI, 3))
        dt = rng.normal(0, 0.5, (NI, 3))
        dr[0] = 0; dt[0] = 0
        if NI > 1:
            dt[1, 0] = 0
        poses[:, :3] += dr
        poses[:, 3:] += dt
        points += rng.normal(0, 0.5, points.shape)
        intr[:, 0:2] *= 1 + rng.uniform(-0.01, 0.01, (NC, 2))

    prob = BAProblem(
        poses=poses, pose_const=pose_const, image_camera=image_camera,
        intrinsics=intr, camera_model=camera_model,
        intr_const=np.full(NC, 0 if refine_camera_params else 1, np.uint8),
        points=points, point_const=np.zeros(len(points), np.uint8),
        obs_uv=uv, obs_image=obs_img, obs_point=obs_pt,
    )
    if rot_priors:
        free = np.nonzero(pose_const == 0)[0].astype(np.int32)
        pri = np.zeros((len(free), 3))
        for k, i in enumerate(free):
            Rn = rodrigues(rng.normal(0, np.deg2rad(0.5), 3))[0] @ R_all[i]
            pri[k] = log_so3(Rn)
        prob.rot_prior_image, prob.rot_prior_rvec, prob.rot_prior_weight = free, pri, 1.0
    prob.truth = dict(poses=poses_true, points=X_true, intrinsics=intr_true, uv_clean=uv_clean)
    return prob


def make_config(name, scale=1.0, **overrides):
    """One of the BASELINE.json configs ("C1", "C2", "C3"/"C4", "C5"); `scale` shrinks images and
    points together (parity tests run the same generator at sizes the oracle finishes in seconds)."""
    key = "C3" if name == "C4" else name
    cfg = dict(CONFIGS[key])
    cfg["num_images"] = max(4, int(round(cfg["num_images"] * scale)))
    cfg["num_points"] = max(50, int(round(cfg["num_points"] * scale)))
    cfg.update(overrides)
    seed = cfg.pop("seed", CONFIG_SEED[name])
    return make_scene(seed=seed, **cfg)


def local_ba_window(prob, first, window=8):
    """The reference's local-BA selection: a sliding window of `window` images with states
    [FIXED, FIXED, FREE x (window-2)] (reference src/mapper.cc:864-866, 989-993) cut out of a
    global problem: observations of other images are dropped, then points with fewer than two
    observations inside the window (min_track_len = 2, bundle_adjustment.cc:330)."""
    imgs = np.arange(first, min(first + window, prob.num_images))
    in_win = np.isin(prob.obs_image, imgs)
    cnt = np.bincount(prob.obs_point[in_win], minlength=prob.num_points)
    keep = in_win & (cnt[prob.obs_point] >= 2)
    q = prob.copy()
    q.pose_const = np.zeros(prob.num_images, np.uint8)
    q.pose_const[imgs[:2]] = A.CONST_POSE
    # reference order: FREE images first, then FIXED
    rank = np.full(prob.num_images, 0, np.int64)
    rank[imgs[:2]] = 1
    idx = np.nonzero(keep)[0]
    idx = idx[np.argsort(rank[prob.obs_image[idx]], kind="stable")]
    q.obs_uv = np.ascontiguousarray(prob.obs_uv[idx])
    q.obs_image = np.ascontiguousarray(prob.obs_image[idx])
    q.obs_point = np.ascontiguousarray(prob.obs_point[idx])
    q.rot_prior_image = np.zeros(0, np.int32)
    q.rot_prior_rvec = np.zeros((0, 3))
    return q
