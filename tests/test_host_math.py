"""The kernel bodies (mavmap_amd/csrc/ba_math.h) compiled for the HOST by g++ and checked against
the oracle — catches maths errors without a GPU. (The GPU tests check the real kernels.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mavmap_amd import _abi as A

HERE = os.path.dirname(os.path.abspath(__file__))
dp = C.POINTER(C.c_double)


def d(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def host():
    src = os.path.join(HERE, "host", "ba_math_host.cpp")
    so = os.path.join(HERE, "host", "_ba_math_host.so")
    hdr = os.path.join(HERE, "..", "mavmap_amd", "csrc", "ba_math.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return C.CDLL(so)


def _case(rng, model, scale):
    from scipy.spatial.transform import Rotation
    K = A.MODEL_NUM_PARAMS[model]
    pose = np.concatenate([rng.normal(0, 1, 3) * scale, rng.normal(0, 1, 3)])
    Xc = np.array([rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(3, 9)])
    X = np.ascontiguousarray(Rotation.from_rotvec(pose[:3]).as_matrix().T @ (Xc - pose[3:]))
    cam = np.zeros(9)
    cam[:4] = [600 + rng.normal(), 610 + rng.normal(), 376, 240]
    if K >= 8:
        cam[4:8] = np.array([-0.1, 0.02, 1e-3, -1e-3]) * (1 + 0.1 * rng.normal(size=4))
    if K == 9:
        cam[8] = rng.uniform(0, 1)
    return pose, X, cam, rng.normal(300, 50, 2)


def test_kernel_projection_jacobian_matches_reference_template_derivatives(host):
    """The KERNEL maths (csrc/ba_math.h, compiled for the host) against the golden derivatives of the reference's own
    world2image<T> templates (tests/golden/world2image_jet_kat.json): no oracle in between."""
    import json
    jet = json.load(open(os.path.join(HERE, "golden", "world2image_jet_kat.json")))
    pose, uv0 = np.zeros(6), np.zeros(2)
    for c in jet["vectors"]:
        K = A.MODEL_NUM_PARAMS[c["code"]]
        cam = np.zeros(9); cam[:K] = c["params"]
        X = np.array(c["Xc"])
        r, Jc, Jp, Jk = np.zeros(2), np.zeros((2, 6)), np.zeros((2, 3)), np.zeros((2, 9))
        host.hm_obs_jacobian(c["code"], d(pose), d(cam), d(X), d(uv0), d(r), d(Jc), d(Jp), d(Jk))
        assert np.abs(r - np.array(c["uv"])).max() <= 1e-12 * np.abs(c["uv"]).max()
        ref_p, ref_k = np.array([c["du"][:3], c["dv"][:3]]), np.array([c["du"][3:], c["dv"][3:]])
        assert np.abs(Jp - ref_p).max() <= 1e-11 * np.abs(ref_p).max(), c["model"]
        assert np.abs(Jk[:, :K] - ref_k).max() <= 1e-11 * np.abs(ref_k).max(), c["model"]
        assert not Jk[:, K:].any()


@pytest.mark.parametrize("model", [1, 2, 3])
def test_obs_jacobian_matches_oracle_jets(host, oracle, model):
    rng = np.random.default_rng(model)
    for it in range(1500):
        scale = [1e-3, 0.3, 1.5, 3.1][it % 4]
        pose, X, cam, uv = _case(rng, model, scale)
        if it % 41 == 0:
            pose[:3] = 0.0
        r0, Jc0, Jp0, Jk0 = oracle.obs_jacobian(0, model, pose, X, cam, uv)
        r, Jc, Jp, Jk = np.zeros(2), np.zeros((2, 6)), np.zeros((2, 3)), np.zeros((2, 9))
        host.hm_obs_jacobian(model, d(pose), d(cam), d(X), d(uv), d(r), d(Jc), d(Jp), d(Jk))
        for a, b in ((r, r0), (Jc, Jc0), (Jp, Jp0), (Jk, Jk0)):
            assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), (model, it)
        r2 = np.zeros(2)
        host.hm_obs_residual(model, d(pose), d(cam), d(X), d(uv), d(r2))
        assert np.abs(r2 - r0).max() <= 1e-11 * max(1.0, np.abs(r0).max())


@pytest.mark.parametrize("model", [1, 2, 3])
def test_backsub_term_by_directional_derivatives_equals_the_contracted_jacobian(host, oracle, model):
    """k_backsub_points_packed needs t = Jp^T (Jc dc + Jk dk) per observation; round 5 evaluates it as directional derivatives
    (obs_backsub_term: A (dt + (Jl dw) x Xr), R^T A^T tau) instead of building the Jacobian and contracting it. Against the
    oracle's Jets, for every model, over small and large rotations."""
    rng = np.random.default_rng(100 + model)
    for it in range(600):
        scale = [1e-3, 0.3, 1.5, 3.1][it % 4]
        pose, X, cam, uv = _case(rng, model, scale)
        if it % 37 == 0:
            pose[:3] = 0.0
        dc = rng.normal(0, 1, 6) * np.array([1e-2, 1e-2, 1e-2, 0.1, 0.1, 0.1])
        dk = np.zeros(9)
        dk[:A.MODEL_NUM_PARAMS[model]] = rng.normal(0, 1, A.MODEL_NUM_PARAMS[model]) * 1e-2
        r0, Jc0, Jp0, Jk0 = oracle.obs_jacobian(0, model, pose, X, cam, uv)
        t0 = Jp0.T @ (Jc0 @ dc + Jk0 @ dk)
        r, t = np.zeros(2), np.zeros(3)
        host.hm_obs_backsub_term(model, d(pose), d(cam), d(X), d(uv), d(dc), d(dk), d(r), d(t))
        assert np.abs(r - r0).max() <= 1e-11 * max(1.0, np.abs(r0).max())
        assert np.abs(t - t0).max() <= 1e-10 * max(1.0, np.abs(t0).max()), (model, it, t, t0)


def test_tiny_rotation_is_more_accurate_than_ceres_formula(host, oracle):
    """For 0 < |rvec| ~ 1e-9 the Rodrigues form the reference differentiates through (ceres <= 1.8)
    loses ~7 digits in d/d rvec; the series used on the device does not. Both must agree with the
    exact small-angle limit -[X]x to their respective accuracies."""
    rng = np.random.default_rng(9)
    pose, X, cam, uv = _case(rng, 1, 1e-9)
    r0, Jc0, *_ = oracle.obs_jacobian(0, 1, pose, X, cam, uv)
    r, Jc, Jp, Jk = np.zeros(2), np.zeros((2, 6)), np.zeros((2, 3)), np.zeros((2, 9))
    host.hm_obs_jacobian(1, d(pose), d(cam), d(X), d(uv), d(r), d(Jc), d(Jp), d(Jk))
    pose0 = pose.copy(); pose0[:3] = 0
    rz, Jcz, *_ = oracle.obs_jacobian(0, 1, pose0, X, cam, uv)  # Taylor branch: exact limit
    assert np.abs(Jc - Jcz).max() < 1e-6 * np.abs(Jcz).max()
    assert np.abs(Jc0 - Jcz).max() < 1e-4 * np.abs(Jcz).max()


def test_rotation_prior_matches_oracle(host, oracle):
    rng = np.random.default_rng(4)
    for it in range(1000):
        sc = [1e-3, 0.5, 2.0, 3.0][it % 4]
        w = rng.normal(0, 1, 3) * sc
        w0 = w + rng.normal(0, 0.02, 3)
        r0, j0 = oracle.rot_prior(w, w0, 1.7)
        res, j = C.c_double(), np.zeros(3)
        host.hm_rot_prior(d(w), d(w0), C.c_double(1.7), C.byref(res), d(j))
        assert abs(res.value - r0) < 1e-11 * max(1, abs(r0))
        assert np.abs(j - j0).max() < 1e-8 * max(1, np.abs(j0).max())
    R = np.zeros(9)
    w = np.array([0.3, -0.2, 0.9])
    host.hm_rot_matrix(d(w), d(R))
    assert np.abs(R.reshape(3, 3).T - oracle.rotation_matrix(w)).max() < 1e-14


def test_chol3_and_cauchy(host):
    rng = np.random.default_rng(5)
    for _ in range(200):
        B = rng.normal(size=(3, 3))
        M = B @ B.T + 0.1 * np.eye(3)
        Cs = np.array([M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2]])
        Gi = np.zeros(6)
        assert host.hm_chol3_inv(d(Cs), d(Gi)) == 1
        G = np.array([[Gi[0], 0, 0], [Gi[1], Gi[2], 0], [Gi[3], Gi[4], Gi[5]]])
        assert np.abs(G.T @ G - np.linalg.inv(M)).max() < 1e-10 * np.abs(np.linalg.inv(M)).max()
    bad = np.array([1.0, 2, 0, 1, 0, 1])
    assert host.hm_chol3_inv(d(bad), d(np.zeros(6))) == 0
    for a in (1.0, 3.0):
        for s in (0.0, 0.5, 40.0):
            w, hr = C.c_double(), C.c_double()
            host.hm_cauchy(C.c_double(s), C.c_double(a), C.byref(w), C.byref(hr))
            assert abs(hr.value - 0.5 * a * a * np.log1p(s / a / a)) < 1e-14 * max(1, hr.value)
            assert abs(w.value - 1 / np.sqrt(1 + s / a / a)) < 1e-15


def test_problem_replay_file_round_trip(tmp_path):
    """BAProblem.save / load (the layout the shim writes with MAVBA_DUMP_DIR) is lossless."""
    from mavmap_amd import synth
    from mavmap_amd.problem import BAProblem
    p = synth.make_scene(num_images=6, num_points=80, track_len=3, models=[1, 3], seed=3, rot_priors=True)
    p.point_const[::9] = 1
    path = str(tmp_path / "p.bin")
    p.save(path, dict(max_num_iterations=33, loss_scale_factor=2.5))
    q, o = BAProblem.load(path)
    for name in ("poses", "pose_const", "image_camera", "intrinsics", "camera_model", "intr_const", "points", "point_const",
                 "obs_uv", "obs_image", "obs_point", "rot_prior_image", "rot_prior_rvec"):
        a, b = getattr(p, name), getattr(q, name)
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), name
    assert q.rot_prior_weight == p.rot_prior_weight and o["max_num_iterations"] == 33 and o["loss_scale_factor"] == 2.5
    with open(path, "r+b") as f:
        f.write(b"XX")
    with pytest.raises(ValueError):
        BAProblem.load(path)


def test_log_series_of_the_device_build(host):
    """log_obs (ba_math.h, round 6): the kernels' short log for the Cauchy cost - the same series evaluated on the host -
    against the library's over the range the loss feeds it (1 + s / b, s >= 0)."""
    host.hm_log_series.restype = C.c_double
    host.hm_log_series.argtypes = [C.c_double]
    rng = np.random.default_rng(5)
    xs = np.concatenate([1.0 + 10.0 ** rng.uniform(-16, 0, 4000), 10.0 ** rng.uniform(0, 12, 4000), [1.0, 2.0, 0.5 ** 0.5 * 2, 2.0 ** 0.5, 4.0]])
    got = np.array([host.hm_log_series(float(x)) for x in xs])
    ref = np.log(xs)
    assert got[-5] == 0.0
    err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
    assert err[ref > 0].max() <= 4.5e-16, err.max()
