"""The session set-up on the device (mavmap_amd/csrc/device_setup.hip): the stable radix sort it is built on, and the
equality of the device path with the host path (same point order definition, same observation orders -> the same bits)."""
import numpy as np
import pytest

from mavmap_amd import _abi as A
from mavmap_amd import synth
from tests.conftest import global_opts

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,key_bytes", [(1, 1), (63, 1), (2048, 2), (2049, 2), (100_003, 3), (1_000_000, 4)])
def test_device_radix_sort_is_numpys_stable_argsort(mavba, n, key_bytes):
    rng = np.random.default_rng(n)
    hi = 1 << (8 * key_bytes)
    keys = rng.integers(0, min(hi, 1 << 32), size=n, dtype=np.uint64).astype(np.uint32)
    if n > 1000:
        keys[::7] = keys[3]          # long runs of equal keys: stability matters
    got = mavba.radix_sort_order(keys, key_bytes)
    want = np.argsort(keys, kind="stable")
    assert np.array_equal(got, want)


@pytest.mark.parametrize("kind", ["c3_small", "long_tracks_priors", "duplicate_images"])
def test_device_setup_equals_host_setup_bit_for_bit(mavba, monkeypatch, kind):
    """MAVBA_SETUP=device | host force one implementation of the ordering block: internal point order, point-major and
    image-major observation orders. Everything downstream (clusters, elimination order, every sum) follows from them, so
    the two solves must agree to the last bit."""
    if kind == "c3_small":
        p = synth.make_config("C3", scale=0.05, seed=31)
    elif kind == "long_tracks_priors":
        p = synth.make_scene(num_images=60, num_points=4000, track_len=5, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=33,
                             rot_priors=True, long_track_frac=0.03, long_track_len=25, spacing=6.0)
    else:
        p = synth.make_scene(num_images=14, num_points=1500, track_len=4, models=[A.MODEL_OPENCV], seed=35)
        # an image seen twice by the same point, points nobody sees, a constant point
        p.obs_image[5] = p.obs_image[4] if p.obs_point[5] == p.obs_point[4] else p.obs_image[5]
        p.points = np.vstack([p.points, np.zeros((3, 3))])
        p.point_const = np.concatenate([p.point_const, np.zeros(3, np.uint8)])
        p.point_const[7] = 1
    out = {}
    for mode in ("host", "device"):
        monkeypatch.setenv("MAVBA_SETUP", mode)
        q = p.copy()
        e = np.full(q.num_points, np.nan)
        with mavba.Session(q, global_opts()) as s:
            info = s.info()
            res = s.solve()
            x = s.get_params()
            cost, r, Jc, Jp, Jk = s.eval_jacobian()   # (exercises the lazily downloaded observation permutation)
            perr = s.point_errors()
        out[mode] = (info, res, x, (cost, r, Jc), perr)
    ih, rh, xh, jh, eh = out["host"]
    idv, rd, xd, jd, ed = out["device"]
    for k in ("num_clusters", "clustered_points", "cluster_partials", "chain_steps", "envelope_tiles", "schur_blocks", "intr_entries"):
        assert ih[k] == idv[k], k
    for k in ("termination", "num_successful_steps", "num_unsuccessful_steps", "final_cost", "initial_cost", "num_residuals",
              "num_parameters_reduced"):
        assert rh[k] == rd[k], k
    for a, b in zip(xh, xd):
        assert np.array_equal(a, b)
    assert jh[0] == jd[0] and np.array_equal(jh[1], jd[1]) and np.array_equal(jh[2], jd[2])
    assert np.array_equal(eh, ed, equal_nan=True)


def test_device_setup_reports_a_bad_observation_index(mavba, monkeypatch):
    monkeypatch.setenv("MAVBA_SETUP", "device")
    p = synth.make_config("C3", scale=0.02, seed=37)
    p.obs_point[11] = p.num_points + 5
    with pytest.raises(mavba.MavbaError) as e:
        mavba.Session(p, global_opts())
    assert e.value.code == A.ERR_BAD_INDEX


@pytest.mark.parametrize("kind", ["window", "window_long_tracks"])
def test_window_setup_shortcuts_change_nothing(mavba, monkeypatch, kind):
    """Round 6, the set-up of a local window (host path): the point order sorted on packed 64-bit keys (at most 31 images) and
    the small uploads / clears collected into one copy + one scatter kernel (upload_batch_begin, host_util.hip). Switched
    off - MAVBA_ORDER_GENERAL=1, MAVBA_UPLOAD_BATCH=0 - the session must be the same to the last bit: same point order (every
    sum follows from it), same tables on the device."""
    if kind == "window":
        p = synth.make_scene(num_images=10, num_points=2500, track_len=4, models=[A.MODEL_OPENCV], seed=3, refine_camera_params=False)
        p.pose_const[:2] = A.CONST_POSE
        p.intr_const[:] = 1
    else:
        # tracks through all 12 images (more than the key's eight), an image seen twice, a constant point, free intrinsics
        p = synth.make_scene(num_images=12, num_points=900, track_len=6, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV], seed=41,
                             long_track_frac=0.2, long_track_len=12)
        p.obs_image[5] = p.obs_image[4] if p.obs_point[5] == p.obs_point[4] else p.obs_image[5]
        p.point_const[11] = 1
    out = {}
    for mode in ("shortcuts", "plain"):
        if mode == "plain":
            monkeypatch.setenv("MAVBA_ORDER_GENERAL", "1")
            monkeypatch.setenv("MAVBA_UPLOAD_BATCH", "0")
        q = p.copy()
        with mavba.Session(q, global_opts()) as s:
            info = s.info()
            res = s.solve()
            x = s.get_params()
            perr = s.point_errors()
        out[mode] = (info, res, x, perr)
    ia, ra, xa, ea = out["shortcuts"]
    ib, rb, xb, eb = out["plain"]
    for k in ("num_clusters", "clustered_points", "cluster_partials", "schur_blocks", "intr_entries"):
        assert ia[k] == ib[k], k
    for k in ("termination", "num_successful_steps", "num_unsuccessful_steps", "final_cost", "initial_cost", "num_residuals"):
        assert ra[k] == rb[k], k
    for a, b in zip(xa, xb):
        assert np.array_equal(a, b)
    assert np.array_equal(ea, eb, equal_nan=True)


def test_batched_small_uploads_with_a_full_arena_and_destinations_written_twice(mavba):
    """The upload batch of the small-problem set-up (csrc/host_util.hip) outside its usual sizes: an arena that fills up several
    times over (pieces leave early, the batch stays open), buffers cleared after being written and uploaded twice (the earlier
    write must leave first: the pieces of one flush run side by side), odd sizes (the tail bytes of a piece), empty buffers."""
    rng = np.random.default_rng(5)
    sizes = [int(x) for x in rng.integers(1, 40_000, size=60)]
    sizes[7] = 0; sizes[11] = -3; sizes[12] = -70_001; sizes[20] = 131_071; sizes[21] = 1; sizes[22] = 16_385
    assert mavba.debug_upload_batch(sizes, arena_bytes=64 << 10) == 0     # 64 KiB arena: ~15 flushes
    assert mavba.debug_upload_batch(sizes, arena_bytes=0) == 0             # default arena: one flush (+ the overlaps)
