"""The multi-rank code path exercised on ONE GPU: W sessions (one per shard of the 3-D points) run in
W threads of this process and exchange through an in-process all-reduce installed with
mavba_session_set_allreduce — the same hook bench.py backs with RCCL. Checks that sharding
reproduces the single-rank solve (and therefore the oracle)."""
import ctypes as C
import threading

import numpy as np
import pytest

from mavmap_amd import _abi as A
from mavmap_amd import synth
from tests.conftest import assert_params_close, global_opts, rel_err

pytestmark = pytest.mark.gpu


class InProcessAllReduce:
    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.bufs = [None] * world
        self.calls = 0
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMemcpy.restype = C.c_int

    def hook(self, rank):
        def fn(ptr, count, op):
            host = np.empty(count)
            assert self.hip.hipMemcpy(host.ctypes.data, ptr, count * 8, 2) == 0   # device -> host
            self.bufs[rank] = host
            self.bar.wait()
            stack = np.stack(self.bufs)                                            # fixed rank order
            tot = stack.max(0) if op == 1 else stack.sum(0)
            if op == 2:                                                            # sums, last element max
                tot[-1] = stack[:, -1].max()
            self.bar.wait()
            assert self.hip.hipMemcpy(ptr, tot.ctypes.data, count * 8, 1) == 0     # host -> device
            if rank == 0:
                self.calls += 1
        return fn


def solve_sharded(mavba, full, world, opts):
    ar = InProcessAllReduce(world)
    out, errs = [None] * world, []

    def worker(rank):
        try:
            shard, owned = full.shard_by_point(rank, world)
            with mavba.Session(shard, opts) as s:
                s.set_allreduce(ar.hook(rank), rank, world)
                res = s.solve()
                poses, intr, pts = s.get_params()
            out[rank] = (res, poses, intr, pts, owned)
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))
            ar.bar.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not errs, errs
    return out, ar


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_solve_matches_single_rank_and_oracle(mavba, oracle, world):
    full = synth.make_scene(num_images=12, num_points=1500, track_len=4, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV],
                            seed=70 + world, rot_priors=True)
    opts = global_opts()
    single = full.copy()
    _, r1 = mavba.bundle_adjustment(single, opts)
    out, ar = solve_sharded(mavba, full, world, opts)
    assert ar.calls > 3 * (r1["num_successful_steps"] + r1["num_unsuccessful_steps"])
    pts = np.zeros_like(full.points)
    for res, poses, intr, p, owned in out:
        # every rank ends with the same cameras and the same summary
        assert res["termination"] == r1["termination"]
        assert res["num_successful_steps"] == r1["num_successful_steps"]
        assert res["num_unsuccessful_steps"] == r1["num_unsuccessful_steps"]
        assert res["num_residuals"] == r1["num_residuals"] and res["num_parameters_reduced"] == r1["num_parameters_reduced"]
        assert abs(res["final_cost"] - r1["final_cost"]) <= 1e-9 * r1["final_cost"]
        assert np.array_equal(poses, out[0][1]) and np.array_equal(intr, out[0][2])
        assert_params_close(dict(poses=poses, intrinsics=intr), single, tol=1e-8)
        pts[owned] = p
    assert_params_close(dict(points=pts), single, tol=1e-8)
    q = full.copy()
    ro, _ = oracle.solve(q, oracle.options(**opts))
    assert abs(out[0][0]["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
    assert_params_close(dict(poses=out[0][1], intrinsics=out[0][2], points=pts), q)


@pytest.mark.parametrize("world", [2, 3])
def test_deferred_evaluation_with_several_ranks(mavba, monkeypatch, world):
    """The RCCL path keeps the single read-back per iteration: the evaluation at an accepted point is enqueued, all-reduced,
    and read back together with the NEXT candidate's scalars, which are all-reduced after it. The two groups of scalar
    slots must be reduced separately - a collective over all slots sums the (already global) cost and |x|^2 over the ranks
    a second time. MAVBA_DEFER_WITH_HOOK runs that protocol over the in-process hook, so one GPU can check it with
    world > 1: every rank must reproduce the single-rank solve step for step."""
    full = synth.make_config("C3", scale=0.02, seed=17)
    opts = global_opts()
    single = full.copy()
    _, r1 = mavba.bundle_adjustment(single, opts)
    monkeypatch.setenv("MAVBA_DEFER_WITH_HOOK", "1")
    out, _ = solve_sharded(mavba, full, world, opts)
    pts = np.zeros_like(full.points)
    for res, poses, intr, p, owned in out:
        assert res["termination"] == r1["termination"]
        assert res["num_successful_steps"] == r1["num_successful_steps"] and res["num_unsuccessful_steps"] == r1["num_unsuccessful_steps"]
        assert abs(res["initial_cost"] - r1["initial_cost"]) <= 1e-10 * r1["initial_cost"]
        assert abs(res["final_cost"] - r1["final_cost"]) <= 1e-9 * r1["final_cost"]
        pts[owned] = p
    assert_params_close(dict(poses=out[0][1], intrinsics=out[0][2], points=pts), single, tol=1e-8)


def test_rank_without_observations_of_an_image_keeps_it_free(mavba):
    """Sharding by contiguous point ranges leaves some ranks with no observation of some images; the
    `used` flags are max-reduced so those images stay free parameters on every rank."""
    full = synth.make_scene(num_images=16, num_points=800, track_len=3, models=[A.MODEL_PINHOLE], seed=81)
    # order the points by x so that shards are spatially compact
    order = np.argsort(full.points[:, 0])
    inv = np.empty_like(order); inv[order] = np.arange(len(order))
    full.points = np.ascontiguousarray(full.points[order]); full.point_const = np.ascontiguousarray(full.point_const[order])
    full.obs_point = inv[full.obs_point].astype(np.int32)
    s0, _ = full.shard_by_point(0, 2)
    assert len(np.unique(s0.obs_image)) < full.num_images
    opts = global_opts()
    single = full.copy()
    _, r1 = mavba.bundle_adjustment(single, opts)
    out, _ = solve_sharded(mavba, full, 2, opts)
    assert abs(out[0][0]["final_cost"] - r1["final_cost"]) <= 1e-9 * r1["final_cost"]
    assert rel_err(out[0][1], single.poses) < 1e-8


def test_tile_envelope_is_the_union_over_ranks(mavba):
    """The factorisation skips tiles left of the reduced system's envelope. With shards the all-reduced
    matrix has the UNION of the ranks' structures: here only the last rank owns the long tracks that
    couple far-apart images, so a rank using its local envelope would drop their blocks."""
    full = synth.make_scene(num_images=48, num_points=2400, track_len=3, models=[A.MODEL_PINHOLE], seed=91,
                            long_track_frac=0.02, long_track_len=40, spacing=6.0)
    cnt = np.bincount(full.obs_point, minlength=full.num_points)
    order = np.argsort(cnt, kind="stable")  # long tracks last -> all on the last rank
    inv = np.empty_like(order); inv[order] = np.arange(len(order))
    full.points = np.ascontiguousarray(full.points[order]); full.point_const = np.ascontiguousarray(full.point_const[order])
    full.obs_point = inv[full.obs_point].astype(np.int32)
    s0, _ = full.shard_by_point(0, 2)
    with mavba.Session(s0) as a, mavba.Session(full) as b:
        assert a.info()["envelope_tiles"] < b.info()["envelope_tiles"]  # rank 0 alone sees a narrower band
    opts = global_opts()
    single = full.copy()
    _, r1 = mavba.bundle_adjustment(single, opts)
    out, _ = solve_sharded(mavba, full, 2, opts)
    for res, poses, intr, p, owned in out:
        assert res["termination"] == r1["termination"]
        assert abs(res["final_cost"] - r1["final_cost"]) <= 1e-9 * r1["final_cost"]
        assert rel_err(poses, single.poses) < 1e-8


def test_sharded_dissection_order_is_the_same_on_every_rank(mavba):
    """A problem large enough for the nested-dissection order: the image graph is max-reduced, so both ranks
    pick the same parts even though each sees only its own points; only the structure's lower tiles travel."""
    full = synth.make_scene(num_images=130, num_points=5000, track_len=4, models=[A.MODEL_PINHOLE, A.MODEL_OPENCV],
                            seed=5, long_track_frac=0.01, long_track_len=12, spacing=6.0)
    opts = global_opts()
    single = full.copy()
    _, r1 = mavba.bundle_adjustment(single, opts)
    out, _ = solve_sharded(mavba, full, 2, opts)
    pts = np.zeros_like(full.points)
    for res, poses, intr, p, owned in out:
        assert res["termination"] == r1["termination"]
        assert res["num_successful_steps"] == r1["num_successful_steps"]
        assert abs(res["final_cost"] - r1["final_cost"]) <= 1e-9 * r1["final_cost"]
        assert np.array_equal(poses, out[0][1])
        # (different summation order than the single-rank run over a 40-iteration path of a 130-image problem)
        assert rel_err(poses, single.poses) < 1e-7 and rel_err(intr, single.intrinsics) < 1e-7
        pts[owned] = p
    assert rel_err(pts, single.points) < 1e-7


def test_torch_zero_copy_view_of_a_device_pointer(mavba):
    """bench.py's RCCL hook wraps the session's raw device pointer as a torch tensor (CUDA array
    interface). Check that the view aliases the memory (no copy) on this GPU."""
    torch = pytest.importorskip("torch")
    from mavmap_amd.dist import tensor_from_ptr
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    base = torch.arange(16, dtype=torch.float64, device="cuda:0")
    view = tensor_from_ptr(base.data_ptr() + 4 * 8, 8, torch.device("cuda:0"))
    assert view.data_ptr() == base.data_ptr() + 32 and view.dtype == torch.float64
    view.mul_(2.0)
    torch.cuda.synchronize()
    host = np.zeros(16)
    assert hip.hipMemcpy(host.ctypes.data, base.data_ptr(), 16 * 8, 2) == 0
    expect = np.arange(16.0); expect[4:12] *= 2
    assert np.array_equal(host, expect)


def test_native_rccl_exchange_single_rank(mavba, monkeypatch):
    """The native collective (librccl loaded by the library, ncclAllReduce enqueued on the session's stream, deferred
    read-back kept) with the only communicator one GPU allows: one rank. MAVBA_FORCE_EXCHANGE makes that rank run the
    whole multi-rank protocol (structure agreement, packed tiles, scalar reductions); the solve must be the plain one."""
    p = synth.make_config("C3", scale=0.02, seed=3)
    with mavba.Session(p, global_opts()) as s:
        ref = s.solve()
        xref = s.get_params()
    monkeypatch.setenv("MAVBA_FORCE_EXCHANGE", "1")
    uid = mavba.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    with mavba.Session(p, global_opts()) as s:
        s.set_rccl(uid, 0, 1)
        got = s.solve()
        xgot = s.get_params()
        with pytest.raises(mavba.MavbaError):
            s.set_rccl(uid, 0, 1)  # after the first iteration
    assert got["termination"] == ref["termination"] and got["num_successful_steps"] == ref["num_successful_steps"]
    assert got["final_cost"] == ref["final_cost"]
    for a, b in zip(xgot, xref):
        assert np.array_equal(a, b)


_RCCL_RANK_SNIPPET = r"""
import os, sys, time, numpy as np
sys.path.insert(0, {root!r})
rank, world, tmp = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
import mavmap_amd
from mavmap_amd import synth
opts = dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10, device=rank)
full = synth.make_config("C3", scale=0.05, seed=11)
uid_path = os.path.join(tmp, "uid.bin")
if rank == 0:
    uid = mavmap_amd.rccl_unique_id()
    with open(uid_path + ".tmp", "wb") as fh:
        fh.write(uid)
    os.replace(uid_path + ".tmp", uid_path)
else:
    t0 = time.time()
    while not os.path.exists(uid_path):
        assert time.time() - t0 < 120, "rank 0 never published the RCCL id"
        time.sleep(0.05)
    uid = open(uid_path, "rb").read()
shard, owned = full.shard_by_point(rank, world)
with mavmap_amd.Session(shard, opts) as s:
    s.set_rccl(uid, rank, world)
    res = s.solve()
    poses, intr, pts = s.get_params()
np.savez(os.path.join(tmp, f"rank{{rank}}.npz"), poses=poses, intr=intr, pts=pts, owned=owned,
         cost=res["final_cost"], initial=res["initial_cost"], ok=res["num_successful_steps"], bad=res["num_unsuccessful_steps"],
         term=res["termination"], nres=res["num_residuals"], npar=res["num_parameters_reduced"])
"""


def test_native_rccl_two_or_more_ranks(mavba, tmp_path):
    """The native RCCL exchange with a REAL communicator: one process per visible GPU (skipped below two), ncclAllReduce
    enqueued on every session's stream, deferred read-back of the evaluation's scalars. Every rank must end with the
    cameras, cost and step counts of the single-GPU solve (a cost summed twice over the ranks - the evaluation's slots
    caught in the candidate's collective - changes the accept/reject decisions within a few iterations)."""
    import os
    import subprocess
    import sys
    from tests.conftest import ROOT
    world = mavba.device_count()
    if world < 2:
        pytest.skip(f"needs >= 2 GPUs for a multi-rank RCCL communicator ({world} visible)")
    world = min(world, 8)
    full = synth.make_config("C3", scale=0.05, seed=11)
    single = full.copy()
    _, r1 = mavba.bundle_adjustment(single, global_opts())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", _RCCL_RANK_SNIPPET.format(root=ROOT), str(r), str(world), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for pr in procs:
        try:
            outs.append(pr.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, (r, outs[r][1][-3000:])
    pts = np.zeros_like(full.points)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for d in ranks:
        assert int(d["term"]) == r1["termination"]
        assert int(d["ok"]) == r1["num_successful_steps"] and int(d["bad"]) == r1["num_unsuccessful_steps"]
        assert int(d["nres"]) == r1["num_residuals"] and int(d["npar"]) == r1["num_parameters_reduced"]
        assert abs(float(d["initial"]) - r1["initial_cost"]) <= 1e-10 * r1["initial_cost"]
        assert abs(float(d["cost"]) - r1["final_cost"]) <= 1e-9 * r1["final_cost"]
        assert np.array_equal(d["poses"], ranks[0]["poses"]) and np.array_equal(d["intr"], ranks[0]["intr"])
        pts[d["owned"]] = d["pts"]
    assert_params_close(dict(poses=ranks[0]["poses"], intrinsics=ranks[0]["intr"], points=pts), single, tol=1e-8)


def test_the_drivers_multi_gpu_bench_command_with_two_ranks_on_one_gpu(mavba):
    """Exactly the command the driver uses for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), here with the
    two ranks SHARING this box's GPU: MAVBA_DIST_BACKEND=gloo (RCCL refuses two ranks on one device), so the exchange goes
    through the torch.distributed hook. What it pins before an 8-GPU node exists: argument parsing and the rank / world
    environment, the sharding of the configuration, the barrier + max-over-ranks timing, ONE JSON line from rank 0 with the
    contract's keys, n_gpus, "scaling": "strong" and the Amdahl bound of the replicated solve."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from tests.conftest import ROOT
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--config", "C2", "--scale", "0.2",
           "--no-cpu-baseline"]
    env = dict(os.environ, MAVBA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints, nobody else
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["cpu_baseline"] is None  # (rank 0 at N = 1 only)
    assert "sharded over 2 ranks" in d["config"]["parallelism"]
    bound = d["scaling_model"]["expected_speedup_bound"]
    assert 1.0 <= bound["2"] <= 2.0 and bound["2"] <= bound["4"] <= bound["8"] <= 8.0
    assert abs(d["ms_per_step"] * d["value"] - 1e3) < 1.0  # iterations / second and milliseconds / iteration of the same clock


@pytest.mark.parametrize("world,staged", [(2, False), (4, False), (3, True)])
def test_one_process_several_ranks_through_mavba_solve(mavba, oracle, monkeypatch, world, staged):
    """MAVBA_GPUS=N: ONE mavba_solve call - what an unchanged mapper.cc issues through the shim - shards the points over N
    ranks (host threads, one session each) inside the process and all-reduces the reduced camera system through the
    in-process exchange (owner-computes-slice kernel over peer pointers; MAVBA_GPUS_STAGED: the hipMemcpyPeer path).
    MAVBA_GPUS_SAME_DEVICE lets the ranks share this box's only GPU: same code, same result as the single-rank call."""
    p = synth.make_config("C3", scale=0.03, seed=23)
    p.point_const[::11] = 1
    opts = global_opts()
    single, e1 = p.copy(), np.full(p.num_points, np.nan)
    _, r1 = mavba.bundle_adjustment(single, opts, point3D_errors=e1)
    monkeypatch.setenv("MAVBA_GPUS", str(world))
    monkeypatch.setenv("MAVBA_GPUS_SAME_DEVICE", "1")
    monkeypatch.setenv("MAVBA_GPUS_MIN_OBS", "0")
    if staged:
        monkeypatch.setenv("MAVBA_GPUS_STAGED", "1")
    multi, e2 = p.copy(), np.full(p.num_points, np.nan)
    cost, r2 = mavba.bundle_adjustment(multi, opts, point3D_errors=e2)
    assert r2["termination"] == r1["termination"]
    assert r2["num_successful_steps"] == r1["num_successful_steps"] and r2["num_unsuccessful_steps"] == r1["num_unsuccessful_steps"]
    for k in ("num_residuals", "num_residuals_reduced", "num_parameters_reduced"):
        assert r2[k] == r1[k], k
    assert abs(r2["initial_cost"] - r1["initial_cost"]) <= 1e-10 * r1["initial_cost"]
    assert abs(r2["final_cost"] - r1["final_cost"]) <= 1e-9 * r1["final_cost"]
    assert_params_close(multi, single, tol=1e-8)
    m = ~np.isnan(e1)
    assert np.array_equal(m, ~np.isnan(e2)) and rel_err(e2[m], e1[m]) < 1e-8
    assert np.array_equal(multi.points[::11], p.points[::11])  # constant points stay put on every rank
    q = p.copy()
    ro, _ = oracle.solve(q, oracle.options(**opts))
    assert abs(r2["final_cost"] - ro["final_cost"]) <= 1e-6 * ro["final_cost"]
    assert_params_close(multi, q)


def test_one_process_several_ranks_a_failing_rank_does_not_hang(mavba, monkeypatch):
    """A rank that cannot build its session (bad camera model) must take the others down with an error, not leave them
    waiting at the exchange's barrier."""
    p = synth.make_config("C3", scale=0.02, seed=29)
    p.camera_model[1] = 7
    monkeypatch.setenv("MAVBA_GPUS", "3")
    monkeypatch.setenv("MAVBA_GPUS_SAME_DEVICE", "1")
    monkeypatch.setenv("MAVBA_GPUS_MIN_OBS", "0")
    with pytest.raises(mavba.MavbaError):
        mavba.bundle_adjustment(p.copy(), global_opts())
