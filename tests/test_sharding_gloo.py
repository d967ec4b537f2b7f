"""Multi-GPU path on CPU: world_size-2 gloo processes. Checks the decomposition the device path
relies on: points (with all their observations) are sharded, cameras are
replicated, and every camera-side quantity that is all-reduced — squared column norms (Jacobi
scaling / LM diagonal), J_c^T J_c blocks, the gradient, the cost — sums to the single-rank value.
Also drives the same `allreduce(ptr, count, op)` hook protocol bench.py installs, over gloo."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _camera_sums(prob, oracle):
    cost, r, Jc, Jp, Jk = oracle.eval_jacobian(prob, jac_mode=1)
    NI, NC = prob.num_images, prob.num_cameras
    cam = prob.image_camera[prob.obs_image]
    out = np.zeros(1 + NI * 6 + NI * 36 + NI * 6 + NC * 9)
    out[0] = cost
    n2 = np.zeros((NI, 6)); np.add.at(n2, prob.obs_image, (Jc ** 2).sum(1))
    pp = np.zeros((NI, 6, 6)); np.add.at(pp, prob.obs_image, np.einsum("oia,oib->oab", Jc, Jc))
    g = np.zeros((NI, 6)); np.add.at(g, prob.obs_image, np.einsum("oia,oi->oa", Jc, r))
    k2 = np.zeros((NC, 9)); np.add.at(k2, cam, (Jk ** 2).sum(1))
    out[1:] = np.concatenate([n2.ravel(), pp.ravel(), g.ravel(), k2.ravel()])
    return out


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mavmap_amd import synth
        from tests import oracle_lib as O
        O.set_threads(1)
        full = synth.make_config("C3", scale=0.02, seed=5)
        shard, owned = full.shard_by_point(rank, world)
        # (1) the shards partition points and observations
        counts = torch.tensor([shard.num_points, shard.num_obs], dtype=torch.int64)
        dist.all_reduce(counts)
        assert counts.tolist() == [full.num_points, full.num_obs]
        assert np.array_equal(shard.points, full.points[owned])
        assert np.array_equal(shard.poses, full.poses) and np.array_equal(shard.intrinsics, full.intrinsics)
        # (2) all-reduced camera-side sums == single-rank sums, through the hook protocol
        local = _camera_sums(shard, O)
        buf = torch.from_numpy(local.copy())

        def hook(tensor, count, op):  # same contract as the device hook: in place, blocking
            if op == 2:
                dist.all_reduce(tensor[:count - 1], op=dist.ReduceOp.SUM)
                dist.all_reduce(tensor[count - 1:count], op=dist.ReduceOp.MAX)
            else:
                dist.all_reduce(tensor[:count], op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)

        hook(buf, buf.numel(), 0)
        ref = _camera_sums(full, O)
        err = np.abs(buf.numpy() - ref).max() / np.abs(ref).max()
        assert err < 1e-12, err
        # (3) "used" flags widen with a max-reduce: an image without observations on this rank is
        # still a free block if another rank sees it
        used = torch.zeros(full.num_images, dtype=torch.float64)
        used[np.unique(shard.obs_image)] = 1.0
        hook(used, used.numel(), 1)
        assert used.min().item() == 1.0
        mixed = torch.tensor([1.0 + rank, 10.0 * (rank + 1), 5.0 - rank], dtype=torch.float64)
        hook(mixed, 3, 2)  # op 2: sums, then max in the last slot
        assert mixed.tolist() == [3.0, 30.0, 5.0]
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_shard_balance_and_rank_zero_owns_the_priors():
    from mavmap_amd import synth
    full = synth.make_scene(num_images=12, num_points=900, track_len=4, models=[1], seed=3, rot_priors=True)
    obs = []
    for r in range(4):
        s, owned = full.shard_by_point(r, 4)
        obs.append(s.num_obs)
        assert len(s.rot_prior_image) == (len(full.rot_prior_image) if r == 0 else 0)
        assert s.obs_point.max() < s.num_points
    assert sum(obs) == full.num_obs
    assert max(obs) - min(obs) <= 0.05 * full.num_obs
