#!/usr/bin/env python
"""bench.py — LM iterations / second of the MI355X bundle-adjustment backend.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3]

A "step" is ONE Levenberg-Marquardt iteration of the reference's global BA (Jacobian sweep when the
previous step was accepted, Schur complement, dense reduced solve, back-substitution, candidate
cost, accept/reject) on the synthetic configuration named in `config.workload`. The session keeps
the problem resident in HBM; when a solve terminates inside the timed region the parameters are
reset to the perturbed start (a device-to-device copy) and the next solve begins, so every timed
step is a real iteration of a real solve. With N > 1 the 3-D points (and their observations) are
sharded over the ranks and the reduced camera system is all-reduced over RCCL once per iteration.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix peak

WORKLOADS = {
    "C2": "C2: 100 images / 30000 points / 300000 obs global BA, PINHOLE, 1 shared camera",
    "C3": "C3: 500 images / 200000 points / 2000000 obs global BA, PINHOLE+OPENCV, 2 shared cameras",
    "C5": "C5: 2000 images / 1000000 points / ~10M obs global BA, PINHOLE+OPENCV, rotation priors, long tracks",
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_work(stats_name, prob, sess_info):
    """(bound, amount per launch, unit) of one kernel — SURVEY.md §8(d) figures, stated in DESIGN.md."""
    from mavmap_amd import _abi as A
    n_obs, n_pts = prob.num_obs, prob.num_points
    K = np.array([A.MODEL_NUM_PARAMS[int(m)] for m in prob.camera_model])
    k_obs = K[prob.image_camera[prob.obs_image]]
    if stats_name in ("jacobian_sweep", "point_front"):
        # the Jacobian sweep is priced with J written out (48 B read + 16 + 2 (9 + K) 8 B written per
        # observation). k_point_front no longer materialises J (residual + Jacobian stay in registers, the per-point sums,
        # the 3x3 factors and the Schur entry records are produced in the same pass): it is still reported against these
        # bytes, as SURVEY asks; what it really moves is `front_own_bytes` below.
        return "hbm", float(np.sum(48 + 16 + 2 * (9 + k_obs) * 8)), "B"
    if stats_name == "cost_only":
        return "hbm", 48.0 * n_obs, "B"
    if stats_name in ("point_sums", "point_reduce"):
        # r, Jp and the Jk planes (padded to the widest model) in; Cu, gu per point and Wk per (point, camera) out
        kmax = int(K.max())
        return "hbm", (64.0 + 16.0 * kmax) * n_obs + 72.0 * n_pts + 216.0 * sess_info["intr_entries"], "B"
    if stats_name in ("schur_clusters", "schur_fused"):
        # Schur formation = sum_p (6 L_p + K_p)^2 * 3 * 2 flops (L_p observations of point p,
        # K_p refined intrinsics of the cameras that see it). The matrix-instruction flops the kernel really
        # executes (structural zeros of the stacked entry matrix included) are reported beside it as executed_flops.
        return "mfma", schur_algorithmic_flops(prob), "FLOP"
    if stats_name == "camera_sweep":
        return "hbm", 44.0 * n_obs, "B"
    if stats_name == "entries_pose":
        return "hbm", (96 + 48 + 8 + 192.0) * n_obs, "B"
    if stats_name == "backsub_points":
        # k_backsub_points_packed recomputes the Jacobians: pixel + image index per observation (20 B), per point its
        # coordinates, factor, h, diagonal, gradient, scales in and candidate + step out (216 B).
        return "hbm", 20.0 * n_obs + 216.0 * n_pts, "B"
    if stats_name == "schur_chunks_pp":
        return "hbm", 296.0 * sess_info["schur_terms"][0], "B"
    if stats_name == "chol_factor":
        # Graded on the work the kernel DOES: the flops of the structured factorisation (tiles inside the envelope of
        # the nested-dissection order, as a sparse Cholesky would). SURVEY.md 8(d)'s dense-equivalent n^3/3 + 2 n^2 is
        # carried beside it as dense_equivalent_*; it is not a roofline (it exceeds the peak at C5).
        return "mfma", sess_info["factor_flops"], "FLOP"
    return None, 0.0, ""


PMC_KERNEL = {  # bench timer name -> rocprofv3 kernel-name prefix in profiles/*pmc_traffic*.json
    "jacobian_sweep": "k_jacobian_sweep", "cost_only": "k_cost_only", "point_reduce": "k_point_reduce",
    "camera_sweep": "k_camera_sweep", "entries_pose": "k_entries_pose", "entries_intr": "k_entries_intr",
    "backsub_points": "k_backsub_points", "schur_chunks_pp": "k_schur_chunks<6, 6", "schur_chunks_ip": "k_schur_chunks<9, 6",
    "schur_chunks_ii": "k_schur_chunks<9, 9", "schur_clusters": "k_schur_clusters", "schur_finalize": "k_schur_finalize",
    "chol_factor": "k_chol_persist", "chol_backsolve": "k_chol_backsolve_all", "point_front": "k_point_front<8, true",
    "point_front_sums": "k_point_front<8, false", "schur_fused": "k_schur_rows",
}
# kernels every rank runs in full when the points are sharded (the reduced camera system is factorised redundantly)
REPLICATED = {"chol_factor", "chol_backsolve", "schur_finalize", "update_cameras", "camera_reduce", "eval_tail", "reduce", "memset_S", "scales",
              "cam_prepare", "rot_prior"}


PMC_ROUNDS = ("r06", "r05", "r04")  # newest first: the committed counter passes (scripts/measure_all.sh, scripts/_dbg/evidence_c5.sh)


def pmc_file(kind, config):
    """profiles/<round>_pmc_<kind>_<config>.json of the newest round that has one (relative path), or None."""
    for rnd in PMC_ROUNDS:
        rel = os.path.join("profiles", f"{rnd}_pmc_{kind}_{config}.json")
        if os.path.exists(os.path.join(ROOT, rel)):
            return rel
    return None


def mfma_counters(config, scale, world):
    """MFMA utilisation of the reduced solve / cluster kernels from the committed rocprofv3 counter pass
    (scripts/pmc_mfma.sh -> profiles/<round>_pmc_mfma_<config>.json), or None."""
    rel = pmc_file("mfma", config)
    if scale != 1.0 or world != 1 or rel is None:
        return None
    return json.load(open(os.path.join(ROOT, rel))).get("summary")


def pmc_traffic(name, config, scale, world):
    """HBM bytes per launch of one bench kernel from the committed rocprofv3 --pmc passes (separate
    FETCH_SIZE / WRITE_SIZE runs of this same command, scripts/pmc_traffic.sh), or None."""
    rel = pmc_file("traffic", config)
    if scale != 1.0 or world != 1 or rel is None:
        return None
    k = json.load(open(os.path.join(ROOT, rel)))["kernels"]
    if name == "chol_factor" and not any(n.startswith("k_chol_persist") for n in k):
        # launch-per-panel schedule: every k_chol_* launch of a solve but the backward substitution
        chol = {n: v for n, v in k.items() if n.startswith("k_chol_") and not n.startswith("k_chol_backsolve")}
        solves = max((v["launches"] for n, v in k.items() if n.startswith("k_chol_backsolve_all")), default=0)  # one per solve
        return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in chol.values()) / solves if solves else None
    pre = PMC_KERNEL.get(name)
    hit = [v for n, v in k.items() if pre and n.startswith(pre)]
    return hit[0]["hbm_bytes_per_launch"] if hit else None


def pmc_traffic_per_iteration(config, scale, world, launches_per_iteration):
    """HBM bytes one LM iteration moves: sum over the kernels of the committed counter pass of bytes per launch x launches
    per iteration (launch counts of THIS run's timed region). None without a committed pass."""
    rel = pmc_file("traffic", config)
    if scale != 1.0 or world != 1 or rel is None:
        return None
    k = json.load(open(os.path.join(ROOT, rel)))["kernels"]
    total, used = 0.0, []
    for timer, per_iter in launches_per_iteration.items():
        pre = PMC_KERNEL.get(timer)
        hit = [v for n, v in k.items() if pre and n.startswith(pre)]
        if hit:
            total += hit[0]["hbm_bytes_per_launch"] * per_iter
            used.append(timer)
    return dict(bytes=total, kernels=used, source=rel) if used else None


def schur_algorithmic_flops(prob):
    """sum over points of (6 L_p + K_p)^2 * 3 * 2."""
    from mavmap_amd import _abi as A
    K = np.array([A.MODEL_NUM_PARAMS[int(m)] for m in prob.camera_model], np.int64)
    K = np.where(np.asarray(prob.intr_const) != 0, 0, K)
    L = np.bincount(prob.obs_point, minlength=prob.num_points).astype(np.int64)
    cam = prob.image_camera[prob.obs_image].astype(np.int64)
    pair = np.unique(prob.obs_point.astype(np.int64) * prob.num_cameras + cam)   # distinct (point, camera)
    Kp = np.bincount(pair // prob.num_cameras, weights=K[pair % prob.num_cameras], minlength=prob.num_points)
    return float(np.sum((6 * L + Kp) ** 2 * 6.0))


def count_pp_terms(prob):
    """Number of (observation, observation) pairs inside a point with image(a) >= image(b)."""
    order = np.argsort(prob.obs_point, kind="stable")
    cnt = np.bincount(prob.obs_point, minlength=prob.num_points).astype(np.int64)
    return int(np.sum(cnt * (cnt + 1) // 2))


def measure_other_config(name, steps, warmup, device):
    """A short pass over another BASELINE config on this GPU (single rank): LM iterations/s without instrumentation, then the
    same steps with HIP events for the dominant kernel and its roofline fraction. No CPU baseline, no counters."""
    import mavmap_amd
    from mavmap_amd import synth, _abi as A
    t0 = time.time()
    prob = synth.make_config(name)
    opts = dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10, device=device, profile_kernels=0)
    with mavmap_amd.Session(prob, opts) as sess:
        import torch

        def run_steps(k):
            remaining, idle = k, 0
            while remaining > 0:
                done, term = sess.iterate(remaining)
                remaining -= done
                idle = idle + 1 if done == 0 else 0
                if idle > 2:
                    raise RuntimeError("bench: solver makes no progress")
                if term != A.TERM_RUNNING and remaining > 0:
                    sess.reset()
        run_steps(warmup)
        torch.cuda.synchronize()
        t = time.perf_counter()
        run_steps(steps)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t
        sess.reset()
        sess.set_profiling(True)
        run_steps(warmup)
        s0 = sess.kernel_stats()
        run_steps(steps)
        torch.cuda.synchronize()
        s1 = sess.kernel_stats()
        sess.set_profiling(False)
        info = sess.info()
        per = {}
        for k, v in s1.items():
            b = s0.get(k, dict(launches=0, total_ms=0.0))
            n, ms = v["launches"] - b["launches"], v["total_ms"] - b["total_ms"]
            if n > 0:
                per[k] = dict(launches=n, total_ms=ms, avg_ms=ms / n)
        tot = sum(v["total_ms"] for v in per.values()) or 1.0
        rows = []
        for k, v in sorted(per.items(), key=lambda kv: -kv[1]["total_ms"]):
            bound, amount, unit = algorithmic_work(k, prob, info)
            row = dict(kernel=k, avg_ms=round(v["avg_ms"], 5), share=round(v["total_ms"] / tot, 4))
            if bound:
                peak = HBM_PEAK_GBS if bound == "hbm" else FP64_MFMA_PEAK_TFLOPS
                ach = amount / (v["avg_ms"] * 1e-3) / (1e9 if bound == "hbm" else 1e12)
                row.update(bound=bound, achieved=round(ach, 3), peak=peak, unit="GB/s" if bound == "hbm" else "TFLOP/s", frac=round(ach / peak, 4))
            rows.append(row)
        dom = next((r for r in rows if r.get("bound")), None)
        sess.reset()
        final = sess.solve()
    out = {"workload": WORKLOADS[name], "value": round(steps / elapsed, 3), "unit": "iter/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(1e3 * elapsed / steps, 4), "dominant_kernel": dom, "kernels": rows[:6],
           "solve": {"iterations": final["num_successful_steps"] + final["num_unsuccessful_steps"], "termination": final["termination_name"],
                     "rmse_px": round(float(np.sqrt(final["final_cost"] / max(final["num_residuals"], 1))), 6),
                     "setup_seconds": round(final["setup_seconds"], 4)},
           "wall_seconds": round(time.time() - t0, 1)}
    log(f"other config {name}: {out['value']} iter/s, {out['ms_per_step']} ms per iteration, dominant {dom['kernel'] if dom else None}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--config", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debugging only)")
    ap.add_argument("--problem", default=None, help="replay file (BAProblem.save / the shim's MAVBA_DUMP_DIR) instead of a synthetic config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=6)
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short C2 / C5 passes of the default single-GPU run (--no-cpu-baseline skips them too: the tuning scripts' form)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        log(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
        sys.exit(2)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        log("bench.py: no GPU visible — the mavba backend has no CPU path")
        sys.exit(3)
    # one process per GPU; MAVBA_DIST_BACKEND=gloo lets several ranks share one GPU (testing only)
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("MAVBA_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    import mavmap_amd
    from mavmap_amd import synth, _abi as A
    from mavmap_amd import build as mavba_build
    if rank == 0 and not mavba_build.is_current():
        # never measure a stale library: rebuild from the sources that travel with it, or refuse
        log("bench.py: libmavba.so does not match mavmap_amd/csrc (build stamp) - rebuilding")
        try:
            mavba_build.build(force=True)
        except Exception as e:  # noqa: BLE001
            log(f"bench.py: rebuild failed ({e}); refusing to benchmark a stale library")
            sys.exit(4)
    if world > 1:
        dist.barrier()
    mavmap_amd.load()

    t0 = time.time()
    if args.problem:
        from mavmap_amd.problem import BAProblem
        full, _ = BAProblem.load(args.problem)
        args.config = "replay"
        WORKLOADS["replay"] = (f"replay of {os.path.basename(args.problem)}: {full.num_images} images / {full.num_points} points / "
                               f"{full.num_obs} obs")
    else:
        full = synth.make_config(args.config, scale=args.scale)
    prob, _ = full.shard_by_point(rank, world)
    log(f"[rank {rank}] scene {args.config}: {full.num_images} images / {full.num_points} points / "
        f"{full.num_obs} obs (this rank: {prob.num_points} points / {prob.num_obs} obs), generated in {time.time() - t0:.1f}s")

    # (no event brackets in the timed region: a bracket drains the queue for ~10 us per kernel - 0.1 ms of a 1 ms iteration;
    # the kernel timers come from a second pass over the same K steps, below)
    opts = dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10,  # mapper.cc:170-174
                device=local_rank, profile_kernels=0)
    sess = mavmap_amd.Session(prob, opts)

    exchange = "none"
    if world > 1:
        # the reduced camera system is summed over ranks once per linear solve: natively (ncclAllReduce enqueued on the
        # session's stream by the library itself) unless MAVBA_DIST=torch asks for the torch.distributed hook
        native = os.environ.get("MAVBA_DIST", "rccl") == "rccl" and os.environ.get("MAVBA_DIST_BACKEND", "nccl") == "nccl"
        if native:
            ok = 1
            try:
                uid = [mavmap_amd.rccl_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                sess.set_rccl(uid[0], rank, world)
            except Exception as e:  # librccl not loadable, communicator set-up failed, ...
                log(f"[rank {rank}] native RCCL exchange unavailable ({e}); falling back to the torch.distributed hook")
                ok = 0
            # every rank must take the same path
            flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{local_rank}")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native = bool(flag.item())
            if not native:
                sess.close()
                sess = mavmap_amd.Session(prob, opts)
        if native:
            exchange = "RCCL all-reduce inside the library (stream-ordered)"
        else:
            from mavmap_amd.dist import make_allreduce
            sess.set_allreduce(make_allreduce(torch.device(f"cuda:{local_rank}")), rank, world)
            exchange = "torch.distributed all-reduce hook (host-synchronised)"

    # who takes part (N > 1): every rank's device and shard, and a count of the ranks through the SAME exchange the solve uses
    # (the session's collective sums a one per rank) - the first run on a multi-GPU node should be readable from the line alone
    rank_info = dict(rank=rank, local_rank=local_rank, device=torch.cuda.get_device_name(local_rank), pid=os.getpid(),
                     points=int(prob.num_points), observations=int(prob.num_obs))
    ranks = [rank_info]
    ranks_confirmed = 1
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, rank_info)
        one = torch.ones(1, dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(one)
        ranks_confirmed = int(round(float(one.item())))

    def run_steps(k):
        remaining, solves, idle = k, 0, 0
        while remaining > 0:
            done, term = sess.iterate(remaining)
            remaining -= done
            idle = idle + 1 if done == 0 else 0
            if idle > 2:
                raise RuntimeError("bench: solver makes no progress")
            if term != A.TERM_RUNNING:
                solves += 1
                if remaining > 0:
                    sess.reset()
        return solves

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    barrier()
    t_start = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the same K steps again, from the same start, with HIP events around every kernel: the per-kernel averages of the
    # roofline / kernel table (rocprofv3's kernel stats of this command agree with them, profiles/)
    sess.reset()
    sess.set_profiling(True)
    run_steps(args.warmup)
    stats0 = sess.kernel_stats()
    barrier()
    t_prof = time.perf_counter()
    run_steps(args.steps)
    barrier()
    elapsed_profiled = time.perf_counter() - t_prof
    stats1 = sess.kernel_stats()
    sess.set_profiling(False)

    # one complete solve for the record (RMSE, iteration count, setup time). Single process: on a SECOND session of the
    # process, without the event brackets - what a bundle_adjustment() call of a running mapper costs (device buffers,
    # page-locked blocks and host scratch come from the process-wide pools; the first session's set-up, which fills them,
    # is reported beside it).
    # The north star's "Jacobian kernel": the materialising sweep (J written out, SURVEY 8(d)'s 272 / 336 B per observation)
    # is no longer part of the LM loop - it is timed here as a probe, HIP events around 20 launches on the session's stream.
    jacobian_probe = None
    if rank == 0:
        ms = sess.time_jacobian(20)
        jb = algorithmic_work("jacobian_sweep", prob, None)[1]
        jacobian_probe = {"kernel": "k_jacobian_sweep (probe: not in the LM loop since the J-free front end)", "avg_ms": round(ms, 5),
                          "obs_per_sec": round(prob.num_obs / (ms * 1e-3), 1), "bound": "hbm",
                          "achieved": round(jb / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(jb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "traffic": pmc_traffic("jacobian_sweep", args.config, args.scale, world),
                          "note": "algorithmic bytes 48 + 16 + 2*(9+K)*8 per observation, this rank, J materialised"}
    first_setup = None
    info = sess.info()
    if world == 1:
        first_setup = float(sess.result()["setup_seconds"])
        sess.close()  # (its buffers go back to the pools: the next session is what a second call of a mapper run sees)
        with mavmap_amd.Session(prob, dict(opts, profile_kernels=0)) as s2:
            final = s2.solve()
    else:
        sess.reset()
        final = sess.solve()
    rmse = float(np.sqrt(final["final_cost"] / max(final["num_residuals"], 1)))

    if rank == 0:
        per = {}
        for name, s1 in stats1.items():
            s0 = stats0.get(name, dict(launches=0, total_ms=0.0))
            n, ms = s1["launches"] - s0["launches"], s1["total_ms"] - s0["total_ms"]
            if n > 0:
                per[name] = dict(launches=n, total_ms=ms, avg_ms=ms / n)
        tot_ms = sum(v["total_ms"] for v in per.values()) or 1.0
        table = []
        hybrid = "schur_fused" in per and "point_front" in per  # the tail of the point order (long tracks, constant points) beside the fused kernel
        for name, v in sorted(per.items(), key=lambda kv: -kv[1]["total_ms"]):
            bound, amount, unit = algorithmic_work(name, prob, info)
            if hybrid and name == "point_front":
                bound = None  # (covers the tail's observations only: not priced against the whole problem's sweep bytes)
            row = dict(kernel=name, launches=v["launches"], avg_ms=round(v["avg_ms"], 5),
                       share=round(v["total_ms"] / tot_ms, 4))
            if bound == "hbm":
                ach = amount / (v["avg_ms"] * 1e-3) / 1e9
                row.update(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                           frac=round(ach / HBM_PEAK_GBS, 4), algorithmic_bytes=amount)
            elif bound == "mfma":
                ach = amount / (v["avg_ms"] * 1e-3) / 1e12
                row.update(bound="mfma", achieved=round(ach, 3), peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                           frac=round(ach / FP64_MFMA_PEAK_TFLOPS, 4), algorithmic_flops=amount)
            if name == "schur_fused" and per.get("camera_sweep", {}).get("launches", 0) * 2 < v["launches"]:
                # (problems of up to 1 M observations: csrc/session_lm.hip, MAVBA_SWEEP_RIDE_MAX_OBS)
                row["note"] = ("this launch also runs the camera sweep's chunks as extra work-groups (no separate k_camera_sweep "
                               "launch per evaluation); achieved / frac still price the Schur-formation flops alone")
            table.append(row)
            log("  {kernel:18s} n={launches:5d} avg={avg_ms:9.4f} ms share={share:6.1%}  ".format(**row) +
                (f"{row['achieved']} {row['unit']} ({row['frac']:.1%} of {row['bound']} peak)" if bound else ""))
        # `roofline` = the single KERNEL with the largest share of the timed region - the name rocprofv3's kernel stats put
        # first too (profiles/), so its average duration can be cross-checked. Graded on work done: algorithmic bytes for
        # HBM-bound kernels, the flops the algorithm needs for the matrix-core kernels (never a dense-equivalent count).
        traffic_src = f"{pmc_file('traffic', args.config)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; not re-measured in this run)"
        mfma = mfma_counters(args.config, args.scale, world)
        dominant = next((r for r in table if r.get("bound")), None)
        mdl = info.get("chol_model_forward_us", 0.0)
        chain_note = (f" (persistent launch: the host-side timing model that fills its task queues predicts {mdl:.0f} us for the forward pass"
                      f" - 12.2 us per dependent tile column, 9-12 us per child -> parent hand-off, profiles/r06_chol_trace_C3.txt)") if mdl > 0 else \
                     f" (launch-per-panel schedule: {info['chain_steps']} dependent 64-column panel steps)"
        roofline = None
        if dominant:
            roofline = dict(kernel=dominant["kernel"], rocprof_kernel=PMC_KERNEL.get(dominant["kernel"]), bound=dominant["bound"],
                            achieved=dominant["achieved"], peak=dominant["peak"], unit=dominant["unit"], frac=dominant["frac"],
                            avg_ms=dominant["avg_ms"], share=dominant["share"],
                            traffic=pmc_traffic(dominant["kernel"], args.config, args.scale, world), traffic_source=traffic_src)
            if dominant["kernel"] == "schur_clusters":
                ex = info["cluster_flops"] / (dominant["avg_ms"] * 1e-3) / 1e12
                roofline.update(executed_flops=info["cluster_flops"], executed_tflops=round(ex, 3),
                                executed_frac=round(ex / FP64_MFMA_PEAK_TFLOPS, 4))
                roofline["note"] = ("k_schur_clusters: Schur complement of point clusters as E E^T on v_mfma_f64_16x16x4_f64; achieved = "
                                    "SURVEY 8(d) algorithmic flops sum_p (6 L_p + K_p)^2 * 6 / kernel time; executed_* = the matrix-"
                                    "instruction flops really issued (structural zeros of the stacked entry matrix included); traffic "
                                    "= HBM bytes per launch")
            if dominant["kernel"] == "schur_fused":
                sweep_bytes = algorithmic_work("jacobian_sweep", prob, info)[1]
                ex = info["cluster_flops"] / (dominant["avg_ms"] * 1e-3) / 1e12
                roofline.update(executed_flops=info["cluster_flops"], executed_tflops=round(ex, 3),
                                executed_frac=round(ex / FP64_MFMA_PEAK_TFLOPS, 4),
                                sweep_equivalent_gbs=round(sweep_bytes / (dominant["avg_ms"] * 1e-3) / 1e9, 1),
                                sweep_equivalent_frac=round(sweep_bytes / (dominant["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                mfma_util=None if not mfma else (mfma.get("schur_fused") or {}).get("mfma_util"),
                                mfma_counters=None if not mfma else mfma.get("schur_fused"))
                roofline["note"] = ("k_schur_rows: Jacobian evaluation, per-point sums, 3x3 factors AND the Schur complement of the point "
                                    "clusters (E E^T on v_mfma_f64_16x16x4_f64) in one kernel, no Jacobian and no entry records in HBM. "
                                    "While a v_mfma_f64_16x16x4 executes no vector instruction of either wave issues on its SIMD, and two waves' vector instructions do not overlap either (profiles/r06_issue_bench.txt), so the kernel is bound by the SUM of both; "
                                    "achieved / frac price only SURVEY 8(d)'s Schur-formation flops sum_p (6 L_p + K_p)^2 * 6 over the "
                                    "kernel's time; executed_* = matrix-instruction flops really issued; sweep_equivalent_* = SURVEY "
                                    "8(d)'s Jacobian-sweep bytes (J counted as if written) over the same time")
            if dominant["kernel"] == "chol_factor":
                de = info["dense_factor_flops"] / (dominant["avg_ms"] * 1e-3) / 1e12
                roofline.update(executed_flops=info["factor_flops"], dense_equivalent_tflops=round(de, 3),
                                mfma_util=None if not mfma else (mfma.get("reduced_solve") or {}).get("mfma_util"),
                                mfma_counters=None if not mfma else mfma.get("reduced_solve"))
                roofline["note"] = (f"forward factorisation of the reduced camera system (k_chol_persist; launch-per-panel k_chol_* at C5): "
                                    f"achieved = the flops of the structured factorisation ({info['factor_flops'] / 1e9:.2f} GFLOP: "
                                    f"{info['envelope_tiles']} of {info['dense_tiles']} tiles, {info['nd_parts']} concurrent fronts) / kernel "
                                    f"time. Bound by the dependent chain of tile columns and child -> parent hand-offs, not by matrix "
                                    f"throughput{chain_note}. dense_equivalent_tflops prices SURVEY 8(d)'s n^3/3 + 2 n^2 = "
                                    f"{info['dense_factor_flops'] / 1e9:.2f} GFLOP and is NOT a roofline (> peak at C5)")
        # (C3: k_schur_rows and k_chol_persist are within a few per cent of each other - which one leads changes from run to run;
        # the runner-up's figures are carried beside the dominant kernel's)
        second = next((r for r in table if r.get("bound") and r is not dominant), None)
        if roofline is not None and second is not None:
            roofline["runner_up"] = dict(kernel=second["kernel"], rocprof_kernel=PMC_KERNEL.get(second["kernel"]), bound=second["bound"],
                                         achieved=second["achieved"], peak=second["peak"], unit=second["unit"], frac=second["frac"],
                                         avg_ms=second["avg_ms"], share=second["share"])
        chol = next((r for r in table if r["kernel"] == "chol_factor"), None)
        back = next((r for r in table if r["kernel"] == "chol_backsolve"), None)
        reduced_solve = None
        if chol:
            tot = chol["avg_ms"] + (back["avg_ms"] if back else 0.0)
            reduced_solve = dict(avg_ms=round(tot, 5), factor_ms=chol["avg_ms"], backsolve_ms=back["avg_ms"] if back else None,
                                 share=round(chol["share"] + (back["share"] if back else 0.0), 4), bound="mfma",
                                 achieved=chol["achieved"], peak=chol["peak"], unit=chol["unit"], frac=chol["frac"],
                                 executed_flops=info["factor_flops"],
                                 dense_equivalent_flops=info["dense_factor_flops"],
                                 dense_equivalent_tflops=round(info["dense_factor_flops"] / (tot * 1e-3) / 1e12, 3),
                                 traffic=pmc_traffic("chol_factor", args.config, args.scale, world), traffic_source=traffic_src,
                                 mfma_counters=mfma,
                                 note=(f"achieved / frac = flops of the structured factorisation / time of the forward factorisation; "
                                       f"dense_equivalent_* = SURVEY 8(d)'s n^3/3 + 2n^2 over factor + backward substitution, a label "
                                       f"for comparison with dense solvers, not a roofline{chain_note}"),
                                 model_forward_us=round(info.get("chol_model_forward_us", 0.0), 1) or None)
        sweep = next((r for r in table if r["kernel"] == "jacobian_sweep"), None)
        front = next((r for r in table if r["kernel"] == "point_front"), None)
        fused = next((r for r in table if r["kernel"] == "schur_fused"), None)
        if fused is not None and (front is None or hybrid):  # the front end runs inside the cluster kernel: its share of that kernel is not separable
            tail_ms = front["avg_ms"] if front is not None else 0.0
            sb = algorithmic_work("jacobian_sweep", prob, info)[1]
            fe_ms = fused["avg_ms"] + tail_ms  # (+ k_point_front over the tail of the point order, where there is one)
            front = dict(kernel="schur_fused", avg_ms=round(fe_ms, 5), achieved=round(sb / (fe_ms * 1e-3) / 1e9, 1),
                         frac=round(sb / (fe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), tail_ms=tail_ms)

        front_own_bytes = float(48 * prob.num_obs + 192 * prob.num_obs + 288 * info["intr_entries"] + 144 * prob.num_points)
        cpu_baseline = None
        if world == 1 and not args.no_cpu_baseline:
            from tests import oracle_lib
            cores = oracle_lib.max_threads()
            oracle_lib.set_threads(cores)
            # the oracle's sparse linear solver (block-sparse Schur complement, one owner thread per row block, envelope
            # Cholesky: how a CPU solver organises SPARSE_SCHUR) - not its dense checker path
            oracle_lib.set_linear_solver(oracle_lib.SPARSE)
            q = full.copy()
            o = oracle_lib.options(max_num_iterations=args.cpu_iters, function_tolerance=1e-6, gradient_tolerance=1e-10)
            tc = time.time()
            ro, _ = oracle_lib.solve(q, o, jac_mode=1)
            oracle_lib.set_linear_solver(oracle_lib.DENSE)
            its = ro["num_successful_steps"] + ro["num_unsuccessful_steps"]
            cpu_baseline = dict(value=round(its / ro["solve_seconds"], 4), unit="iter/s", cores=cores, kind="port",
                                sample=f"first {its} LM iterations of the same {args.config} solve by the CPU oracle "
                                       f"(analytic Jacobians, OpenMP on {cores} threads, block-sparse Schur complement + envelope "
                                       f"Cholesky in acquisition order), {ro['solve_seconds']:.1f}s of {time.time() - tc:.1f}s wall")
            # the reference's own solver, where the box has it; otherwise said explicitly
            from tests import ceres_harness
            ceres = "unavailable"
            if ceres_harness.build() is None and "IS installed" in ceres_harness.why_unavailable():
                ceres = "failed: " + ceres_harness.why_unavailable()   # a box with Ceres on which the harness does not compile says so
            elif ceres_harness.build() is not None:
                try:
                    rc = ceres_harness.solve(full, dict(max_num_iterations=args.cpu_iters, function_tolerance=1e-6, gradient_tolerance=1e-10,
                                                        loss_scale_factor=1.0), threads=cores)
                    its_c = rc["num_successful_steps"] + rc["num_unsuccessful_steps"]
                    ceres = dict(value=round(its_c / rc["solve_seconds"], 4), unit="iter/s", cores=cores, kind="reference",
                                 sample=f"first {its_c} LM iterations of the same solve by Ceres (SPARSE_SCHUR) through oracle/ceres_check")
                except Exception as e:  # noqa: BLE001
                    ceres = f"failed: {e}"
            cpu_baseline["ceres"] = ceres
            if ceres == "unavailable":
                cpu_baseline["note"] = ("Ceres is not installed on this box: the baseline is the build's own CPU restatement, NOT the "
                                        "reference's Ceres path; the north star's >= 10x-over-Ceres target is unmeasured")
            log("cpu_baseline:", cpu_baseline)

        # Amdahl on this run's own split: what sharding the points can and cannot shorten (the exchange is left out: an upper bound)
        rep_ms = sum(r["avg_ms"] * r["launches"] for r in table if r["kernel"] in REPLICATED) / args.steps
        shd_ms = sum(r["avg_ms"] * r["launches"] for r in table if r["kernel"] not in REPLICATED) / args.steps
        one_gpu_shd = shd_ms * world  # (the sharded kernels of this rank cover 1 / world of the points)
        packed_bytes = (info["envelope_tiles"] * 4096 + info["matrix_dim"]) * 8.0

        def exchange_ms(n):
            bw = min(50.0 * (n - 1), 300.0) * 1e9
            return 1e3 * (20e-6 + 2.0 * (n - 1) / n * packed_bytes / bw) + 2 * 0.02

        scaling_model = {
            "replicated_ms_per_iteration": round(rep_ms, 4), "sharded_ms_per_iteration_this_rank": round(shd_ms, 4),
            "expected_speedup_bound": {str(n): round((one_gpu_shd + rep_ms) / (one_gpu_shd / n + rep_ms), 3) for n in (2, 4, 8)},
            "expected_speedup_with_exchange": {str(n): round((one_gpu_shd + rep_ms) / (one_gpu_shd / n + rep_ms + exchange_ms(n)), 3)
                                               for n in (2, 4, 8)},
            "exchange_ms": {str(n): round(exchange_ms(n), 4) for n in (2, 4, 8)},
            "packed_tiles_mb": round(packed_bytes / 1e6, 1),
            "note": "Amdahl bound from the event timers of this run: the reduced camera system is all-reduced and then factorised "
                    "REDUNDANTLY on every rank (chol_factor, chol_backsolve, schur_finalize, ...), only the per-point work shards. "
                    "expected_speedup_with_exchange adds the all-reduce of the packed tiles + right-hand side and two small "
                    "collectives per iteration, priced as 20 us + 2 (R-1)/R bytes / min(50 (R-1), 300) GB/s (one xGMI link per "
                    "peer): the honest expectation for this design. profiles/r05_multi_gpu_pricing.txt and r06_multi_gpu_pricing_C5_depth3.txt price the "
                    "alternatives on the real structures (subtree-aligned shards + subtree-to-rank factorisation: 1.36x at C3; C5 with its "
                    "depth-3 tree 1.68 / 2.20 / 2.68x at 2 / 4 / 8 ranks - designed in DESIGN.md section 7, not built: no multi-GPU node to run it on)"}
        # the other single-GPU configurations of BASELINE.json, short passes (the driver times only this command: C2 and C5,
        # whose dominant kernel is the factorisation, would otherwise be builder-only numbers)
        other_configs = None
        if world == 1 and args.config == "C3" and args.scale == 1.0 and not args.problem and not args.no_other_configs and not args.no_cpu_baseline:
            sess.close()
            other_configs = {}
            for name, k in (("C2", 60), ("C5", 30)):
                try:
                    other_configs[name] = measure_other_config(name, k, 6, local_rank)
                except Exception as e:  # noqa: BLE001
                    other_configs[name] = {"error": str(e)}
        value = args.steps / elapsed
        # HBM bytes one LM iteration moves (committed counter pass x this run's launch counts) against the algorithmic
        # bytes of ONE Jacobian sweep (48 + 16 + 2 (9 + K) 8 per observation)
        traffic_iter = pmc_traffic_per_iteration(args.config, args.scale, world,
                                                 {r["kernel"]: r["launches"] / args.steps for r in table})
        if traffic_iter:
            sweep_bytes = algorithmic_work("jacobian_sweep", prob, info)[1]
            traffic_iter.update(algorithmic_sweep_bytes=sweep_bytes, ratio=round(traffic_iter["bytes"] / sweep_bytes, 3))
        out = {
            "metric": "global-BA LM iterations/sec", "value": round(value, 3), "unit": "iter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong",
            "ms_per_step_with_event_timers": round(1e3 * elapsed_profiled / args.steps, 4),
            "timing_note": "value / ms_per_step: K steps without any instrumentation; the kernel table, roofline and reduced_solve "
                           "averages come from a second pass over the same K steps with HIP events around every kernel "
                           "(ms_per_step_with_event_timers: each bracket drains the queue for ~10 us; rounds 1-3 quoted that figure as value)",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not args.problem else "replay file",
            "config": {"workload": WORKLOADS[args.config] + ("" if args.scale == 1.0 else f" (scaled x{args.scale})"),
                       "images": full.num_images, "points": full.num_points, "observations": full.num_obs,
                       "parallelism": "single GPU" if world == 1 else f"points sharded over {world} ranks, " + exchange +
                       " of the reduced camera system",
                       "options": "max_iter 200, ftol 1e-6, gtol 1e-10, ptol 1e-8, Cauchy a=1, refine intrinsics"},
            "roofline": roofline,
            "reduced_solve": reduced_solve,
            "jacobian_sweep": jacobian_probe,
            "front_end": None if not front else {
                "kernel": "k_point_front" if front["kernel"] == "point_front" else "k_schur_rows (front end + cluster Schur complement in one kernel: the time is the WHOLE kernel's)" + (" + k_point_front over the points behind the clusters (%.4f ms)" % front["tail_ms"] if front.get("tail_ms") else ""), "avg_ms": front["avg_ms"], "obs_per_sec": round(prob.num_obs / (front["avg_ms"] * 1e-3), 1),
                "bound": "hbm", "achieved": front["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": front["frac"],
                "front_own_bytes": front_own_bytes if front["kernel"] == "point_front" else None,
                "own_achieved": round(front_own_bytes / (front["avg_ms"] * 1e-3) / 1e9, 1) if front["kernel"] == "point_front" else None,
                "traffic": pmc_traffic(front["kernel"], args.config, args.scale, world),
                "note": "the J-free Schur front end inside the LM loop: residual + Jacobian per observation in registers, per-point "
                        "sums, 3x3 factors and entry records in one pass. achieved / frac price SURVEY 8(d)'s Jacobian-sweep bytes "
                        "(J counted as if written, 272 / 336 B per observation) over this kernel's time although it does the work of "
                        "four former kernels; front_own_bytes = what it really reads and writes (48 B / observation in, 192 B / "
                        "observation + 288 B / (point, camera) + 144 B / point out)"},
            "reduced_system": {"n": info["reduced_dim"], "envelope_tiles": info["envelope_tiles"],
                               "dense_tiles": info["dense_tiles"], "matrix_dim": info["matrix_dim"], "nd_parts": info["nd_parts"],
                               "chain_steps": info["chain_steps"], "schur_clusters": info["num_clusters"],
                               "clustered_points": info["clustered_points"], "cluster_partials": info["cluster_partials"],
                               "factor_gflop_envelope": round(info["factor_flops"] / 1e9, 3),
                               "factor_gflop_dense_equivalent": round(info["dense_factor_flops"] / 1e9, 3),
                               "store_mb": round(info["reduced_store_bytes"] / 1e6, 2),
                               "store_note": "device bytes of S | v as the envelope's 64 x 64 tiles (as many again for the factor); "
                                             "a dense (n + 64) n array would be "
                                             f"{(info['matrix_dim'] + 64) * info['matrix_dim'] * 8 / 1e6:.0f} MB"},
            "cpu_baseline": cpu_baseline,
            "scaling_model": scaling_model,
            "ranks": {"world": world, "confirmed_by_all_reduce": ranks_confirmed, "exchange": exchange, "per_rank": ranks,
                      "observations_max_over_mean": round(max(r["observations"] for r in ranks) / max(1.0, float(np.mean([r["observations"] for r in ranks]))), 4),
                      "exchanged_bytes_per_linear_solve": 0 if world == 1 else int(packed_bytes + 8 * (81 * full.num_images + 54 * full.num_cameras) + 8 * 16),
                      "note": "per linear solve every rank all-reduces the packed non-zero tiles of the reduced camera system + its right-hand side, "
                              "the camera-side sums (81 doubles per image, 54 per camera) and two groups of scalars; the speculative evaluation "
                              "of the single-GPU loop is off when sharded (its collective would re-sum a rejected step's sums): ~40 us per iteration"},
            "other_configs": other_configs,
            "traffic_per_iteration": traffic_iter,
            "kernels": table,
            "solve": {"iterations": final["num_successful_steps"] + final["num_unsuccessful_steps"],
                      "termination": final["termination_name"], "rmse_px": round(rmse, 6),
                      "solve_seconds": round(final["solve_seconds"], 4), "setup_seconds": round(final["setup_seconds"], 4),
                      "iterations_per_second_incl_setup": round((final["num_successful_steps"] + final["num_unsuccessful_steps"]) /
                                                                max(final["solve_seconds"] + final["setup_seconds"], 1e-9), 2),
                      "setup_seconds_first_session": None if first_setup is None else round(first_setup, 4),
                      "note": "one complete solve of a freshly created session (the second of the process: pools warm, no event "
                              "brackets): host indexing + upload (setup_seconds) and the LM loop; the set-up-inclusive rate is what "
                              "one bundle_adjustment() call of a running mapper delivers and is never `value`"},
        }
        print(json.dumps(out), flush=True)
    sess.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
