"""Host-only pricing of a multi-GPU split that could pass 2x (round 5). No device needed.

Two questions, answered on the real problems (synthetic C3 / C5 scenes) and the real tile structures
(tests/golden/chol_structure_*.txt) with the timing model of scripts/_dbg/chol_schedule_lab.py:

 (i)  SHARDING. Today a rank owns a contiguous, observation-balanced range of the point order and all-reduces every
      structurally non-zero tile of S. If instead a rank owns the points whose images lie in "its" subtrees of the
      elimination tree (a point's images always lie on ONE root-to-leaf path: leaves are mutually uncoupled), which tiles
      does a rank touch, and which tiles receive contributions from more than one rank (only those need a collective)?
 (ii) SUBTREE-TO-RANK FACTORISATION. Each rank factorises the nodes of its subtrees, applies their updates to the
      separator tiles above, the separator tiles are summed over the ranks (ONE collective that carries the Schur terms and
      the factorisation's updates together), the top of the tree is factorised by every rank redundantly. Modelled forward
      time = slowest rank's local part + exchange + top part.

  python scripts/_dbg/multi_gpu_pricing.py [C3] [C5]        (C5 takes a few minutes: 10 M observations on the host)
"""
import collections
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mavmap_amd import api, synth  # noqa: E402
import chol_schedule_lab as lab    # noqa: E402

TILE_BYTES = 64 * 64 * 8
LINK_GBS = 50.0       # one xGMI link, one direction (VERDICT's figure); a GPU has ONE link to each of its 7 peers, and RCCL's
                      # rings use all the links the participating GPUs share: R - 1 per GPU, capped where large all-reduces saturate
BUS_CAP_GBS = 300.0
COLL_LATENCY_US = 20.0  # launch + completion of one RCCL collective
# measured single-GPU iteration split (profiles/r04_bench_{C3,C5}.json, ms): sharded = per-point work, replicated = the rest
MEASURED = {
    "C3": dict(iteration=0.855, sharded=0.47, factor=0.27, backsolve=0.037, other_replicated=0.078, packed_tiles_mb=19.0),
    "C5": dict(iteration=5.82, sharded=3.30, factor=2.09, backsolve=0.13, other_replicated=0.30, packed_tiles_mb=164.0),
    # round 6: C5 with a depth-3 elimination tree (the default since then; tests/golden/chol_structure_C5d3.txt, dumped with
    # MAVBA_CHOL_DUMP; profiles/r06_nd_depth_sweep.txt: 4.92 ms per iteration, factor 1.53, back-substitution 0.10)
    "C5d3": dict(iteration=4.92, sharded=2.95, factor=1.53, backsolve=0.10, other_replicated=0.34, packed_tiles_mb=134.0),
}
SCENE = {"C5d3": "C5"}    # the synthetic scene a structure belongs to
DEPTH = {"C5d3": 3}       # depth of the elimination tree to ask for (default: the session's rule of rounds 3-5)


def ring_allreduce_us(nbytes, ranks):
    if ranks <= 1 or nbytes <= 0:
        return 0.0
    bw = min(LINK_GBS * (ranks - 1), BUS_CAP_GBS)
    return COLL_LATENCY_US + 2.0 * (ranks - 1) / ranks * nbytes / (bw * 1e3)  # bytes / (GB/s * 1e3) = us


# ---------------------------------------------------------------------------------------------------------------------
# (i) sharding: which tiles does a rank touch
# ---------------------------------------------------------------------------------------------------------------------
def image_tree(p, tree_depth=None):
    """node_of_image, parent[], depth[], column tile of every image / camera block (the session's layout: nodes in elimination
    order, 6 columns per image, the intrinsics at the end of the root, every node padded to whole 64-column tiles)."""
    NI, NC = p.num_images, p.num_cameras
    inc = sp.csr_matrix((np.ones(p.num_obs, np.int8), (p.obs_point, p.obs_image)), shape=(p.num_points, NI))
    adj = (inc.T @ inc).tocoo()
    m = adj.row < adj.col
    pairs = np.stack([adj.row[m], adj.col[m]], axis=1).astype(np.int32)
    tiles0 = (6 * NI + 9 * NC + 63) // 64
    node, parent = api.elimination_tree(NI, NC, pairs, max_depth=tree_depth if tree_depth else (3 if tiles0 <= 96 else 2))
    nn = len(parent)
    depth = np.zeros(nn, int)
    for n in range(nn - 1, -1, -1):   # parents come after their children
        depth[n] = 0 if parent[n] < 0 else depth[parent[n]] + 1
    col = 0
    img_tile = np.zeros(NI, int)
    node_tiles = []
    for n in range(nn):
        begin = col
        for i in np.nonzero(node == n)[0]:
            img_tile[i] = col // 64   # (a 6-column block may straddle two tiles: the first one is what we count)
            col += 6
        if n == nn - 1:
            cam_tile = np.array([(col + 9 * c) // 64 for c in range(NC)])
            col += 9 * NC
        col = max((col + 63) // 64 * 64, begin + 64)
        node_tiles.append((begin // 64, col // 64))
    return node, parent, depth, img_tile, cam_tile, node_tiles, inc


def touched_tiles(inc_rows, img_tile, cam_tile, img_cam, nbt):
    """boolean nbt x nbt (lower) of the tiles that the points in inc_rows (a CSR points x images incidence) contribute to"""
    NI = len(img_tile)
    # points x tiles incidence: the tiles of the point's images and of their cameras' intrinsics blocks
    coo = inc_rows.tocoo()
    rows = np.concatenate([coo.row, coo.row])
    cols = np.concatenate([img_tile[coo.col], cam_tile[img_cam[coo.col]]])
    pt = sp.csr_matrix((np.ones(len(rows), np.int32), (rows, cols)), shape=(inc_rows.shape[0], nbt))
    pt.data[:] = 1
    t = (pt.T @ pt).toarray() > 0
    return np.tril(t | t.T)


def sharding_report(name, p, ranks_list):
    node, parent, depth, img_tile, cam_tile, node_tiles, inc = image_tree(p, DEPTH.get(name))
    nn = len(parent)
    nbt = node_tiles[-1][1]
    img_cam = np.asarray(p.image_camera)
    print(f"[{name}] elimination tree: {nn} nodes, {nbt} tile columns; nodes (tiles, parent, depth): "
          + " ".join(f"{e - b}/{parent[n]}/{depth[n]}" for n, (b, e) in enumerate(node_tiles)))
    # every point's deepest node
    coo = inc.tocoo()
    pt_depth = np.zeros(p.num_points, int)
    pt_node = np.full(p.num_points, nn - 1)
    order = np.argsort(depth[node[coo.col]], kind="stable")
    pt_node[coo.row[order]] = node[coo.col[order]]       # the last write per point is its deepest image's node
    pt_depth = depth[pt_node]
    obs_per_pt = np.asarray(inc.sum(axis=1)).ravel()
    all_touched = touched_tiles(inc, img_tile, cam_tile, img_cam, nbt)
    print(f"[{name}] structurally non-zero lower tiles (this count, from tiles of the blocks' first columns): {int(all_touched.sum())}")
    out = {}
    for R in ranks_list:
        # --- today: contiguous, observation-balanced ranges of the point order (the order groups points by image set; here: by deepest node, then index)
        porder = np.lexsort((np.arange(p.num_points), pt_node))
        cum = np.cumsum(obs_per_pt[porder])
        bounds = np.searchsorted(cum, cum[-1] * np.arange(1, R) / R)
        today = np.zeros(p.num_points, int)
        today[porder] = np.searchsorted(bounds, np.arange(p.num_points), side="right")
        # --- leaf-aligned: cut the tree where it has at least R subtrees, deal the subtrees to ranks by observation count
        cut = 0
        while cut < depth.max() and np.sum(depth == cut) < R:
            cut += 1
        roots = [n for n in range(nn) if depth[n] == cut or (depth[n] < cut and not np.any(parent == n))]
        anc = np.arange(nn)                      # every node's ancestor at the cut level (-1: above the cut)
        for n in range(nn - 1, -1, -1):
            if depth[n] > cut:
                anc[n] = anc[parent[n]]
            elif depth[n] < cut and n not in roots:
                anc[n] = -1
        sub_obs = collections.Counter()
        for n in range(nn):
            if anc[n] >= 0:
                sub_obs[anc[n]] += int(obs_per_pt[pt_node == n].sum())
        load = np.zeros(R)
        rank_of_sub = {}
        for s, w in sorted(sub_obs.items(), key=lambda kv: -kv[1]):
            r = int(np.argmin(load)); rank_of_sub[s] = r; load[r] += w
        leafy = np.full(p.num_points, -1)
        for n in range(nn):
            if anc[n] >= 0:
                leafy[pt_node == n] = rank_of_sub[anc[n]]
        top_pts = np.nonzero(leafy < 0)[0]       # points that only see separator images above the cut: anywhere (balance)
        for q in top_pts:
            r = int(np.argmin(load)); leafy[q] = r; load[r] += obs_per_pt[q]
        res = {}
        for label, assign in (("contiguous ranges (today)", today), ("subtree-aligned", leafy)):
            count = np.zeros((nbt, nbt), int)
            per_rank, obs_r = [], []
            for r in range(R):
                sel = np.nonzero(assign == r)[0]
                t = touched_tiles(inc[sel], img_tile, cam_tile, img_cam, nbt)
                count += t
                per_rank.append(int(t.sum())); obs_r.append(int(obs_per_pt[sel].sum()))
            shared = int((count > 1).sum())
            res[label] = dict(tiles_per_rank=per_rank, shared_tiles=shared, obs_imbalance=max(obs_r) / (sum(obs_r) / R))
            print(f"[{name}] R={R} {label:28s}: tiles touched per rank {min(per_rank)}-{max(per_rank)} of {int(all_touched.sum())}, "
                  f"tiles with more than one contributor {shared} = {shared * TILE_BYTES / 1e6:.1f} MB, "
                  f"largest rank's observations {res[label]['obs_imbalance']:.2f} x the mean")
        out[R] = res
    return out


# ---------------------------------------------------------------------------------------------------------------------
# (ii) subtree-to-rank factorisation on the real tile structure
# ---------------------------------------------------------------------------------------------------------------------
class SubModel(lab.Model):
    """The lab's model restricted to a set of active tile columns: the chain only walks those, inactive columns are 'done at
    time 0' (their tiles count as published), tasks are whatever the caller passes."""

    def restrict(self, active, tasks, helpers):
        self.active = set(active); self.tasks = tasks; self.helpers = max(1, helpers)

    def chain_begin(self, T, j):
        if j not in self.active:
            T["begin"][j] = 0.0
            return
        n = self.seg[j]; b = self.nodes[n][0]; first = j == b
        kids = [c for c in self.children[n] if self.nodes[c][1] - 1 in self.active]
        start = max([T["fin"][self.nodes[c][1] - 1] for c in kids] + [0.0]) if first else T["fin"][j - 1]
        start = max(start, T["pre"].get(2 * j, 0.0), T["pre"].get(2 * j + 1, 0.0))
        T["begin"][j] = start
        if not first: T["L"][(j, j - 1)] = start + lab.cSub

    def chain_end(self, T, j):
        if j not in self.active:
            T["fin"][j] = 0.0
            return
        T["fin"][j] = T["begin"][j] + (lab.cColFirst if j == self.nodes[self.seg[j]][0] else lab.cCol)

    def forward(self):
        I = self.ideal()
        vals = [v for j, v in I["fin"].items() if j in self.active] + [I["L"][(i, j)] for (k, i, j, u) in self.tasks if k == 0]
        ideal = max(vals + [0.0])
        fwd = self.schedule(I)[0] if self.tasks else ideal
        return fwd, ideal


def factor_report(name, ranks_list, helpers_per_gpu=234):
    base = SubModel(name)
    nb, nodes = base.nb, base.nodes
    nn = len(nodes)
    depth = [0] * nn
    for n in range(nn - 1, -1, -1):
        depth[n] = 0 if nodes[n][2] < 0 else depth[nodes[n][2]] + 1
    all_tasks = list(base.tasks)
    base.restrict(range(nb), all_tasks, base.helpers)
    full_fwd, full_ideal = base.forward()
    print(f"[{name}] one GPU, whole tree: modelled forward {full_fwd:.0f} us ({full_ideal:.0f} on unlimited helpers), "
          f"{len(all_tasks)} helper tasks, {sum(len(t[3]) for t in all_tasks)} tile updates")
    res = {1: dict(forward=full_fwd)}
    for R in ranks_list:
        cut = 0
        while cut < max(depth) and sum(d == cut for d in depth) < R:
            cut += 1
        anc = list(range(nn))
        for n in range(nn - 1, -1, -1):
            if depth[n] > cut: anc[n] = anc[nodes[n][2]]
            elif depth[n] < cut: anc[n] = -1
        subs = sorted({a for a in anc if a >= 0})
        # a leaf above the cut level is a subtree of its own
        for n in range(nn):
            if anc[n] < 0 and not any(nodes[c][2] == n for c in range(nn)):
                anc[n] = n; subs.append(n)
        col_sub = {}
        for n in range(nn):
            for c in range(nodes[n][0], nodes[n][1]): col_sub[c] = anc[n]
        work = collections.Counter()
        for (k, i, j, u) in all_tasks:
            if col_sub[j] >= 0: work[col_sub[j]] += len(u) + 1
        load = [0.0] * R; rank_of = {}
        for s in sorted(subs, key=lambda s: -work[s]):
            r = load.index(min(load)); rank_of[s] = r; load[r] += work[s] + 1
        top_cols = [c for c in range(nb) if col_sub[c] < 0]
        # local parts
        local = []
        top_partial_tiles = set()
        for r in range(R):
            cols = [c for c in range(nb) if col_sub[c] >= 0 and rank_of[col_sub[c]] == r]
            cs = set(cols)
            pure = [t for t in all_tasks if t[2] in cs]
            # this rank's share of the updates of the top tiles (accumulate + store, no solve): priced as helper work beside
            # the local factorisation
            extra_work = 0.0
            for (k, i, j, u) in all_tasks:
                if j in cs: continue
                mine = sum(1 for x in u if x in cs)
                if mine:
                    extra_work += mine * lab.cU + lab.cP
                    top_partial_tiles.add((i, j))
            m = SubModel(name); m.restrict(cols, pure, helpers_per_gpu)
            fwd, ideal = m.forward() if cols else (0.0, 0.0)
            # what does not fit under the local factorisation extends the local part (total helper work / helpers is a lower
            # bound), and the updates by the subtree's LAST column can only start when it is factorised: one more update + store
            fwd = max(fwd, ideal) + lab.cU + lab.cP
            spill = max(0.0, (sum(len(t[3]) * lab.cU + (lab.cS if t[0] == 0 else lab.cP) for t in pure) + extra_work) / helpers_per_gpu - fwd)
            local.append(fwd + spill)
        # the exchange: every top tile that a local column updates or that the sharded assembly fills = all top x top tiles (lower)
        ts = set(top_cols)
        top_tiles = len({(i, j) for (k, i, j, u) in all_tasks if j in ts and i in ts} | {(c, c) for c in top_cols})
        exch = ring_allreduce_us(top_tiles * TILE_BYTES, R)
        # the top part, factorised by every rank (its updates from below have arrived with the exchange)
        top_tasks = [(k, i, j, [x for x in u if x in ts]) for (k, i, j, u) in all_tasks if j in ts]
        m = SubModel(name); m.restrict(top_cols, top_tasks, helpers_per_gpu)
        top_fwd, top_ideal = m.forward() if top_cols else (0.0, 0.0)
        res[R] = dict(cut=cut, subtrees=len(subs), local=max(local), local_all=local, top_cols=len(top_cols), top_tiles=top_tiles,
                      exchange=exch, top=top_fwd, forward=max(local) + exch + top_fwd)
        print(f"[{name}] R={R}: cut below depth {cut} ({len(subs)} subtrees), local part {max(local):.0f} us (ranks: "
              + " ".join(f"{x:.0f}" for x in local) + f"), exchange of {top_tiles} top tiles = {top_tiles * TILE_BYTES / 1e6:.1f} MB: {exch:.0f} us, "
              f"top part ({len(top_cols)} columns) {top_fwd:.0f} us -> forward {res[R]['forward']:.0f} us (one GPU: {full_fwd:.0f})")
    return res


def iteration_table(name, shard, fact, ranks_list):
    M = MEASURED[name]
    one = M["iteration"]
    rep = M["factor"] + M["backsolve"] + M["other_replicated"]
    print(f"[{name}] modelled LM iteration (ms) from the measured one-GPU split (sharded {M['sharded']}, factor {M['factor']}, "
          f"other replicated {M['backsolve'] + M['other_replicated']:.3f}; all-reduce: {COLL_LATENCY_US:.0f} us + 2 (R-1)/R bytes / min({LINK_GBS:.0f} (R-1), {BUS_CAP_GBS:.0f}) GB/s):")
    print(f"[{name}]   ranks | (A) today: contiguous shards, every tile all-reduced, replicated solve | (B) subtree-aligned shards: shared tiles summed, the others gathered, "
          f"replicated solve | (C) B + subtree-to-rank factorisation")
    f1 = fact[1]["forward"]
    scale = M["factor"] / (f1 / 1e3)      # the model's one-GPU forward time against the measured factor kernel
    for R in ranks_list:
        small = 2 * COLL_LATENCY_US / 1e3   # the camera sums and the scalars: two more collectives per iteration
        a = M["sharded"] / R + rep + ring_allreduce_us(M["packed_tiles_mb"] * 1e6, R) / 1e3 + small
        imb = shard[R]["subtree-aligned"]["obs_imbalance"] if shard else 1.0
        shared = shard[R]["subtree-aligned"]["shared_tiles"] * TILE_BYTES if shard else fact[R]["top_tiles"] * TILE_BYTES
        # (B): a REPLICATED solve needs every tile on every rank: the tiles with one contributor still travel once (all-gather,
        # (R-1)/R of their bytes per rank), only the shared ones are summed (all-reduce, twice that)
        excl = max(0.0, M["packed_tiles_mb"] * 1e6 - shared)
        bw = min(LINK_GBS * (R - 1), BUS_CAP_GBS) * 1e3
        b = M["sharded"] / R * imb + rep + (ring_allreduce_us(shared, R) + (R - 1) / R * excl / bw) / 1e3 + small
        cf = fact[R]["forward"] / 1e3 * scale
        c = M["sharded"] / R * imb + cf + M["backsolve"] + M["other_replicated"] + small
        print(f"[{name}]   {R:5d} | {a:6.3f} ms {one / a:4.2f}x | {b:6.3f} ms {one / b:4.2f}x (shared tiles {shared / 1e6:.1f} MB, imbalance {imb:.2f}) | "
              f"{c:6.3f} ms {one / c:4.2f}x (factor incl. its exchange {cf:.3f})")
    print(f"[{name}]   bound of (A)/(B) with a free exchange and perfect balance: {one / rep:.2f}x (the replicated part is {rep:.3f} of {one} ms)")


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in MEASURED] or ["C3"]
    ranks = (2, 4, 8)
    for name in names:
        t0 = time.time()
        shard = None
        if "--no-shards" not in sys.argv:
            p = synth.make_config(SCENE.get(name, name))
            print(f"[{name}] scene: {p.num_images} images, {p.num_points} points, {p.num_obs} observations ({time.time() - t0:.0f} s)")
            shard = sharding_report(name, p, ranks)
        fact = factor_report(name, ranks)
        iteration_table(name, shard, fact, ranks)
        print(f"[{name}] done in {time.time() - t0:.0f} s")
