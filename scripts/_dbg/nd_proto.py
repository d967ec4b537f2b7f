"""Prototype of the elimination-tree choice (mavba_session::choose_elimination_order) on the image graph of a synthetic
config: replicates the C++ recursion (cut of the acquisition order / BFS level structure) and tries alternatives.
Debug harness only."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from mavmap_amd import synth

def image_graph(p):
    order = np.argsort(p.obs_point, kind="stable")
    pts = p.obs_point[order]; img = p.obs_image[order]
    start = np.flatnonzero(np.r_[True, pts[1:] != pts[:-1], True])
    adj = [set() for _ in range(p.num_images)]
    seen = set()
    for a, b in zip(start[:-1], start[1:]):
        key = tuple(sorted(set(img[a:b].tolist())))
        if key in seen: continue
        seen.add(key)
        for x in key:
            adj[x].update(key)
    for i in range(p.num_images): adj[i].discard(i)
    return [sorted(a) for a in adj]

def tiles_of(cols): return (cols + 63) // 64

def run(adj, NI, tail, max_depth=3, use_levels=True, verbose=False, min_leaf_tiles=6, min_n=32, extra=None):
    lower = [[c for c in adj[r] if c < r] for r in range(NI)]
    tn = []  # (imgs, parent)
    pos = [-1] * NI; level = [-1] * NI
    def rec(M, depth, tl):
        n = len(M); leaf_tiles = tiles_of(6 * n + tl)
        def make_leaf():
            tn.append([M, -1]); return len(tn) - 1, leaf_tiles
        if depth >= max_depth or n < min_n or leaf_tiles < min_leaf_tiles: return make_leaf()
        for t, v in enumerate(M): pos[v] = t
        mnp = []
        for t, v in enumerate(M):
            m = t
            for c in lower[v]:
                if pos[c] >= 0: m = min(m, pos[c])
            mnp.append(m)
        best, best_c = leaf_tiles, -1
        for c in range(n // 4, 3 * n // 4 + 1, max(1, n // 64)):
            ns = sum(1 for t in range(c, n) if mnp[t] < c)
            if ns == 0 and tl == 0: continue
            est = max(tiles_of(6 * c), tiles_of(6 * (n - c - ns))) + tiles_of(6 * ns + tl)
            if est < best: best, best_c = est, c
        best_lv, best_m = leaf_tiles, -1
        if use_levels:
            def bfs(start):
                for v in M: level[v] = -1
                queue = []; head = 0; seed = start; base = 0; scan = 0
                while True:
                    level[seed] = base; queue.append(seed)
                    while head < len(queue):
                        v = queue[head]; head += 1
                        for w in adj[v]:
                            if pos[w] >= 0 and level[w] < 0: level[w] = level[v] + 1; queue.append(w)
                    if len(queue) == n: break
                    base = level[queue[-1]] + 2
                    while level[M[scan]] >= 0: scan += 1
                    seed = M[scan]
                return queue
            start = M[0]
            for _ in range(2):
                q = bfs(start); last = level[q[-1]]; pick = -1
                for v in q:
                    if level[v] == last and (pick < 0 or len(adj[v]) < len(adj[pick]) or (len(adj[v]) == len(adj[pick]) and v < pick)): pick = v
                start = pick
            bfs(start)
            nlev = max(level[v] for v in M) + 1
            lsize = [0] * nlev; lsep = [0] * nlev
            for v in M:
                lsize[level[v]] += 1
                if any(pos[w] >= 0 and level[w] == level[v] + 1 for w in adj[v]): lsep[level[v]] += 1
            before = 0
            for m in range(nlev):
                ns = lsep[m]; na = before + lsize[m] - ns; nbh = n - before - lsize[m]
                before += lsize[m]
                if na < 8 or nbh < 8 or (ns == 0 and tl == 0): continue
                est = max(tiles_of(6 * na), tiles_of(6 * nbh)) + tiles_of(6 * ns + tl)
                if est < best_lv: best_lv, best_m = est, m
        A, B, S = [], [], []
        if best_m >= 0 and best_lv < best:
            for v in M:
                if level[v] < best_m: A.append(v)
                elif level[v] > best_m: B.append(v)
                else:
                    fwd = any(pos[w] >= 0 and level[w] == best_m + 1 for w in adj[v])
                    (S if fwd else A).append(v)
            best = best_lv
        elif best_c >= 0:
            A = M[:best_c]
            for t in range(best_c, n): (S if mnp[t] < best_c else B).append(M[t])
        if extra is not None:
            cand = extra(M, adj, pos, tl)
            if cand is not None and cand[0] < best:
                best, A, B, S = cand
        for v in M: pos[v] = -1
        if not A or best > leaf_tiles - max(2, leaf_tiles // 8): return make_leaf()
        if len(A) < 8 or len(B) < 8: return make_leaf()
        ra = rec(A, depth + 1, 0); rb = rec(B, depth + 1, 0)
        sep_tiles = tiles_of(6 * len(S) + tl)
        tn.append([S, -1]); me = len(tn) - 1
        tn[ra[0]][1] = me; tn[rb[0]][1] = me
        return me, max(ra[1], rb[1]) + sep_tiles
    root = rec(list(range(NI)), 0, tail)
    # heights + level-synchronous chain, and the longest root-to-leaf path
    nt = len(tn)
    tl_of = [tiles_of(6 * len(tn[t][0]) + (tail if t == nt - 1 else 0)) for t in range(nt)]
    tl_of = [max(x, 1) for x in tl_of]
    height = [0] * nt
    for t in range(nt):
        if tn[t][1] >= 0: height[tn[t][1]] = max(height[tn[t][1]], height[t] + 1)
    H = max(height)
    chain = sum(max(tl_of[t] for t in range(nt) if height[t] == h) for h in range(H + 1))
    path = [0] * nt
    for t in range(nt - 1, -1, -1):
        par = tn[t][1]
        path[t] = tl_of[t] + (path[par] if par >= 0 else 0)
    longest = max(path[t] for t in range(nt) if height[t] == 0)
    if verbose:
        for t in range(nt): print(t, "imgs", len(tn[t][0]), "tiles", tl_of[t], "parent", tn[t][1], "height", height[t])
    return chain, longest, sum(tl_of), nt

if __name__ == "__main__":
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    p = synth.make_config(cfg)
    adj = image_graph(p)
    NI = p.num_images; tail = 9 * p.num_cameras
    print("images", NI, "avg degree", np.mean([len(a) for a in adj]))
    for d in (2, 3, 4, 5):
        print("depth", d, run(adj, NI, tail, max_depth=d, verbose=(d == 3)))


def cxx_tree(p, adj, max_depth=3):
    """The C++ choice (mavba_debug_elimination_tree) summarised like run()."""
    import mavmap_amd
    from mavmap_amd import api
    api.load()
    pairs = np.array([(i, j) for i in range(len(adj)) for j in adj[i] if j < i], np.int32)
    node, parent = api.elimination_tree(p.num_images, p.num_cameras, pairs, max_depth)
    nt = len(parent)
    if nt == 0: return None
    tail = 9 * p.num_cameras
    cnt = np.bincount(node, minlength=nt)
    tl = [max(1, tiles_of(6 * int(cnt[t]) + (tail if t == nt - 1 else 0))) for t in range(nt)]
    height = [0] * nt
    for t in range(nt):
        if parent[t] >= 0: height[parent[t]] = max(height[parent[t]], height[t] + 1)
    H = max(height)
    chain = sum(max(tl[t] for t in range(nt) if height[t] == h) for h in range(H + 1))
    path = [0] * nt
    for t in range(nt - 1, -1, -1):
        path[t] = tl[t] + (path[parent[t]] if parent[t] >= 0 else 0)
    return chain, max(path[t] for t in range(nt) if height[t] == 0), sum(tl), nt, [int(c) for c in cnt], tl, list(parent)
