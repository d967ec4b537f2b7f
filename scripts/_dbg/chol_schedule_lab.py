"""Python copy of the persistent factorisation's timing model on the real tile structures (tests/golden/chol_structure_*.txt), for trying
schedule ideas without a device: earliest- vs latest-finish visiting order, pools, split update lists.
  python scripts/_dbg/chol_schedule_lab.py"""
import sys, collections, time
sys.path.insert(0, "/root/repo")
from mavmap_amd import api
cU, cS, cP, cCol, cColFirst, cSub = 3.5, 3.1, 1.0, 12.2, 7.4, 4.0  # (round 5: a node's first column is the systolic tile factorisation)
def load(path):
    rows = [tuple(int(x) for x in l.split()) for l in open(path)]
    nb, nn, npairs = rows[0]
    return nb, rows[1:1 + nn], rows[1 + nn:1 + nn + npairs]
class Model:
    def __init__(self, name):
        self.nb, self.nodes, pairs = load(f"/root/repo/tests/golden/chol_structure_{name}.txt")
        s = api.debug_chol_schedule(self.nb, self.nodes, pairs, cus=256)
        self.info = s["chain_info"]; self.helpers = s["grid"] - s["chain_wgs"]
        self.tasks = [(k, i, j, u) for _, k, i, j, u in s["tasks"] if k != 3]
        self.seg = {}
        for n, (b, e, p) in enumerate(self.nodes):
            for c in range(b, e): self.seg[c] = n
        self.children = collections.defaultdict(list)
        for n, (b, e, p) in enumerate(self.nodes):
            if p >= 0: self.children[p].append(n)
    def in_ready(self, T, i, j, k):
        r = T["L"].get((i, k), 0.0)
        if i != j: r = max(r, T["L"].get((j, k), 0.0))
        return r
    def run_updates(self, T, i, j, upd, start):
        f = start
        for k in upd: f = max(f, self.in_ready(T, i, j, k)) + cU
        return f
    def chain_begin(self, T, j):
        n = self.seg[j]; b = self.nodes[n][0]; first = j == b
        start = max([T["fin"][self.nodes[c][1] - 1] for c in self.children[n]] + [0.0]) if first else T["fin"][j - 1]
        start = max(start, T["pre"].get(2 * j, 0.0), T["pre"].get(2 * j + 1, 0.0))
        T["begin"][j] = start
        if not first: T["L"][(j, j - 1)] = start + cSub
    def chain_end(self, T, j):
        T["fin"][j] = T["begin"][j] + (cColFirst if j == self.nodes[self.seg[j]][0] else cCol)
    def ideal(self):
        T = dict(L={}, pre={}, begin={}, fin={})
        pre_tasks = collections.defaultdict(list); tile_tasks = collections.defaultdict(list)
        for t in self.tasks:
            k, i, j, u = t
            if k == 1: pre_tasks[j].append(t)
            elif k == 2: pre_tasks[i].append(t)
            else: tile_tasks[j].append(t)
        for j in range(self.nb):
            for k, i, jj, u in pre_tasks[j]:
                T["pre"][2 * j if k == 1 else 2 * j + 1] = self.run_updates(T, i, jj, u, 0.0) + cP
            self.chain_begin(T, j); self.chain_end(T, j)
            for k, i, jj, u in tile_tasks[j]:
                T["L"][(i, j)] = max(self.run_updates(T, i, j, u, 0.0), T["fin"][j]) + cS
        return T
    def fin_of(self, T, t):
        k, i, j, u = t
        return T["L"][(i, j)] if k == 0 else T["pre"][2 * j if k == 1 else 2 * i + 1]
    def schedule(self, est, order_key=None, pool=0):
        ev = []
        for g, t in enumerate(self.tasks):
            ev.append(((self.fin_of(est, t) if order_key is None else order_key(t)), 1, g))
        for j in range(self.nb):
            ev.append((est["fin"][j] if order_key is None else order_key(("fin", j)), 0, -1 - j)); ev.append((est["begin"][j] if order_key is None else order_key(("begin", j)), 2, -1 - j))
        ev.sort()
        act = dict(L={}, pre={}, begin={}, fin={})
        free = [0.0] * self.helpers
        work = occ = 0.0
        for tm, kind, g in ev:
            if kind == 2: self.chain_begin(act, -1 - g); continue
            if kind == 0: self.chain_end(act, -1 - g); continue
            t = self.tasks[g]; k, i, j, u = t
            release = max(0.0, self.run_updates(est, i, j, u, 0.0) - cU * len(u) - 10.0)
            w0, w1 = (pool, self.helpers) if (pool and k == 0) else ((0, pool) if pool else (0, self.helpers))
            best = None; ff = w0
            for w in range(w0, w1):
                if free[w] < free[ff]: ff = w
                if free[w] <= release and (best is None or free[w] > free[best]): best = w
            if best is None: best = ff
            f = self.run_updates(act, i, j, u, free[best])
            if k == 0:
                f = max(f, act["fin"][j]) + cS; act["L"][(i, j)] = f
            else:
                f += cP; act["pre"][2 * j if k == 1 else 2 * i + 1] = f
            work += cU * len(u) + (cS if k == 0 else cP); occ += f - free[best]
            free[best] = f
        fwd = max(max(act["fin"].values()), max(act["L"].values()))
        return fwd, work, occ, act
if __name__ == "__main__":
    for name in ("C3", "C5"):
        m = Model(name); t0 = time.time(); I = m.ideal()
        print(name, "ideal forward", max(max(I["fin"].values()), max(I["L"].values())), "helpers", m.helpers)
        print("  EF order, shared pool:", m.schedule(I)[:3], "%.1fs" % (time.time() - t0))

def latest_finish(m, I):
    """ALAP times on the unlimited-helpers DAG: latest finish of every helper task / chain event so that the end E is kept."""
    E = max(max(I["fin"].values()), max(I["L"].values()))
    INF = float("inf")
    need = collections.defaultdict(lambda: INF)   # ('L',i,k) / ('d',j) / ('p',slot) -> latest ready time
    LF = {}
    # order: reverse ideal finish of tasks and chain events
    items = [(m.fin_of(I, t), 1, g) for g, t in enumerate(m.tasks)] + [(I["fin"][j], 0, -1 - j) for j in range(m.nb)] + [(I["begin"][j], 2, -1 - j) for j in range(m.nb)]
    items.sort(reverse=True)
    lf_begin = {}; lf_fin = {}
    for tm, kind, g in items:
        if kind == 1:
            k, i, j, u = m.tasks[g]
            out = ('L', i, j) if k == 0 else ('p', 2 * j if k == 1 else 2 * i + 1)
            lf = min(need[out], E)
            LF[g] = lf
            tail = cS if k == 0 else cP
            if k == 0: need[('d', j)] = min(need[('d', j)], lf - cS)
            n = len(u)
            for q, kk in enumerate(u):
                ls = lf - tail - (n - q) * cU
                need[('L', i, kk)] = min(need[('L', i, kk)], ls)
                if i != j: need[('L', j, kk)] = min(need[('L', j, kk)], ls)
        elif kind == 0:   # column end
            j = -1 - g; n = m.seg[j]; b, e, p = m.nodes[n]
            lf = min(need[('d', j)], E)
            if j + 1 < e: lf = min(lf, lf_begin[j + 1])
            elif p >= 0: lf = min(lf, lf_begin[m.nodes[p][0]])
            lf_fin[j] = lf
        else:             # column begin
            j = -1 - g; first = j == m.nodes[m.seg[j]][0]
            lb = lf_fin[j] - (cColFirst if first else cCol)
            if not first: lb = min(lb, need[('L', j, j - 1)] - cSub)
            lf_begin[j] = lb
            need[('p', 2 * j)] = min(need[('p', 2 * j)], lb); need[('p', 2 * j + 1)] = min(need[('p', 2 * j + 1)], lb)
    return LF, lf_begin, lf_fin

if __name__ == "__main__":
    for name in ("C3", "C5"):
        m = Model(name); I = m.ideal()
        LF, lb, lfin = latest_finish(m, I)
        gidx = {id(t): g for g, t in enumerate(m.tasks)}
        def key(x):
            if isinstance(x, tuple) and x and x[0] == "fin": return lfin[x[1]]
            if isinstance(x, tuple) and x and x[0] == "begin": return lb[x[1]]
            return LF[gidx[id(x)]]
        for pool in (0, 32):
            print(name, "LF order pool", pool, m.schedule(I, order_key=key, pool=pool)[:3])

def split_model(m, thresh, parts=2):
    """What if a task with more than `thresh` updates were shared by `parts` work-groups (each applies a slice of the list into its
    own accumulator; the owner adds the partial tiles - one extra tile load + add per helper part, priced as one update)?"""
    new_tasks = []
    for (k, i, j, u) in m.tasks:
        if len(u) <= thresh:
            new_tasks.append((k, i, j, u, None)); continue
        # helper parts: kind 9 = partial accumulate publishing ('X', i, j, part); owner waits for them
        sl = [u[q::parts] for q in range(parts)]
        for q in range(1, parts): new_tasks.append((9, i, j, sl[q], (k, q)))
        new_tasks.append((k, i, j, sl[0], ('own', parts - 1)))
    return new_tasks

def schedule_split(m, I, tasks, keyfun):
    # generic list scheduling on the extended task set; partial tasks publish ('X', kind, i, j, q); owners add (parts-1)*cU after their own list
    ev = []
    for g, t in enumerate(tasks): ev.append((keyfun(t), 1, g))
    for j in range(m.nb): ev.append((keyfun(("fin", j)), 0, -1 - j)); ev.append((keyfun(("begin", j)), 2, -1 - j))
    ev.sort()
    act = dict(L={}, pre={}, begin={}, fin={}); X = {}
    free = [0.0] * m.helpers
    for tm, kind, g in ev:
        if kind == 2: m.chain_begin(act, -1 - g); continue
        if kind == 0: m.chain_end(act, -1 - g); continue
        k, i, j, u, extra = tasks[g]
        rel = max(0.0, m.run_updates(I, i, j, u, 0.0) - cU * len(u) - 10.0)
        best = None; ff = 0
        for w in range(m.helpers):
            if free[w] < free[ff]: ff = w
            if free[w] <= rel and (best is None or free[w] > free[best]): best = w
        if best is None: best = ff
        f = m.run_updates(act, i, j, u, free[best])
        if k == 9:
            f += cP; X[(extra[0], i, j, extra[1])] = f
        else:
            if extra is not None:
                for q in range(1, extra[1] + 1): f = max(f, X[(k, i, j, q)]) + cU
            if k == 0: f = max(f, act["fin"][j]) + cS; act["L"][(i, j)] = f
            else: f += cP; act["pre"][2 * j if k == 1 else 2 * i + 1] = f
        free[best] = f
    return max(max(act["fin"].values()), max(act["L"].values()))

if __name__ == "__main__":
    m = Model("C5"); I = m.ideal()
    LF, lb, lfin = latest_finish(m, I)
    lfmap = {(t[0], t[1], t[2]): LF[g] for g, t in enumerate(m.tasks)}
    def keyfun(x):
        if x[0] == "fin": return lfin[x[1]]
        if x[0] == "begin": return lb[x[1]]
        k, i, j = x[0], x[1], x[2]
        if k == 9: return lfmap[(x[4][0], i, j)] - cU - 0.001 * x[4][1]   # a partial has to be there before its owner's end
        return lfmap[(k, i, j)]
    import statistics
    ns = [len(t[3]) for t in m.tasks]
    print("updates per task: median", statistics.median(ns), "max", max(ns), "tasks > 40:", sum(n > 40 for n in ns), "> 80:", sum(n > 80 for n in ns))
    for thresh, parts in ((10**9, 2), (80, 2), (40, 2), (40, 4), (20, 4)):
        tasks = split_model(m, thresh, parts)
        print("split lists longer than", thresh, "into", parts, "->", round(schedule_split(m, I, tasks, keyfun), 1))
