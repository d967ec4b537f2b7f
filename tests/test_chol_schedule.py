"""The persistent factorisation's static schedule, checked WITHOUT a GPU (host logic of csrc/dense_chol.hip).

CholStructure::build orders every tile's updates and fills the helper work-groups' queues from a timing model of the launch
(DESIGN.md section 4). The launch cannot dead-lock only if every wait of a queued task is for something that some work-group
produces without waiting for that task in turn. mavba_debug_chol_schedule returns the queues for a tile structure; the replay
below runs them the way k_chol_persist does - a work-group walks its queue in order and blocks on flags - and must reach the end
whatever the structure. (Flags are only ever set and every work-group is sequential, so one replay decides it for all timings.)
"""
import numpy as np
import pytest

import mavmap_amd
from mavmap_amd import api

TILE, PRE_DIAG, PRE_SUB, CHAIN = 0, 1, 2, 3


def _random_tree(rng, depth):
    """Nested dissection shaped tree: [(begin, end, parent)], children before parents, columns contiguous."""
    nodes = []

    def build(d, at):
        if d == 0 or rng.random() < 0.25:
            w = int(rng.integers(1, 5))
            nodes.append([at, at + w, -1])
            return len(nodes) - 1, at + w
        left, at = build(d - 1, at)
        right, at = build(d - 1, at)
        w = int(rng.integers(1, 4))
        nodes.append([at, at + w, -1])
        me = len(nodes) - 1
        nodes[left][2] = me
        nodes[right][2] = me
        return me, at + w

    _, nb = build(depth, 0)
    return nodes, nb


def _random_pairs(rng, nodes, nb):
    """Non-zero lower tiles: a skyline inside every node, ancestors' rows coupled to (some of) their descendants."""
    seg = np.zeros(nb, int)
    for n, (b, e, _) in enumerate(nodes):
        seg[b:e] = n

    def ancestors(n):
        out = []
        while nodes[n][2] >= 0:
            n = nodes[n][2]
            out.append(n)
        return out

    pairs = []
    for n, (b, e, _) in enumerate(nodes):
        for i in range(b, e):
            pairs.append((i, int(rng.integers(b, i + 1))))  # envelope of row i inside its own node
            pairs.append((i, i))
        for a in ancestors(n):
            for i in range(nodes[a][0], nodes[a][1]):
                if rng.random() < 0.7:
                    pairs.append((i, int(rng.integers(b, e))))
    return pairs


def _replay(sched, nb):
    """Runs the queues to a fixed point; returns (finished, what every work-group is stuck on)."""
    queues = {}
    for wg, kind, i, j, upd in sched["tasks"]:
        queues.setdefault(wg, []).append((kind, i, j, upd))
    info = sched["chain_info"]
    L, dflag, pflag = set(), set(), set()  # published factor tiles (i, k), column inverses, PRE slots
    produced_twice = []

    def publish(store, key):
        if key in store:
            produced_twice.append(key)
        store.add(key)

    # a work-group's program as a list of steps: ("wait", [(store, key), ...]) / ("publish", store, key)
    programs = {}
    for wg, tasks in queues.items():
        prog = []
        for kind, i, j, upd in tasks:
            if kind == CHAIN:
                for c in range(i, j):
                    waits = []
                    if c > i and (info[c] & 2):
                        waits.append((pflag, 2 * c + 1))
                    if info[c] & 1:
                        waits.append((pflag, 2 * c))
                    prog.append(("wait", waits))
                    if c > i:
                        prog.append(("publish", L, (c, c - 1)))
                    prog.append(("publish", dflag, c))
                continue
            for k in upd:
                prog.append(("wait", [(L, (i, k))] + ([(L, (j, k))] if i != j else [])))
            if kind == TILE:
                prog.append(("wait", [(dflag, j)]))
                prog.append(("publish", L, (i, j)))
            elif kind == PRE_DIAG:
                prog.append(("publish", pflag, 2 * j))
            else:
                prog.append(("publish", pflag, 2 * i + 1))
        programs[wg] = prog
    at = {wg: 0 for wg in programs}
    progress = True
    while progress:
        progress = False
        for wg, prog in programs.items():
            while at[wg] < len(prog):
                step = prog[at[wg]]
                if step[0] == "wait":
                    if not all(key in store for store, key in step[1]):
                        break
                else:
                    publish(step[1], step[2])
                at[wg] += 1
                progress = True
    stuck = {wg: programs[wg][at[wg]] for wg in programs if at[wg] < len(programs[wg])}
    return not stuck, stuck, produced_twice, L, dflag


@pytest.mark.parametrize("seed", range(40))
def test_random_structures_give_queues_that_run_to_the_end(seed):
    rng = np.random.default_rng(seed)
    nodes, nb = _random_tree(rng, depth=int(rng.integers(1, 5)))
    pairs = _random_pairs(rng, nodes, nb)
    cus = int(rng.choice([8, 16, 64, 256]))
    s = api.debug_chol_schedule(nb, nodes, pairs, cus=cus)
    assert s["nodes"] == len(nodes), "the tree was not accepted"
    if not s["ok"]:
        pytest.skip("no persistent schedule for this structure on so few work-groups")
    assert s["grid"] <= cus and s["model_forward_us"] > 0.0
    done, stuck, twice, L, dflag = _replay(s, nb)
    assert done, f"dead-lock: {list(stuck.items())[:4]}"
    assert not twice, twice[:4]
    assert dflag == set(range(nb))
    # every tile some task multiplies with was produced, the right-hand-side row of every column among them
    assert all((nb, k) in L for k in range(nb))
    # a task's updates are columns left of its tile, each once
    for wg, kind, i, j, upd in s["tasks"]:
        if kind != CHAIN:
            assert len(set(upd)) == len(upd) and all(k < j or (kind == PRE_SUB and k < j) for k in upd), (kind, i, j, upd)


def test_a_chain_of_dense_columns_and_the_model_s_arithmetic():
    """One node of six dense tile columns: the chain is the critical path, 7.4 us for the first column (round 5: the systolic
    tile factorisation; 9.6 with the look-ahead variant) and 12.2 us for every further one in the model, plus the last
    column's panel solves (3.1 us)."""
    nb = 6
    s = api.debug_chol_schedule(nb, [(0, nb, -1)], [(i, 0) for i in range(nb)] + [(i, i) for i in range(nb)], cus=256)
    assert s["ok"] and s["chain_wgs"] == 1
    assert abs(s["model_forward_us"] - (7.4 + 5 * 12.2 + 3.1)) < 1e-6
    done, stuck, twice, L, dflag = _replay(s, nb)
    assert done and not twice


def test_the_order_of_a_tile_s_updates_follows_their_availability():
    """Two leaves of different length under one separator: the separator's first diagonal tile takes the updates of the SHORT
    leaf's columns before the long leaf's last ones (rounds 2-3 interleaved them by position in the node)."""
    nodes = [(0, 2, 2), (2, 8, 2), (8, 10, -1)]
    nb = 10
    pairs = [(i, i) for i in range(nb)] + [(1, 0)] + [(i, 2) for i in range(3, 8)] + [(i, j) for i in (8, 9) for j in range(8)]
    s = api.debug_chol_schedule(nb, nodes, pairs, cus=256)
    assert s["ok"]
    pre = [t for t in s["tasks"] if t[1] == PRE_DIAG and t[2] == 8]
    assert len(pre) == 1
    upd = pre[0][4]
    assert sorted(upd) == list(range(8))
    assert upd.index(0) < upd.index(6) and upd.index(1) < upd.index(7) and upd[-1] == 7


def test_the_replay_notices_a_bad_order():
    """(the checker checked) All helper tasks of a two-level structure on ONE work-group in REVERSE order: the tile tasks of the
    separator come before the leaf tiles they multiply with, and the replay must report the dead-lock."""
    nodes = [(0, 2, 2), (2, 4, 2), (4, 6, -1)]
    nb = 6
    pairs = [(i, j) for i in range(nb) for j in range(i + 1) if (i < 2 and j < 2) or (2 <= i < 4 and 2 <= j < 4) or i >= 4]
    s = api.debug_chol_schedule(nb, nodes, pairs, cus=16)
    assert s["ok"] and _replay(s, nb)[0]
    chains = [t for t in s["tasks"] if t[1] == CHAIN]
    helpers = [t for t in s["tasks"] if t[1] != CHAIN]
    one_wg = max(t[0] for t in s["tasks"]) + 1
    bad = dict(s, tasks=chains + [(one_wg, k, i, j, u) for _, k, i, j, u in reversed(helpers)])
    done, stuck, *_ = _replay(bad, nb)
    assert not done and stuck


def _load_structure(name):
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"chol_structure_{name}.txt")
    rows = [tuple(int(x) for x in line.split()) for line in open(path)]
    nb, nn, npairs = rows[0]
    return nb, rows[1:1 + nn], rows[1 + nn:1 + nn + npairs]


@pytest.mark.parametrize("name,measured_us", [("C2", None), ("C3", 258.5), ("C5", 1955.4), ("C5d3", 1520.8)])
def test_the_schedules_of_the_benchmark_configurations(name, measured_us):
    """The tile structures of C2 / C3 / C5 as the sessions hand them to CholStructure::build on the GPU box (dumped there with
    MAVBA_CHOL_DUMP; tests/golden/chol_structure_*.txt): their queues run to the end, the persistent launch is modelled faster
    than the launch-per-panel schedule, and the model stays near what MAVBA_CHOL_TRACE measured for the forward pass on the
    MI355X (profiles/r05_chol_trace_C3.txt, r04_chol_trace_C5.txt) - the schedule is only as good as that agreement.
    C5d3: C5 with the depth-3 elimination tree that is the default since round 6 (195 tile columns, 15 nodes, 8 concurrent leaves;
    profiles/r06_chol_trace_C5.txt); "C5" is the depth-2 structure of rounds 3-5, kept for the digests below."""
    nb, nodes, pairs = _load_structure(name)
    s = api.debug_chol_schedule(nb, nodes, pairs, cus=256)
    assert s["ok"] and s["nodes"] == len(nodes) and s["grid"] <= 256
    assert s["model_forward_us"] < s["launch_per_panel_us"]
    done, stuck, twice, L, dflag = _replay(s, nb)
    assert done and not twice and dflag == set(range(nb))
    if measured_us:
        # (within 2 % where the chains are the critical path - C3 -, 10 % optimistic where the helpers are - C5: the model does not
        # know about the memory traffic of 252 helpers re-reading tiles at once)
        assert abs(s["model_forward_us"] - measured_us) < 0.12 * measured_us, s["model_forward_us"]


def _schedule_digest_in_a_fresh_process(name, threads):
    """The queues of a benchmark structure built by a process with `threads` host workers (the count is read once per process)."""
    import os
    import subprocess
    import sys
    code = (
        "import hashlib, json, sys\n"
        "sys.path.insert(0, %r)\n"
        "from tests.test_chol_schedule import _load_structure\n"
        "from mavmap_amd import api\n"
        "nb, nodes, pairs = _load_structure(%r)\n"
        "s = api.debug_chol_schedule(nb, nodes, pairs, cus=256)\n"
        "print(hashlib.sha256(json.dumps([s['ok'], s['model_forward_us'], s['grid'], s['tasks'], s['chain_info']]).encode()).hexdigest())\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), name)
    env = dict(os.environ, MAVBA_HOST_THREADS=str(threads))
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True, timeout=300).stdout.strip()


def test_the_schedule_does_not_depend_on_the_number_of_host_threads():
    """C5's structure is large enough for the candidate launches to be simulated on the host workers (one pool size per
    thread); the kept schedule - queues, update orders, modelled time - must be the one a single thread finds."""
    one = _schedule_digest_in_a_fresh_process("C5", 1)
    assert len(one) == 64
    assert _schedule_digest_in_a_fresh_process("C5", 4) == one
    assert _schedule_digest_in_a_fresh_process("C5", 16) == one


def _digest(s):
    import hashlib
    import json
    return hashlib.sha256(json.dumps([s["ok"], s["model_forward_us"], s["launch_per_panel_us"], s["grid"], s["chain_wgs"], s["tiles"],
                                      s["updates"], s["nodes"], s["tasks"], s["chain_info"]]).encode()).hexdigest()[:16]


def test_the_schedules_are_the_ones_of_the_scan_based_builder():
    """Round 5 rebuilt the builder for speed (one visiting order per estimate, helpers in a sorted array, candidate launches on
    host threads). tests/golden/chol_schedule_digests.json holds the digests of the schedules the previous builder - a scan over
    all helpers per task, everything redone per simulated pool size - produced for C2 / C3 / C5 on 256 / 64 / 32 work-groups and
    for 40 random structures on 8 / 16 / 64 / 256: the queues, the update orders and the modelled times must be those."""
    import json
    import os
    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chol_schedule_digests.json")))
    for name in ("C2", "C3", "C5"):
        nb, nodes, pairs = _load_structure(name)
        for cus in (256, 64, 32):
            assert _digest(api.debug_chol_schedule(nb, nodes, pairs, cus=cus)) == golden[f"{name}/{cus}"], (name, cus)
    for seed in range(40):
        rng = np.random.default_rng(seed)
        nodes, nb = _random_tree(rng, depth=int(rng.integers(1, 6)))
        pairs = _random_pairs(rng, nodes, nb)
        for cus in (8, 16, 64, 256):
            assert _digest(api.debug_chol_schedule(nb, nodes, pairs, cus=cus)) == golden[f"r{seed}/{cus}"], (seed, cus)
