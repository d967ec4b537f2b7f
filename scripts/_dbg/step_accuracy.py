import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
import numpy as np
from tests.test_gpu_random import random_problem
from tests import oracle_lib as O
import mavmap_amd.api as M
for seed in map(int, sys.argv[1:]):
    p, opts, _ = random_problem(seed)
    for radius in (1e4, 9e4, 1e6):
        ref = O.linear_step(p, radius, O.options(**opts))
        with M.Session(p, opts) as s:
            S, v = s.reduced_system(radius)
            st = s.linear_step(radius)
        So, vo = ref["S"], ref["v"]
        act = np.abs(np.diag(So)) > 0
        # refined solution of the ORACLE's system in extended precision
        A = So.astype(np.longdouble); b = vo.astype(np.longdouble)
        y = np.linalg.solve(So, vo).astype(np.longdouble)
        for _ in range(5):
            r = b - A @ y
            y = y + np.linalg.solve(So, r.astype(np.float64)).astype(np.longdouble)
        y = y.astype(np.float64)
        def flat(d):  # reduced camera vector in uniform indexing: 6*NI poses then 9*NC intrinsics
            return np.concatenate([d["d_poses"].ravel(), d["d_intr"].ravel()])
        n = len(y)
        print("seed %d radius %.0e n=%d cond=%.2e |S_gpu-S_or|/|S|=%.1e  |v|: %.1e" % (seed, radius, n, np.linalg.cond(So),
              np.abs(S - So).max() / np.abs(So).max(), np.abs(v - vo).max() / np.abs(vo).max()))
        print("   keys", [k for k in ref.keys()], [k for k in st.keys()])
        if "y" in ref and "y" in st:
            print("   oracle y err %.2e   gpu y err %.2e" % (np.abs(ref["y"] - y).max() / np.abs(y).max(), np.abs(st["y"] - y).max() / np.abs(y).max()))
