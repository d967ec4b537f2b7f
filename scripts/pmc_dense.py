import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mavmap_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
B = rng.normal(size=(n, n)); A = B @ B.T + n * np.eye(n); b = rng.normal(size=n)
for _ in range(5):
    x = mavmap_amd.dense_spd_solve(A, b)
print(np.abs(x - np.linalg.solve(A, b)).max())
