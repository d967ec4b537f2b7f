import ctypes as C, numpy as np, os, sys
rng = np.random.default_rng(0); B = rng.normal(size=(64, 64)); A = np.ascontiguousarray(B @ B.T + 64 * np.eye(64))
for v in ("full", "nox", "norsq", "nochk"):
    L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"libtile_{v}.so"))
    ts = np.zeros(64, np.int64)
    L.dbg_run(A.ctypes.data_as(C.c_void_p), ts.ctypes.data_as(C.c_void_p))
    print(v, "diag16 cycles:", [int(ts[3+4*cb]-ts[2+4*cb]) for cb in range(4)], "total", int(ts[22]-ts[0]))
