import ctypes as C, numpy as np, os
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtile_dbg.so"))
rng = np.random.default_rng(0); B = rng.normal(size=(64, 64)); A = np.ascontiguousarray(B @ B.T + 64 * np.eye(64))
ts = np.zeros(64, np.int64)
L.dbg_run(A.ctypes.data_as(C.c_void_p), ts.ctypes.data_as(C.c_void_p))
t0 = ts[0]
names = {0: "start", 1: "enter potrf_inv", 20: "begin inverse assembly", 21: "end potrf_inv", 22: "end kernel"}
for cb in range(4):
    names[2 + 4 * cb] = f"cb{cb} begin"; names[3 + 4 * cb] = f"cb{cb} after diag16"; names[4 + 4 * cb] = f"cb{cb} after panel"
order = sorted(names)
prev = t0
for i in order:
    print(f"{names[i]:28s} t={ts[i]-t0:8d} cycles  (+{ts[i]-prev})")
    prev = ts[i]
