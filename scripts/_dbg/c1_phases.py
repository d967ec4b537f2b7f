import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
from mavmap_amd import synth, api, _abi as A
full = synth.make_config("C1", 1.0)
w = synth.local_ba_window(full, 0, 8)
opts = dict(max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10)
api.bundle_adjustment(w.copy(), opts)
lib = api.load()
for rep in range(3):
    q = w.copy()
    t0 = time.time(); cp = q.c_struct(); o = api.make_options(opts); t1 = time.time()
    h = C.c_void_p()
    rc = lib.mavba_session_create(C.byref(cp), C.byref(o), C.byref(h)); t2 = time.time()
    done = C.c_int(); term = C.c_int()
    lib.mavba_session_iterate(h, 51, C.byref(done), C.byref(term)); t3 = time.time()
    res = A.CResult(); lib.mavba_session_result(h, C.byref(res)); t4 = time.time()
    lib.mavba_session_get_params(h, cp.poses, cp.intrinsics, cp.points); t5 = time.time()
    lib.mavba_session_destroy(h); t6 = time.time()
    print("py %.2f | create %.2f | iterate %.2f | result %.2f | get_params %.2f | destroy %.2f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)))
