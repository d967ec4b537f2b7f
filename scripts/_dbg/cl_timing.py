import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
from mavmap_amd import synth, api
p = synth.make_config("C3", 1.0)
with api.Session(p) as s:
    print(s.info())
    for _ in range(3): s.linear_step(1e4)
    out = (C.c_longlong * 8)()
    api.load().mavba_debug_cl(out)
    print("total %d zero %d scatter %d mfma %d emit %d nbatch %d fetch-issue %d (cycles)" % tuple(out[:7]))
