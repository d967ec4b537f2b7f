"""Times the front end of a linear solve (k_schur_rows) for library variants, each in a fresh interpreter (debug harness).

  python scripts/_dbg/time_front.py [config] lib1.so lib2.so ...     (no library: the built one)"""
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(root, "mavmap_amd", "lib", "libmavba.so")
args = sys.argv[1:]
cfg = args.pop(0) if args and args[0] in ("C2", "C3", "C5") else "C3"
child = r'''
import sys; sys.path.insert(0, %r)
import mavmap_amd
from mavmap_amd import synth
p = synth.make_config(%r)
with mavmap_amd.Session(p, dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10)) as s:
    ms = [s.time_front(1e4, 30) for _ in range(3)]
print(" ".join("%%.4f" %% m for m in ms))
''' % (root, cfg)
def run():
    out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True)
    return out.stdout.strip() + (" ERR " + out.stderr.strip()[-300:] if out.returncode else "")
if not args:
    print(cfg, "built library:", run())
else:
    keep = lib + ".keep"
    shutil.copy(lib, keep)
    try:
        for v in args:
            shutil.copy(v, lib)
            print(cfg, os.path.basename(v), "|", run(), flush=True)
    finally:
        shutil.move(keep, lib)
