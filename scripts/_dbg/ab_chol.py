"""A/B timing of the factorisation between two builds of the library in ONE process-per-variant run (debug harness).

  python scripts/_dbg/ab_chol.py mavmap_amd/lib/libmavba_A.so mavmap_amd/lib/libmavba_B.so [rounds]

Each variant is copied over libmavba.so and timed in a fresh interpreter (C3 and C2, event timers), alternating."""
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(root, "mavmap_amd", "lib", "libmavba.so")
child = r'''
import sys; sys.path.insert(0, %r)
import mavmap_amd
from mavmap_amd import synth
for cfg in ("C3", "C2"):
    p = synth.make_config(cfg)
    with mavmap_amd.Session(p, dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10, profile_kernels=1)) as s:
        s.iterate(6)
        s0 = s.kernel_stats()
        left = 60
        while left > 0:
            done, term = s.iterate(left); left -= done
            if term != 0 and left > 0: s.reset()
            if done == 0: break
        s1 = s.kernel_stats()
    out = []
    for k in ("chol_factor", "schur_fused", "chol_backsolve"):
        n = s1[k]["launches"] - s0[k]["launches"]; t = s1[k]["total_ms"] - s0[k]["total_ms"]
        out.append("%%s %%.4f" %% (k, t / max(n, 1)))
    print(cfg, " ".join(out))
''' % root
variants = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
keep = lib + ".keep"
shutil.copy(lib, keep)
try:
    for r in range(rounds):
        for v in variants:
            shutil.copy(v, lib)
            out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True)
            print(os.path.basename(v), "|", " | ".join(out.stdout.strip().splitlines()), out.stderr.strip()[-200:] if out.returncode else "")
finally:
    shutil.move(keep, lib)
