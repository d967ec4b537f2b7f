"""Opportunistic cross-check against real Ceres (oracle/ceres_check). TEST INFRASTRUCTURE.

build() configures the harness with CMake; where Ceres is not installed (this image) it returns None and the callers
say so explicitly ("ceres": "unavailable") instead of implying a Ceres comparison."""
import os
import shutil
import struct
import subprocess
import tempfile

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SRC = os.path.join(_ROOT, "oracle", "ceres_check")
_BUILD = os.path.join(_ROOT, "oracle", "_build", "ceres_check")
_state = {}


def why_unavailable():
    """Why build() returned None: "no cmake", "Ceres not found by CMake" - or, on a box that HAS Ceres, the tail of the
    compiler output (the harness has never met a real Ceres installation: a box that has one must say why it failed
    instead of silently reporting "unavailable")."""
    build()
    return _state.get("why", "")


def build():
    """Path of the mavba_ceres_check binary, or None when Ceres (or cmake) is unavailable (see why_unavailable())."""
    if "exe" in _state:
        return _state["exe"]
    exe = None
    cmake = shutil.which("cmake")
    if not cmake:
        _state["why"] = "no cmake"
    else:
        try:
            os.makedirs(_BUILD, exist_ok=True)
            cfg = subprocess.run([cmake, "-S", _SRC, "-B", _BUILD], capture_output=True, text=True, timeout=300)
            if cfg.returncode != 0:
                _state["why"] = "cmake configure failed: " + (cfg.stderr or cfg.stdout)[-1500:]
            elif "harness not built" in cfg.stdout:
                _state["why"] = "Ceres not found by CMake (find_package(Ceres QUIET))"
            else:
                mk = subprocess.run([cmake, "--build", _BUILD, "-j", "8"], capture_output=True, text=True, timeout=1200)
                cand = os.path.join(_BUILD, "mavba_ceres_check")
                if mk.returncode == 0 and os.path.exists(cand):
                    exe = cand
                else:
                    _state["why"] = "Ceres IS installed but oracle/ceres_check did not compile:\n" + (mk.stderr or mk.stdout)[-3000:]
        except (OSError, subprocess.SubprocessError) as e:
            exe = None
            _state["why"] = repr(e)
    _state["exe"] = exe
    return exe


def solve(problem, options, threads=0):
    """Run the reference's Ceres path on `problem` (a BAProblem; not modified). Returns a dict or None (unavailable)."""
    exe = build()
    if exe is None:
        return None
    with tempfile.TemporaryDirectory() as d:
        pin, pout = os.path.join(d, "p.bin"), os.path.join(d, "r.bin")
        problem.save(pin, options)
        run = subprocess.run([exe, pin, pout, str(int(threads))], capture_output=True, text=True, timeout=3600)
        if run.returncode != 0:
            raise RuntimeError("mavba_ceres_check failed: " + run.stderr[-2000:])
        raw = open(pout, "rb").read()
    assert raw[:7] == b"MAVBAR1"
    initial_cost, final_cost = struct.unpack_from("<dd", raw, 8)
    succ, unsucc, term, nres = struct.unpack_from("<iiii", raw, 24)
    secs, = struct.unpack_from("<d", raw, 40)
    off = 48
    ni, nc, npt = problem.num_images, problem.num_cameras, problem.num_points

    def take(n, shape):
        nonlocal off
        a = np.frombuffer(raw, "<f8", n, off).reshape(shape).copy()
        off += 8 * n
        return a
    return dict(initial_cost=initial_cost, final_cost=final_cost, num_successful_steps=succ, num_unsuccessful_steps=unsucc,
                termination_type=term, num_residuals=nres, solve_seconds=secs, poses=take(ni * 6, (ni, 6)),
                intrinsics=take(nc * 9, (nc, 9)), points=take(npt * 3, (npt, 3)), point_errors=take(npt, (npt,)),
                log=run.stdout)
