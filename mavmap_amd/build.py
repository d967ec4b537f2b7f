"""Builds libmavba.so (HIP, gfx950 only) in-tree: mavmap_amd/lib/libmavba.so.

hipcc cross-compiles without a GPU, so this runs on the CPU-only build container as well as on
the MI355X box. The built library is git-ignored but travels with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
LIBDIR = os.path.join(ROOT, "lib")
LIB = os.path.join(LIBDIR, "libmavba.so")
OBJDIR = os.path.join(LIBDIR, "obj")
SOURCES = ["kernels.hip", "schur_rows.hip", "dense_chol.hip", "pose_refine.hip", "host_util.hip", "session_build.hip", "session_lm.hip", "scene.hip", "multi_gpu.hip", "device_setup.hip", "api.hip"]
HEADERS = ["ba_math.h", "dev_reduce.h", "internal.h", "session.h", "lm_decide.h", "lm_bodies.h", "sweep_body.h", os.path.join("..", "..", "include", "mavba.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wall",
         "-Wno-unused-result"]
# Per-file additions. dense_chol.hip: matrix instructions with their accumulators in ordinary vector registers - the
# factorisation's pivot chain reads every result back at once (v_readlane, selects), and with the accumulators in the AGPR half
# of the file the compiler wrapped each matrix instruction in 8-16 v_accvgpr moves (3 076 of them in this file, 30 with the flag).
FILE_FLAGS = {"dense_chol.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmavba.so)")


STAMP = os.path.join(LIBDIR, "build_stamp.json")


def source_digest():
    """sha256 over the sources and headers the library is built from (content, not mtimes: the snapshot that
    travels to the GPU box need not preserve them)."""
    import hashlib
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.normpath(os.path.join(CSRC, f)), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def is_current():
    """True when libmavba.so exists and was built from exactly the sources present now."""
    import json
    try:
        return os.path.exists(LIB) and json.load(open(STAMP))["digest"] == source_digest()
    except (OSError, ValueError, KeyError):
        return False


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    hipcc = _hipcc()
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _stale(obj, [sp] + hdrs):
            cmd = [hipcc] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    if not is_current():
        import json
        with open(STAMP, "w") as fh:
            json.dump(dict(digest=source_digest()), fh)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
