"""Build libeigensolver_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m eigensolver_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box.
-ffp-contract=off: fused multiply-adds appear only where the source calls fma() explicitly, so that the
arithmetic is reproducible against the CPU oracle ("numerical contract").
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libeigensolver_amd.so")
# test-only variant compiled with -DES_IEEE_DIVISION: every reciprocal of the hot loops is the IEEE quotient the CPU port
# computes, so the interior march is bit-identical to oracle/c/shoot_port.c (tests/test_shoot_gpu.py); never loaded by
# the product path
LIB_IEEE = os.path.join(LIBDIR, "libeigensolver_amd_ieee.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    out = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    out.append(os.path.join(os.path.dirname(HERE), "include", "eigensolver_amd.h"))
    return out


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(s) > t for s in _deps())


def _compile_and_link(lib, extra_flags, tag, verbose):
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + tag + ".o")
        cmd = [hipcc] + FLAGS + extra_flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return lib


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    # ES_BUILD_ALL_SHAPES=1: the measuring build of tools/probe/time_grid_shapes.py -- every (points per lane, waves per
    # SIMD) launch shape of the grid kernel that ES_GRID_SHAPE can name, not only the ones the tables select
    extra = ["-DES_ALL_GRID_SHAPES"] if os.environ.get("ES_BUILD_ALL_SHAPES") == "1" else []
    extra += os.environ.get("ES_BUILD_EXTRA_FLAGS", "").split()          # experiments (-D...)
    if force or needs_build(LIB):
        _compile_and_link(LIB, extra, "", verbose)
    if force or needs_build(LIB_IEEE):
        _compile_and_link(LIB_IEEE, ["-DES_IEEE_DIVISION"], ".ieee", verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
