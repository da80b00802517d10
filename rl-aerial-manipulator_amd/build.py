"""Build libamenv.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box
with the repo snapshot.  Usage: python -m build  (from the package dir) or build.build().
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libamenv.so")
SOURCES = ["amenv_capi.hip"]
DEPS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))) + [os.path.join("..", "..", "include", "amenv.h")]   # every source of the one translation unit
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function",
         # episode-end code adds its own lane's numbers with plain no-return atomics (usually ONE lane is active): the wave-reduction
         # the compiler would wrap around same-address atomics is a hundred instructions on a path whose cost is its instruction count
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm to build libamenv.so)")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
