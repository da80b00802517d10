#!/usr/bin/env python3
"""Build a diagnostic variant of libamenv.so with the library's own flags plus extra ones.
  python tools/build_variant.py stamps -DAMENV_STAMPS        -> tools/micro/libamenv_stamps.so  (use with AMENV_LIB=...)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rl_aerial_manipulator_amd.build as B

name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", f"libamenv_{name}.so")
cmd = [B.hipcc()] + [f for f in B.FLAGS if f != "-Wall"] + extra + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", out]
subprocess.check_call(cmd, cwd=B.CSRC)
print(out)
