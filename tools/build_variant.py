#!/usr/bin/env python3
"""Build a diagnostic variant of libamenv.so with the library's own flags plus extra ones.
  python tools/build_variant.py stamps -DAMENV_STAMPS        -> tools/micro/libamenv_stamps.so  (use with AMENV_LIB=...)
  python tools/build_variant.py slp --drop=-fno-slp-vectorize -> the library with packed fp32 (SLP vectorisation) allowed"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rl_aerial_manipulator_amd.build as B

name, extra = sys.argv[1], sys.argv[2:]
drop = {"-Wall"} | {a[len("--drop="):] for a in extra if a.startswith("--drop=")}   # --drop=<flag>: remove one of the library's flags
extra = [a for a in extra if not a.startswith("--drop=")]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", f"libamenv_{name}.so")
cmd = [B.hipcc()] + [f for f in B.FLAGS if f not in drop] + extra + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", out]
subprocess.check_call(cmd, cwd=B.CSRC)
print(out)
