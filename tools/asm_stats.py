#!/usr/bin/env python3
"""Instruction mix of the gfx950 kernels: hipcc -S the library and count per kernel.
Usage: python tools/asm_stats.py [filter-substring]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rl-aerial-manipulator_amd", "csrc")
out = "/tmp/amenv.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-gpu-rdc", "-S",
                       "--cuda-device-only", "-o", out, "amenv_capi.hip"], cwd=CSRC)
filt = sys.argv[1] if len(sys.argv) > 1 else "step_kernelIf"
lines = open(out).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if l.startswith("_ZN9amenv_dev") and ":" in l]
for (i, name), nxt in zip(starts, starts[1:] + [(len(lines), "")]):
    if filt not in name:
        continue
    body = lines[i:nxt[0]]
    end = next((k for k, l in enumerate(body) if ".end_amdhsa_kernel" in l), len(body))
    ins = [l.strip().split()[0] for l in body[:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    grp = lambda p: sum(n for k, n in c.items() if k.startswith(p))
    meta = "\n".join(body[:end])
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta)
    sg = re.search(r"\.amdhsa_next_free_sgpr (\d+)", meta)
    print(f"{name[:72]}\n  total {len(ins)}  valu {grp('v_')}  (pk {grp('v_pk')}, trans {sum(c[k] for k in c if re.match(r'v_(rcp|rsq|sqrt|exp|log|sin|cos)', k))})"
          f"  salu {grp('s_')} (s_load {grp('s_load')}, waitcnt {c['s_waitcnt']})  gload {grp('global_load')}  gstore {grp('global_store')}"
          f"  ds {grp('ds_')}  scratch {grp('scratch_')}  vgpr {vg.group(1) if vg else '?'} sgpr {sg.group(1) if sg else '?'}")
