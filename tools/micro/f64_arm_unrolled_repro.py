#!/usr/bin/env python3
"""Reproducer / record for the fp64 arm kernel that "computed garbage" in round 1.

The fp64 arm step kernel with its four RK4 stages INLINED (four copies of the multibody RHS) needs more than the 512 registers a
wavefront can have: hipcc (ROCm 7.2, gfx950) spills 110 VGPRs to scratch (332 B per lane) next to 324 spilled SGPRs.  The product build
wraps ONE copy of the RHS in a 4-iteration loop instead (256 VGPRs + 202 AGPRs, no scratch; csrc/amenv_arm.hpp dynamics_arm).

  build container :  python tools/micro/f64_arm_unrolled_repro.py --build     -> tools/micro/libamenv_f64unrolled.so (-DAMENV_F64_ARM_UNROLLED)
  GPU box         :  python tools/micro/f64_arm_unrolled_repro.py             -> one teacher-forced step of both builds against the fp64 oracle

Prints the max error of every state row for the product library and for the unrolled (spilling) one.  (Test infrastructure: uses the
oracle as the checker.)"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ALT = os.path.join(ROOT, "tools", "micro", "libamenv_f64unrolled.so")
sys.path.insert(0, ROOT)


def build():
    csrc = os.path.join(ROOT, "rl-aerial-manipulator_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-ffp-contract=off", "-fno-slp-vectorize",
           "-DAMENV_F64_ARM_UNROLLED", "-Rpass-analysis=kernel-resource-usage", os.path.join(csrc, "amenv_capi.hip"), "-o", ALT]
    r = subprocess.run(cmd, cwd=csrc, stderr=subprocess.PIPE, text=True)
    want = None
    for ln in r.stderr.splitlines():   # resource usage of the fp64 arm step kernel only
        if "Function Name" in ln:
            want = "step_kernelIdLi6ELi1ELi0ELi3" in ln
        elif want and any(k in ln for k in ("VGPRs:", "AGPRs:", "ScratchSize", "Spill")):
            print(ln.split("remark:")[1].strip())
    sys.exit(r.returncode)


def child():
    import numpy as np
    import torch

    import rl_aerial_manipulator_amd as amd
    from oracle import oracle as O
    from tests.test_arm_cpu import arm_cfg
    n = 256
    env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=9, dtype="f64", auto_reset=False)
    cfg = arm_cfg(); cfg.num_envs = n; cfg.seed = 9; cfg.flags = 0
    orc = O.OracleEnv(cfg)
    env.reset()
    rng = np.random.RandomState(3)
    f, i = env.get_state()
    f = f.cpu().numpy(); i = i.cpu().numpy()
    q = rng.normal(size=(4, n)) * 0.2; q[0] += 1; q /= np.linalg.norm(q, axis=0)
    f[6:10] = q; f[10:13] = rng.normal(0, 1.0, (3, n)); f[3:6] = rng.normal(0, 0.5, (3, n))
    f[19:22] = rng.uniform(-1, 1, (3, n)); f[22:25] = rng.normal(0, 1.0, (3, n))
    env.set_state(f, i)
    orc.fstate[:] = f; orc.istate[:] = i
    a = rng.uniform([0.6, -1, -1, -1, -1, -1, -1], [1.4, 1, 1, 1, 1, 1, 1], (n, 7)).astype(np.float32)
    a[:, 1:4] *= 0.05
    env.step(torch.from_numpy(a).cuda())
    orc.step(a)
    f2, _ = env.get_state()
    err = np.abs(f2.cpu().numpy() - orc.fstate).max(1)
    print(json.dumps({"lib": os.environ.get("AMENV_LIB", "product"), "kernel": env.kernel_name, "max_abs_err_rows_0_12": float(err[:13].max()),
                      "max_abs_err_joint_rows": float(err[19:25].max())}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true"); ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.build:
        build()
    elif a.child:
        child()
    else:
        for lib in (None, ALT):
            env = dict(os.environ)
            if lib:
                env["AMENV_LIB"] = lib
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)
