#!/usr/bin/env python3
"""HBM traffic of the step kernel from rocprofv3 PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass); on gfx950 FETCH_SIZE reads exactly
half of a wide (16 B/lane) coalesced read stream -> doubled; WRITE_SIZE is exact for 16-B/lane stores.  Both counters
are in KiB.  A calibration copy kernel of known size runs in the same passes to confirm unit and correction.

Run on the GPU box from the repo root:   python tools/pmc_traffic.py --envs 4096 32768 1048576
Writes gpurun_out/traffic.json (copy to profiles/traffic.json) and prints a table."""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "auto"


def collect(counter, envs, vehicle, outdir):
    d = os.path.join(outdir, f"pmc_{counter}_{vehicle}_{envs}")
    import shutil
    shutil.rmtree(d, ignore_errors=True)   # a directory left by another run would mix its dispatches into this one's averages
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "pmc_step.py"),
           "--envs", str(envs), "--steps", "120", "--vehicle", vehicle, "--calibrate", "--kernel", KERNEL]
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=300)
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    step = [(k, v) for k, v in acc.items() if "step_kernel" in k]
    mean = lambda xs: sum(xs) / len(xs)
    # calibration dispatches are selected BY NAME: the library's calibration_copy_kernel<float4> (16 B per lane) / <float> (4 B per lane)
    cal = {w: next((mean(v) for k, v in acc.items() if "calibration_copy_kernel" in k and (("float4" in k or "HIP_vector_type" in k) == (w == 16))), None) for w in (16, 4)}
    name, vals = step[0]
    return mean(vals[20:]), cal, ("team" in name)


CAL_KIB = 256 * 1024   # the calibration copy reads and writes 256 MiB


def factor(kib, what):
    """counter units -> bytes for this access width, from the calibration copy (must land within 5 % of a simple ratio: 1 or 2)"""
    if kib is None or kib <= 0:
        return None
    f = CAL_KIB / kib
    for want in (1.0, 2.0):
        if abs(f / want - 1.0) < 0.05:
            return f
    print(f"calibration {what}: the 256 MiB copy reads as {kib} KiB (factor {f:.3f}): not a clean 1x / 2x", flush=True)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", nargs="+", type=int, default=[4096])
    ap.add_argument("--vehicle", default="hexa")
    ap.add_argument("--kernel", default="auto")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    a = ap.parse_args()
    global KERNEL
    KERNEL = a.kernel
    a.out = os.path.abspath(a.out)   # rocprofv3 runs with cwd=/tmp
    os.makedirs(a.out, exist_ok=True)
    path = os.path.join(a.out, "traffic.json")
    res = json.load(open(path)) if os.path.exists(path) else {}    # merge: one invocation per vehicle
    for n in a.envs:
        f_kib, f_cal, team = collect("FETCH_SIZE", n, a.vehicle, a.out)
        w_kib, w_cal, _ = collect("WRITE_SIZE", n, a.vehicle, a.out)
        width = 4 if team else 16     # access width of the kernel under test: lane-team kernels move dwords, the others 16 B per lane
        kf, kw = factor(f_cal[width], f"FETCH_SIZE @{width} B/lane"), factor(w_cal[width], f"WRITE_SIZE @{width} B/lane")
        key = f"{a.vehicle}_{n}_f32"
        if kf is None or kw is None:   # never record a traffic figure whose unit / correction check failed
            print(f"N={n}: calibration failed ({f_cal}, {w_cal}): entry NOT written", flush=True)
            res.pop(key, None)
            continue
        fetch, write = kf * f_kib * 1024.0, kw * w_kib * 1024.0
        res[key] = dict(fetch_size_kib_raw=f_kib, write_size_kib_raw=w_kib, access_bytes_per_lane=width, fetch_factor=kf, write_factor=kw,
                        fetch_bytes_per_launch=fetch, write_bytes_per_launch=write, hbm_bytes_per_launch=fetch + write,
                        calibration_copy_256MiB=dict(fetch_kib_raw=f_cal, write_kib_raw=w_cal))
        print(f"N={n:8d} ({width} B/lane): FETCH_SIZE {f_kib:10.1f} KiB x{kf:.2f} -> {fetch/1e6:8.3f} MB   WRITE_SIZE {w_kib:10.1f} KiB x{kw:.2f} -> {write/1e6:8.3f} MB   "
              f"per env-step {(fetch + write) / n:7.1f} B   calibration copies (256 MiB): fetch {f_cal} KiB, write {w_cal} KiB", flush=True)
    with open(path, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
