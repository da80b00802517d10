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
    step = [v for k, v in acc.items() if "step_kernel" in k]
    mean = lambda xs: sum(xs) / len(xs)
    # the calibration dispatch is selected BY NAME: torch's elementwise kernel of `torch.mul(src, 1.0, out=dst)` over 256 MiB (pmc_step.py);
    # among the elementwise dispatches of the run it is the only one of that size
    cal = max((max(v) for k, v in acc.items() if "elementwise_kernel" in k), default=None)
    return mean(step[0][20:]), cal


CAL_KIB = 256 * 1024   # the calibration copy reads and writes 256 MiB


def calibration_ok(f_cal, w_cal):
    """FETCH_SIZE must read half the copy (the gfx950 16-B/lane correction), WRITE_SIZE all of it, both within 5 %."""
    return f_cal is not None and w_cal is not None and abs(2.0 * f_cal / CAL_KIB - 1.0) < 0.05 and abs(w_cal / CAL_KIB - 1.0) < 0.05



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
        f_kib, f_cal = collect("FETCH_SIZE", n, a.vehicle, a.out)
        w_kib, w_cal = collect("WRITE_SIZE", n, a.vehicle, a.out)
        fetch = 2.0 * f_kib * 1024.0   # gfx950 correction for 16-B/lane coalesced reads
        write = w_kib * 1024.0
        if not calibration_ok(f_cal, w_cal):   # never record a traffic figure whose unit / correction check failed
            print(f"N={n}: calibration copy reads fetch {f_cal} KiB (want ~{CAL_KIB // 2}) write {w_cal} KiB (want ~{CAL_KIB}): entry NOT written", flush=True)
            res.pop(f"{a.vehicle}_{n}_f32", None)
            continue
        res[f"{a.vehicle}_{n}_f32"] = dict(fetch_size_kib_raw=f_kib, write_size_kib_raw=w_kib, fetch_bytes_per_launch=fetch,
                                          write_bytes_per_launch=write, hbm_bytes_per_launch=fetch + write,
                                          calibration_copy_256MiB=dict(fetch_kib_raw=f_cal, write_kib_raw=w_cal))
        print(f"N={n:8d}: FETCH_SIZE {f_kib:10.1f} KiB (x2 -> {fetch/1e6:8.3f} MB)  WRITE_SIZE {w_kib:10.1f} KiB ({write/1e6:8.3f} MB)  "
              f"per env-step {(fetch + write) / n:7.1f} B   calibration copy(256 MiB): fetch {f_cal} KiB write {w_cal} KiB", flush=True)
    with open(path, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
