#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched waypoint environment on N MI355X (one process per GPU).

A "step" is ONE amenv_step() launch over this GPU's whole env batch (4096 envs/GPU by default):
mixer -> RK4 -> reward -> state machine -> auto-reset -> observation, actions already resident in
HBM (a pre-generated ring of action batches; no RNG, no policy inside the timed region).
Envs shard embarrassingly over GPUs (global env id = rank*N + i keys the reset RNG): there is NO
collective on the step path; torch.distributed is used only for the barrier and the max-over-ranks
of the timing.  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu 4096] [--vehicle hexa|quad]
                  [--mode graph|eager] [--no-cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
GRAPH_CHUNK = 64       # control steps captured per hipGraph (= action ring length)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16384)
    ap.add_argument("--warmup", type=int, default=1024)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--vehicle", default="hexa_arm", choices=["quad", "hexa", "hexa_arm"],
                    help="hexa_arm = BASELINE configs[2], the configuration the metric is quoted on; hexa = configs[1]; quad = the reference vehicle")
    ap.add_argument("--mode", default="graph", choices=["graph", "eager"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--block-size", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_actions(torch, n, ring, device, seed, act_dim=4):
    """Hover-centred synthetic actions (distribution B): thrust ~ N(1, 0.1), moments ~ N(0, 0.1), joint
    position commands (arm) ~ N(0, 0.3), clipped to the action box.  Long episodes with a steady trickle of crashes / resets."""
    g = torch.Generator(device=device).manual_seed(seed)
    a = torch.randn(ring, n, act_dim, device=device, generator=g) * 0.1
    a[..., 0] += 1.0
    a[..., 4:] *= 3.0
    lo = torch.tensor([0.0] + [-1.0] * (act_dim - 1), device=device)
    hi = torch.tensor([2.0] + [1.0] * (act_dim - 1), device=device)
    return torch.max(torch.min(a, hi), lo).contiguous()


def cpu_baseline(args, amd, n):
    """The CPU oracle (fp64 RK4 restatement of the reference step, oracle/amenv_oracle.c) timed on this
    box's host cores on a bounded sample of the same workload.  A reported baseline, not the target."""
    import ctypes as C

    import numpy as np

    from oracle import oracle as O

    cfg = O.reference_quad_config(num_envs=n, seed=0)
    pc = amd._lib.default_config(args.vehicle, n)
    C.memmove(C.byref(cfg.vehicle), C.byref(pc.vehicle), C.sizeof(O.Vehicle))  # same vehicle parameters
    threads = O.max_threads()
    orc = O.OracleEnv(cfg)
    orc.reset()
    rng = np.random.RandomState(0)
    T, ad = 64, orc.act_dim
    a = (rng.randn(T, n, ad) * 0.1).astype(np.float32)
    a[..., 0] += 1.0
    a[..., 4:] *= 3.0
    a = np.clip(a, [0] + [-1] * (ad - 1), [2] + [1] * (ad - 1)).astype(np.float32)
    orc.rollout(a[:8], nthreads=threads)  # warm
    # the box may expose more logical cores than this job's CPU share: time a short probe per thread count, keep the fastest
    best = (0.0, threads)
    for th in sorted({threads, 16, 32, 64, len(os.sched_getaffinity(0))}):
        if 1 <= th <= threads:
            orc.rollout(a[:4], nthreads=th)   # the first region after a change of thread count pays for the new team
            tp = time.perf_counter(); orc.rollout(a[:32], nthreads=th); rate = 32 * n / (time.perf_counter() - tp)
            best = max(best, (rate, th))
    threads = best[1]
    orc.rollout(a[:4], nthreads=threads)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < args.cpu_seconds:
        orc.rollout(a, nthreads=threads)
        steps += T * n
    dt = time.perf_counter() - t0
    # single-thread rate on a smaller sample, for the record
    t1 = time.perf_counter(); orc.rollout(a[:16], nthreads=1); st = 16 * n / (time.perf_counter() - t1)
    return {"value": steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{steps} env-steps ({n} envs, {steps // n} steps) of the same workload in {dt:.1f}s; "
                      f"fp64 RK4 C restatement (oracle/amenv_oracle.c), OpenMP over envs; single-thread {st:.3g} env-steps/s; "
                      f"host has {os.cpu_count()} logical cores, thread count chosen by a probe over 16/32/64/all"}


def main():
    args = parse()
    import torch

    import rl_aerial_manipulator_amd as amd

    n = args.envs_per_gpu
    shard = amd.sharding.shard_from_env(n)
    world, rank = shard.world, shard.rank
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = amd.sharding.init_process_group("nccl", device)  # RCCL; None for a single process

    env = amd.GpuWaypointEnv(n, device=local, vehicle=args.vehicle, seed=0, dtype=args.dtype, env_id_offset=shard.env_id_offset,
                             block_size=args.block_size)
    env.reset()
    ring = make_actions(torch, n, GRAPH_CHUNK, device, seed=1234 + rank, act_dim=env.act_dim)
    K, W = args.steps, args.warmup

    def run_eager(k):
        for t in range(k):
            env.step(ring[t % GRAPH_CHUNK])

    graph = None
    if args.mode == "graph":
        run_eager(GRAPH_CHUNK)  # first launches outside capture (module load)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run_eager(GRAPH_CHUNK)

    def run(k):
        if graph is None:
            run_eager(k)
        else:
            full, rem = divmod(k, GRAPH_CHUNK)
            for _ in range(full):
                graph.replay()
            run_eager(rem)

    def barrier():
        amd.sharding.barrier(dist)

    run(W)
    torch.cuda.synchronize(); barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(K)
    ev1.record()
    torch.cuda.synchronize(); barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    wall = amd.sharding.max_over_ranks(dist, wall, device)
    stats = env.stats()

    # per-launch kernel duration: HIP events stamped by the kernel's own dispatch (hipExtLaunchKernelGGL
    # start/stop events on the launch stream), same workload continuing; mean over `pairs` launches
    pairs = 512
    per = [env.step_timed(ring[i % GRAPH_CHUNK]) for i in range(pairs)]
    kern_ms = sum(per) / len(per) * 1e-3

    bytes_step = env.bytes_per_env_step
    total_envs = shard.global_envs
    value = amd.sharding.whole_job_rate(shard, K, wall)
    traffic = None  # HBM bytes per launch from the PMC passes (tools/pmc_traffic.py -> profiles/traffic.json), same workload
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            traffic = json.load(f).get(f"{args.vehicle}_{n}_{args.dtype}", {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    out = {
        "metric": "env-steps/sec (whole node), waypoint task at 4096 envs/GPU" if n == 4096 else f"env-steps/sec (whole node), waypoint task at {n} envs/GPU",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{n} envs/GPU, " + ("hexacopter + 3-link arm coupled dynamics (19 states, 7-D action)" if args.vehicle == "hexa_arm"
                                                     else f"{args.vehicle} 6-DOF rigid body") +
                               f" + {env.cfg.vehicle.n_rotors}-rotor mixer, RK4 dt=5ms, waypoint reward + reach/hold state machine + termination + "
                               f"auto-reset masks + {env.obs_dim}-D obs, one launch per control step",
                   "envs_per_gpu": n, "global_envs": total_envs, "vehicle": args.vehicle, "launch_mode": args.mode,
                   "graph_chunk": GRAPH_CHUNK if graph is not None else 0, "kernel": env.kernel_name,
                   "actions": "hover-centred N(1,0.1)/N(0,0.1) clipped, pre-generated ring in HBM", "parallelism": f"env-shard x{world}, no step-path collective"},
        # kernel duration = HIP events on the launch stream around the K timed launches (back-to-back graph replays: launch gaps
        # included, so an upper bound; rocprofv3 --kernel-trace --stats of the same command agrees to <1 %, profiles/)
        "roofline": {"bound": "hbm", "achieved": n * bytes_step / (dev_ms / K * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": n * bytes_step / (dev_ms / K * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                     "bytes_per_env_step": bytes_step, "kernel_us": dev_ms / K * 1e3,
                     "timing": f"HIP events on the launch stream over the {K} timed launches",
                     "kernel_us_isolated": kern_ms * 1e3, "kernel_us_isolated_min": min(per),
                     "isolated_timing": f"mean of {pairs} single launches, start/stop events stamped by the kernel dispatch (amenv_step_timed), each followed by a host sync",
                     "note": f"{n} envs = {(n + 63) // 64 * (2 if 'arm2w' in env.kernel_name else 1)} wavefronts on 256 CUs, {n * bytes_step / 1e6:.2f} MB algorithmic per launch"
                             + ("; latency-bound by construction: the launch is as long as one wave's instruction stream (one VALU instruction per 4 clocks for a lone wave; "
                                "SQ counters in profiles/: the critical wave issues VALU ~56 % of its lifetime, the rest is load / LDS / barrier latency), "
                                "so the HBM fraction is small by design -- DESIGN.md section 6" if n <= 65536 else "")},
        "device_ms_per_step": dev_ms / K,
        "episodes_finished_rank0": stats["episodes"],
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, amd, n)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
