#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched waypoint environment on N MI355X (one process per GPU).

A "step" is ONE amenv_step() launch over this GPU's whole env batch (4096 envs/GPU by default): mixer -> RK4 -> forward
kinematics -> reward -> state machine -> auto-reset -> observation, actions already resident in HBM (a pre-generated ring of
action batches; no RNG, no policy inside the timed region).  Envs shard embarrassingly over GPUs (global env id = rank*N + i
keys the reset RNG): there is NO collective on the step path; torch.distributed (RCCL) is used only for the barrier and the
max-over-ranks of the timing.  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu 4096] [--vehicle hexa_arm|hexa|quad]
                  [--mode graph|eager] [--repeats 5] [--preroll 4096] [--no-cpu-baseline] [--no-extras]

Protocol:
  * pre-roll (outside every timed region, independent of --warmup): `--preroll` steps so that episodes are staggered and the
    auto-reset trickle is in steady state (time-limit truncations included: 2 x the 2000-step episode limit by default);
  * W untimed warm-up steps, then `--repeats` windows of EXACTLY K steps, each bracketed by barrier + synchronize on both sides;
    `value` = whole-job env-steps / the MEDIAN window's wall time (max over ranks); every window is listed under "windows";
  * graph mode replays ONE hipGraph of min(K, 64) steps (plus one graph for the remainder), so every timed step is a graph
    launch and `launch_mode` says what ran; eager mode issues K plain launches;
  * `--gpus N` with no torchrun environment: this process starts the N ranks itself as CHILD processes (torch.distributed.run),
    before it has touched torch or the GPU, and relays rank 0's line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
GRAPH_CHUNK = 64       # most control steps captured per hipGraph (= action ring length)
BASELINE_METRIC = "env-steps/sec (whole node), hexacopter+arm waypoint task at 4096 envs/GPU"   # BASELINE.json
REFERENCE_PYTHON_STEPS_PER_S = [424, 642]   # the reference's own step(), 1 core, LSODA, measured in the build container


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--vehicle", default="hexa_arm", choices=["quad", "hexa", "hexa_arm"],
                    help="hexa_arm = BASELINE configs[2], the configuration the metric is quoted on; hexa = configs[1]; quad = the reference vehicle")
    ap.add_argument("--mode", default="graph", choices=["graph", "eager"])
    ap.add_argument("--kernel", default="auto", choices=["auto", "lane", "helper", "team", "staged"])
    ap.add_argument("--actions", default="hover", choices=["hover", "uniform"], help="SURVEY 8d action sets B (hover-centred) / A (uniform stress)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--block-size", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--preroll", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the rollout-kernel and SURVEY 8d extra measurements")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry-run", action="store_true", help="launcher / process-group / timing plumbing only, with a no-op step (CPU test of --gpus N)")
    return ap.parse_args(argv)


# ---- --gpus N without a torchrun environment: start the ranks as child processes ------------------------------------------------
def launch_children(args):
    """Parent of an N-rank job.  Runs before torch is imported (nothing here has initialised the GPU); never exec()s."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in p.stdout.splitlines():
        try:
            if "metric" in json.loads(ln):
                line = ln
        except ValueError:
            sys.stderr.write(ln + "\n")
    if p.returncode != 0 or line is None:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank job failed (exit {p.returncode})\n")
        return p.returncode or 1
    print(line, flush=True)
    return 0


def baseline_config(vehicle, n, world):
    """Which entry of BASELINE.json's `configs` a run is."""
    if vehicle == "hexa" and n == 4096 and world == 1:
        return "configs[1]: 4096 parallel envs, hexacopter-only 6-DOF rigid body + 6-rotor mixer, RK4, fp32, 1xMI355X"
    if vehicle == "hexa_arm" and n == 4096:
        return "configs[2]: 4096 envs, hexacopter + 3-link arm coupled dynamics + waypoint reward + reset masks" + (", 1xMI355X" if world == 1 else f", x{world} MI355X (the metric's 1, 2, 4, 8 sweep)")
    if vehicle == "hexa_arm" and n == 32768:
        return f"configs[3]: {n * world} envs sharded {world}xMI355X (32768/GPU, embarrassingly parallel, no collectives on step path)"
    return None


def make_actions(torch, kind, n, ring, device, seed, act_dim=4):
    """SURVEY 8d action sets.  hover (B): thrust ~ N(1, 0.1), moments ~ N(0, 0.1), joint commands ~ N(0, 0.3), clipped to the action
    box -- long episodes with a steady trickle of crashes / resets.  uniform (A): i.i.d. U(low, high) -- tumbling, frequent resets."""
    g = torch.Generator(device=device).manual_seed(seed)
    lo = torch.tensor([0.0] + [-1.0] * (act_dim - 1), device=device)
    hi = torch.tensor([2.0] + [1.0] * (act_dim - 1), device=device)
    if kind == "uniform":
        return (lo + (hi - lo) * torch.rand(ring, n, act_dim, device=device, generator=g)).contiguous()
    a = torch.randn(ring, n, act_dim, device=device, generator=g) * 0.1
    a[..., 0] += 1.0
    a[..., 4:] *= 3.0
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
    cfg.task.ee_task = pc.task.ee_task
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
    # single-thread rates on smaller samples, for the record: the batch, and ONE env for 1000 steps (BASELINE configs[0]'s shape)
    t1 = time.perf_counter(); orc.rollout(a[:16], nthreads=1); st = 16 * n / (time.perf_counter() - t1)
    cfg1 = O.reference_quad_config(num_envs=1, seed=0)
    C.memmove(C.byref(cfg1.vehicle), C.byref(pc.vehicle), C.sizeof(O.Vehicle)); cfg1.task.ee_task = pc.task.ee_task
    o1 = O.OracleEnv(cfg1); o1.reset()
    a1 = np.ascontiguousarray(np.tile(a[:, :1], (16, 1, 1))[:1000])
    t2 = time.perf_counter(); o1.rollout(a1, nthreads=1); one = 1000 / (time.perf_counter() - t2)
    return {"value": steps / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{steps} env-steps ({n} envs, {steps // n} steps) of the same workload in {dt:.1f}s; "
                      f"fp64 RK4 C restatement (oracle/amenv_oracle.c), OpenMP over envs; single-thread {st:.3g} env-steps/s; "
                      f"host has {os.cpu_count()} logical cores, thread count chosen by a probe over 16/32/64/all",
            "single_thread_env_steps_per_s": st, "single_env_1000_steps_per_s": one,
            "reference_python_steps_per_s": REFERENCE_PYTHON_STEPS_PER_S,
            "reference_python_note": "the reference's own WaypointQuadEnv.step() (quadrotor, scipy LSODA), 1 core of the BUILD container; "
                                     "Python source cannot travel to the GPU box (BASELINE.md section 3)"}


class _DryEnv:
    """No-op stand-in used by --dry-run: exercises the launcher, the process group and the timing protocol without a GPU."""
    obs_dim, act_dim, bytes_per_env_step, kernel_name = 20, 4, 0, "dry-run (no kernel)"

    def step(self, a):
        return None

    def stats(self, reset=False):
        return {"episodes": 0, "steps": 0}

    def close(self):
        pass


def main_worker(args):
    import statistics

    import torch

    import rl_aerial_manipulator_amd as amd

    n = args.envs_per_gpu
    shard = amd.sharding.shard_from_env(n)
    world, rank = shard.world, shard.rank
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run
    if dry:
        device = torch.device("cpu")
    else:
        if local >= torch.cuda.device_count():   # counting devices does not initialise the GPU
            raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but only {torch.cuda.device_count()} are visible (--gpus too large)")
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    dist = amd.sharding.init_process_group(args.backend, None if dry else device)  # RCCL; None for a single process

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def barrier():
        amd.sharding.barrier(dist)

    if dry:
        env, ring = _DryEnv(), [None] * GRAPH_CHUNK
    else:
        env = amd.GpuWaypointEnv(n, device=local, vehicle=args.vehicle, seed=0, dtype=args.dtype, env_id_offset=shard.env_id_offset,
                                 block_size=args.block_size, kernel=args.kernel)
        env.reset()
        ring = make_actions(torch, args.actions, n, GRAPH_CHUNK, device, seed=1234 + rank, act_dim=env.act_dim)
    K, W = args.steps, args.warmup

    def run_eager(k, first=0):
        for t in range(k):
            env.step(ring[(first + t) % GRAPH_CHUNK])

    def capture(k):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run_eager(k)
        return g

    use_graph = args.mode == "graph" and not dry
    plan = {"full": 0, "rem": 0, "chunk": 0}
    g_full = g_rem = None
    if use_graph:
        run_eager(GRAPH_CHUNK)  # first launches outside capture (module load)
        sync()
        chunk = min(K, GRAPH_CHUNK)
        plan = {"chunk": chunk, "full": K // chunk, "rem": K % chunk}
        g_full = capture(chunk)
        g_rem = capture(plan["rem"]) if plan["rem"] else None

    def run_untimed(k):
        if use_graph:
            for _ in range(k // plan["chunk"]):
                g_full.replay()
            run_eager(k % plan["chunk"])
        else:
            run_eager(k)

    def run_timed():   # EXACTLY K steps
        if use_graph:
            for _ in range(plan["full"]):
                g_full.replay()
            if g_rem is not None:
                g_rem.replay()
        else:
            run_eager(K)

    # steady state: staggered episodes, auto-reset trickle running (never timed, independent of --warmup)
    run_untimed(args.preroll)
    sync()
    env.stats(reset=True)
    run_untimed(W)
    windows = []
    for _ in range(max(1, args.repeats)):
        sync(); barrier()
        if not dry:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()   # on torch's current stream = the stream amenv_step launches on
        t0 = time.perf_counter()
        run_timed()
        if not dry:
            ev1.record()
        sync()
        wall = time.perf_counter() - t0   # this rank's K steps, launch to completion; the job's time is the MAX over ranks (below)
        barrier()                         # closes the bracket; its own latency (an RCCL all-reduce) is not part of any rank's K steps
        dev_ms = ev0.elapsed_time(ev1) if not dry else wall * 1e3
        windows.append((amd.sharding.max_over_ranks(dist, wall, device), dev_ms))
    stats = env.stats()
    walls = sorted(w for w, _ in windows)
    wall = statistics.median(walls)
    dev_ms = statistics.median(d for _, d in windows)
    rccl_ranks = dist.get_world_size() if dist is not None else 1
    per_rank_ms = None
    if dist is not None:   # every rank's median window, for the record (gathered after the timing)
        t = torch.zeros(world, dtype=torch.float64, device=device)
        t[rank] = statistics.median(d for _, d in windows) / K
        dist.all_reduce(t)
        per_rank_ms = [float(x) for x in t.tolist()]

    total_envs = shard.global_envs
    value = amd.sharding.whole_job_rate(shard, K, wall)
    launch_mode = "graph" if use_graph else "eager"
    out = {
        "metric": BASELINE_METRIC if (args.vehicle == "hexa_arm" and n == 4096) else f"env-steps/sec (whole node), {args.vehicle} waypoint task at {n} envs/GPU",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic" if not dry else "dry-run (no kernel)",
        "config": {"workload": (f"{n} envs/GPU, " + ("hexacopter + 3-link arm coupled dynamics (19 states, 7-D action) + arm forward kinematics" if args.vehicle == "hexa_arm"
                                                       else f"{args.vehicle} 6-DOF rigid body") +
                                f", rotor mixer, RK4 dt=5ms, waypoint reward + reach/hold state machine + termination + "
                                f"auto-reset masks + {29 if args.vehicle == 'hexa_arm' else 20}-D obs, one launch per control step") + (" [dry-run: no kernel]" if dry else ""),
                   "baseline_config": baseline_config(args.vehicle, n, world),
                   "envs_per_gpu": n, "global_envs": total_envs, "vehicle": args.vehicle, "launch_mode": launch_mode,
                   "graph_steps": plan["chunk"], "graph_replays_per_window": plan["full"] + (1 if plan["rem"] else 0),
                   "eager_launches_per_window": 0 if use_graph else K, "kernel": env.kernel_name,
                   "actions": ("hover-centred N(1,0.1)/N(0,0.1) clipped" if args.actions == "hover" else "i.i.d. uniform over the action box") + ", pre-generated ring in HBM",
                   "preroll_steps": args.preroll, "parallelism": f"env-shard x{world}, no step-path collective"},
        "rccl_ranks": rccl_ranks, "per_rank_device_ms_per_step": per_rank_ms,
        "windows": {"count": len(windows), "statistic": "median", "wall_ms_per_step": [w / K * 1e3 for w in walls],
                    "device_ms_per_step_min_med_max": [min(d for _, d in windows) / K, dev_ms / K, max(d for _, d in windows) / K]},
        "device_ms_per_step": dev_ms / K,
        "episodes_finished_in_timed_windows_rank0": stats["episodes"],
        "episodes_finished_per_window_rank0": stats["episodes"] / max(1, len(windows)),
    }
    if not dry:
        bytes_step = env.bytes_per_env_step
        # per-launch kernel duration, isolated: HIP events stamped by the kernel's own dispatch (hipExtLaunchKernelGGL start/stop events on
        # the launch stream), each launch followed by a host sync; same workload continuing
        per = [env.step_timed(ring[i % GRAPH_CHUNK]) for i in range(256)]
        traffic = None  # HBM bytes per launch from the PMC passes (tools/pmc_traffic.py -> profiles/traffic.json), same workload
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                traffic = json.load(f).get(f"{args.vehicle}_{n}_{args.dtype}", {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        kernel_us = dev_ms / K * 1e3
        kn = env.kernel_name
        tiles = (n + 63) // 64
        waves = ((n + 3) // 4 * 2 if "team" in kn else (n + 15) // 16 * 2 if "step_kernel_quad" in kn else
                 tiles * (5 if "armk" in kn else 2 if "arm2w" in kn else ((4 if "KW=1,v2" in kn else 2) if "step_kernel_pw" in kn else 1)))
        out["roofline"] = {
            "bound": "hbm", "achieved": n * bytes_step / (kernel_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": n * bytes_step / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
            "bytes_per_env_step": bytes_step, "units_per_launch": n, "kernel_us": kernel_us,
            "timing": f"HIP events on the launch stream around each window of {K} back-to-back {launch_mode} launches (median of {len(windows)} windows; launch gaps included: an upper bound on the kernel time)",
            "kernel_us_isolated": sum(per) / len(per), "kernel_us_isolated_min": min(per),
            "isolated_timing": "mean of 256 single launches, start/stop events stamped by the kernel dispatch (amenv_step_timed), each followed by a host sync",
            "note": f"{n} envs = {waves} wavefronts on 256 CUs (1024 SIMDs), {n * bytes_step / 1e6:.2f} MB algorithmic per launch"
                    + ("; latency-bound by construction at this batch: the launch is as long as one wave's instruction stream plus the ~1.7 us dependent-launch floor, "
                       "so the HBM fraction is small by design -- DESIGN.md section 6" if n <= 65536 else "")}
        if not args.no_extras and rank == 0 and world == 1:   # single-GPU measurements; the N > 1 runs report the scaling line only
            out["extras"] = extras(args, amd, torch, env, ring, n, device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dry:
        out["cpu_baseline"] = cpu_baseline(args, amd, n)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    env.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def extras(args, amd, torch, env, ring, n, device):
    """Rank-0 extras outside the headline: T steps per launch (amenv_rollout), and SURVEY 8d's protocol (T = 2048 steps after 64 warm-up
    steps, median of 5) for action sets A / B and, as set C, the closed loop obs -> policy (fused HIP forward) -> clip -> step."""
    import statistics

    def timed(fn, reps):
        out = []
        for _ in range(reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1))
        return statistics.median(out)

    ex = {}
    T = GRAPH_CHUNK
    env.rollout(ring, want_obs=True)   # warm (module load, allocations)
    ms = timed(lambda: [env.rollout(ring, want_obs=True) for _ in range(8)], 5)
    ex["rollout_kernel"] = {"steps_per_launch": T, "env_steps_per_s": 8 * T * n / (ms * 1e-3), "us_per_step": ms * 1e3 / (8 * T),
                            "what": "amenv_rollout: 64 open-loop steps per launch, state in registers between steps, all per-step outputs written"}
    S = 2048
    prot = {}
    for tag, kind in (("A_uniform", "uniform"), ("B_hover", "hover")):
        e2 = amd.GpuWaypointEnv(n, device=device.index, vehicle=args.vehicle, seed=0, dtype=args.dtype, kernel=args.kernel)
        e2.reset()
        r2 = make_actions(torch, kind, n, GRAPH_CHUNK, device, seed=99, act_dim=e2.act_dim)
        for t in range(GRAPH_CHUNK):
            e2.step(r2[t])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(GRAPH_CHUNK):
                e2.step(r2[t])
        e2.stats(reset=True)
        ms = timed(lambda: [g.replay() for _ in range(S // GRAPH_CHUNK)], 5)
        st = e2.stats()
        prot[tag] = {"env_steps_per_s": S * n / (ms * 1e-3), "us_per_step": ms * 1e3 / S, "episodes_per_1000_env_steps": 1000.0 * st["episodes"] / (5.0 * S * n)}
        e2.close()
    # C: policy in the loop (the reference's architecture 128-64-64 tanh, random-init weights of the vehicle's obs / action dims), twice: one launch per
    # operation (fused fp32 policy forward, clip, step) and ONE launch per 64 steps (amenv_rollout_policy: bf16 actor / critic on the matrix cores,
    # sampling, clip, env step; state, constants and weights in registers -- the opt-in rollout mode of amd.PPO(fused_rollout=True)).  For the
    # benchmark's vehicle and, as `reference_vehicle`, for the quadrotor the reference trains (v2/rl_train.py:24,38-56).
    def closed_loop_pair(vehicle):
        out = {}
        e3 = amd.GpuWaypointEnv(n, device=device.index, vehicle=vehicle, seed=0, dtype=args.dtype, kernel=args.kernel if vehicle == args.vehicle else "auto")
        pol = amd.ActorCritic(e3.obs_dim, e3.act_dim).to(device)
        pol.flatten_()   # parameters as views of one flat buffer: what the fused forward kernel reads
        obs = e3.reset()
        lo, hi = pol.action_low, pol.action_high

        def closed_loop(k):
            o = e3.obs
            with torch.no_grad():
                for _ in range(k):
                    a = torch.max(torch.min(pol.actor(o), hi), lo)
                    o = e3.step(a)[0]

        closed_loop(8)
        ms = timed(lambda: closed_loop(S // 8), 5)
        with torch.no_grad():
            fused_fwd = bool(pol.fused_ok(obs))
        out["C_policy_closed_loop"] = {"env_steps_per_s": (S // 8) * n / (ms * 1e-3), "us_per_step": ms * 1e3 / (S // 8), "fused_policy_kernel": fused_fwd,
                                       "weights": "random init (reference architecture 128-64-64 tanh)", "step_kernel": e3.kernel_name}
        e3.close()
        if args.dtype == "f32":
            e4 = amd.GpuWaypointEnv(n, device=device.index, vehicle=vehicle, seed=0, kernel=args.kernel if vehicle == args.vehicle else "auto")
            e4.reset()
            Tf = GRAPH_CHUNK
            f = dict(dtype=torch.float32, device=device)
            bo, ba = torch.zeros(Tf + 1, n, e4.obs_dim, **f), torch.zeros(Tf, n, e4.act_dim, **f)
            bl, bv, br = torch.zeros(Tf, n, **f), torch.zeros(Tf, n, **f), torch.zeros(Tf, n, **f)
            bd = torch.zeros(Tf, n, dtype=torch.uint8, device=device)
            draw = [0]

            def fused():
                for _ in range(S // Tf):
                    e4.rollout_policy(pol.flat_param, Tf, 0, draw[0], bo, ba, bl, bv, br, bd)
                    draw[0] += Tf

            fused()
            ms = timed(fused, 5)
            out["C_policy_closed_loop_one_launch"] = {"env_steps_per_s": S * n / (ms * 1e-3), "us_per_step": ms * 1e3 / S, "steps_per_launch": Tf,
                                                      "policy_arithmetic": "bf16 MFMA, fp32 accumulate",
                                                      "what": "amenv_rollout_policy: value + sampled action + log-prob + env step per step, rollout-buffer rows written"}
            e4.close()
        return out

    prot.update(closed_loop_pair(args.vehicle))
    if args.vehicle != "quad":
        prot["reference_vehicle"] = {"vehicle": "quad (v2/simul_files/model/params.py: the vehicle the reference trains)", "envs": n, **closed_loop_pair("quad")}
    ex["survey_8d"] = {"steps": S, "warmup": GRAPH_CHUNK, "repeats": 5, "statistic": "median", **prot}
    # row f3 (config 5's update): one PPO minibatch step of the reference's policy at 65,536 samples -- forward, SB3 loss, backward and all weight
    # gradients of both MLPs in the fused HIP kernel (csrc/amenv_mlp_train.hpp: fp32 products from bf16 MFMAs on exactly split operands) + its
    # prologue and gradient-reduce launches; synthetic samples of the benchmark vehicle's shapes
    try:
        from rl_aerial_manipulator_amd.ppo import MinibatchStep
        nb = 65536
        e5 = amd.GpuWaypointEnv(64, device=device.index, vehicle=args.vehicle, seed=0)
        D5, A5 = e5.obs_dim, e5.act_dim
        e5.close()
        pol = amd.ActorCritic(D5, A5).to(device).flatten_()
        opt = torch.optim.Adam([pol.flat_param.requires_grad_(True)], lr=1e-3)
        gen = torch.Generator(device=device).manual_seed(5)
        r = lambda *sh: torch.randn(*sh, device=device, generator=gen)  # noqa: E731
        b_obs, b_act, b_olp, b_adv, b_ret = r(nb, D5), r(nb, A5), r(nb) * 0.1 - 5.0, r(nb), r(nb)
        step = MinibatchStep(pol, opt, use_graph=False, fused_mlp=True)
        for _ in range(3):
            step._forward_backward(b_obs, b_act, b_olp, b_adv, b_ret)
        ms = timed(lambda: [step._forward_backward(b_obs, b_act, b_olp, b_adv, b_ret) for _ in range(20)], 5)
        ex["ppo_minibatch_step"] = {"samples": nb, "obs_dim": D5, "act_dim": A5, "us_per_forward_backward": ms * 1e3 / 20,
                                    "samples_per_s": 20 * nb / (ms * 1e-3), "fused_kernel": bool(step.fused_mlp),
                                    "what": "prologue (advantage sums, weight packing) + ppo_mlp_fused_kernel + fixed-order gradient reduce, back to back"}
    except Exception as exc:   # an extra: never takes the headline line down
        ex["ppo_minibatch_step"] = {"error": repr(exc)}
    return ex


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_children(args)
    return main_worker(args)


if __name__ == "__main__":
    sys.exit(main())
