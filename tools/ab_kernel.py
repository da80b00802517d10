#!/usr/bin/env python3
"""A/B timing of libamenv.so builds on the GPU box: interleaved rounds in ONE process per variant is not
possible across .so files, so each variant runs in its own child process, rounds interleaved by the parent.

  python tools/ab_kernel.py --libs a.so b.so --envs 4096 32768 1048576 --rounds 3
Child mode (--child) prints one JSON line: kernel_us (dispatch-stamped), loop_us (hipGraph replay) per env count.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import torch

    import rl_aerial_manipulator_amd as amd
    out = {}
    for n in args.envs:
        env = amd.GpuWaypointEnv(n, vehicle=args.vehicle, seed=0, block_size=args.block_size)
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        ring = torch.randn(64, n, env.act_dim, device="cuda", generator=g) * (0.1 if args.ring == "random" else 0.0)
        ring[..., 0] += 1.0
        ring[..., 4:] *= 3.0
        ring = ring.clamp(min=-1, max=2).contiguous()
        for t in range(64):
            env.step(ring[t])
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for t in range(64):
                env.step(ring[t])
        for _ in range(8):
            graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(4, min(64, int(2e8 / (64 * n))))
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        loop_us = e0.elapsed_time(e1) * 1e3 / (reps * 64)
        per = sorted(env.step_timed(ring[i % 64]) for i in range(256))
        out[str(n)] = dict(loop_us=round(loop_us, 3), kernel_us_med=round(per[128], 3), kernel_us_min=round(per[0], 3), kernel=env.kernel_name)
        env.close()
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="+", default=[])
    ap.add_argument("--envs", nargs="+", type=int, default=[4096, 32768, 1048576])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--vehicle", default="hexa")
    ap.add_argument("--block-size", type=int, default=0)
    ap.add_argument("--block-sizes", nargs="+", type=int, default=None)
    ap.add_argument("--ring", default="random", choices=["random", "hover"])
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        return child(args)
    variants = []
    for lib in args.libs:
        for bs in (args.block_sizes or [args.block_size]):
            variants.append((lib, bs))
    res = {v: [] for v in variants}
    for r in range(args.rounds):
        for lib, bs in variants:
            env = dict(os.environ, AMENV_LIB=os.path.abspath(lib))
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--vehicle", args.vehicle, "--block-size", str(bs), "--ring", args.ring, "--envs"] + [str(n) for n in args.envs]
            o = subprocess.run(cmd, env=env, capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("{")]
            if not line:
                print("FAILED", lib, bs, o.stderr[-500:])
                continue
            res[(lib, bs)].append(json.loads(line[-1]))
    for (lib, bs), runs in res.items():
        for n in args.envs:
            lo = [r[str(n)]["loop_us"] for r in runs]
            ke = [r[str(n)]["kernel_us_med"] for r in runs]
            if lo:
                print(f"{os.path.basename(lib):28s} bs={bs:4d} N={n:8d}  loop_us min {min(lo):8.3f} med {sorted(lo)[len(lo)//2]:8.3f}   kernel_us med {sorted(ke)[len(ke)//2]:8.3f}  -> {n/min(lo)*1e6:.3e} env-steps/s", flush=True)


if __name__ == "__main__":
    main()
