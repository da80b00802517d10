"""The reference's PID demo (initial-implementation-v2/PID Controller/runsim.py:26-49: helix waypoints, minimum-snap trajectory, cascaded PID,
Quadcopter.update in a loop) with every piece on the MI355X, for N vehicles at once:

    python examples/runsim_gpu.py [--vehicles 4096] [--steps 1200]

trajGen3D.get_MST_coefficients -> amenv_minsnap_solve, generate_trajectory -> amenv_minsnap_eval, pid_controller.run -> amenv_pid_run,
Quadcopter.update -> amenv_step (fp64 build of the quadrotor env at the demo's 10 ms period, F and M handed over as actions).  Each
vehicle flies its own helix (radius / height jittered around the demo's (10, 5)); prints the tracking error the way the demo would plot it.
"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def helix_waypoints(radius, height, n=5):
    """trajGen3D.get_helix_waypoints(radius, height) -- `n` waypoints on one turn of a helix, starting at (0.5, 0, 0) like runsim.py:37."""
    import torch
    k = torch.arange(n, dtype=torch.float64)
    ang = k / (n - 1) * 2.0 * math.pi
    w = torch.stack([radius.unsqueeze(-1) * torch.cos(ang) - radius.unsqueeze(-1) + 0.5, radius.unsqueeze(-1) * torch.sin(ang),
                     height.unsqueeze(-1) * k / (n - 1)], -1)
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vehicles", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--speed", type=float, default=1.2)           # runsim.py:27
    a = ap.parse_args()
    import torch
    import rl_aerial_manipulator_amd as amd
    n, dt = a.vehicles, 0.01                                       # runsim.py:28
    g = torch.Generator().manual_seed(0)
    radius = 2.0 + 8.0 * torch.rand(n, generator=g, dtype=torch.float64)
    height = 2.0 + 3.0 * torch.rand(n, generator=g, dtype=torch.float64)
    wp = helix_waypoints(radius, height).cuda()                    # [n, 5, 3]
    traj = amd.MinSnapTrajectory(wp, a.speed)                      # one launch pair for all n trajectories
    cfg = amd._lib.default_config("quad", n)
    cfg.dtype, cfg.flags, cfg.task.dt = amd._lib.F64, 0, dt        # no auto-reset: the task's flags never touch the dynamics
    env = amd.GpuWaypointEnv(n, config=cfg)
    env.reset()
    f, i = env.get_state()
    f[:13] = 0.0; f[6] = 1.0; f[0:3] = wp[:, 0].T                   # at rest on the first waypoint
    f[amd._lib.F_WP0:amd._lib.F_WP0 + 3] = 50.0
    env.set_state(f, i)
    pid = amd.PidController(n, dt, device="cuda", dtype=torch.float64)
    worst = torch.zeros(n, dtype=torch.float64, device="cuda")
    for k in range(a.steps):
        t = torch.full((n,), k * dt, dtype=torch.float64, device="cuda")
        des = traj.evaluate(t)                                     # [n, 11]
        f, _ = env.get_state()
        F, M, _ = pid.run_state(f[:13].T.contiguous(), des)
        act = torch.cat([(F / (0.18 * 9.81)).unsqueeze(-1), M / 0.1], -1).to(torch.float32)
        env.step(act)
        f, _ = env.get_state()
        worst = torch.maximum(worst, (f[0:3].T - des[:, 0:3]).abs().amax(-1))
    end_err = (f[0:3].T - wp[:, -1]).abs().amax(-1)
    flight = traj.S[:, -1]
    done = flight < a.steps * dt
    print(f"{n} vehicles, {a.steps} control steps of {dt} s; trajectories last {float(flight.min()):.1f} .. {float(flight.max()):.1f} s")
    print(f"worst tracking error per vehicle: median {float(worst.median()):.3f} m, max {float(worst.max()):.3f} m")
    if bool(done.any()):
        print(f"{int(done.sum())} vehicles finished their trajectory: distance to the last waypoint median {float(end_err[done].median()):.4f} m, max {float(end_err[done].max()):.4f} m")


if __name__ == "__main__":
    main()
