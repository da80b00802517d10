"""Row f4 on the MI355X: the PID + minimum-snap baseline in closed loop on the HIP environment, everything resident on the GPU."""
import pytest
import torch

import rl_aerial_manipulator_amd as amd
from rl_aerial_manipulator_amd.baselines import PidWaypointPolicy

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("vehicle,floor", [("quad", 0.8), ("hexa", 0.75)])
def test_pid_baseline_flies_the_gpu_env(vehicle, floor):
    env = amd.GpuWaypointEnv(1024, vehicle=vehicle, seed=3)
    pol = PidWaypointPolicy.for_env(env)
    obs = env.reset()
    env.stats(reset=True)
    done = None
    for _ in range(1700):
        obs, _, done, _ = env.step(pol.predict(obs, done))
    s = env.stats()
    assert s["episodes"] > 1000 and s["success"] / s["episodes"] > floor
    assert s["nonfinite"] == 0


def test_pid_policy_is_sync_free_and_on_device():
    env = amd.GpuWaypointEnv(256, seed=1)
    pol = PidWaypointPolicy.for_env(env)
    a = pol.predict(env.reset())
    assert a.is_cuda and a.dtype == torch.float32 and a.shape == (256, 4)
    assert bool((a[:, 0] >= 0).all()) and bool((a[:, 0] <= 2).all()) and bool((a[:, 1:].abs() <= 1).all())
    # the first action of an episode at rest on the trajectory start is hover thrust, no moments
    assert torch.allclose(a[:, 0], torch.ones(256, device=a.device), atol=1e-4) and float(a[:, 1:].abs().max()) < 1e-3
