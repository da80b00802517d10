"""rl-aerial-manipulator_amd -- MI355X-native batched replacement for the one hot path of
LahiruCooray/rl-aerial-manipulator: `WaypointQuadEnv.step()/reset()` + the `simul_files`
rigid-body integrator.  HIP kernels behind a C ABI (include/amenv.h);
this package is the thin host-side mirror of the reference's Python interface.

The directory name carries a hyphen (project convention), so import it through the alias
module at the repo root:  `import rl_aerial_manipulator_amd as amd`.
"""
from . import _lib, sharding
from ._lib import AmenvError
from . import vec_env
from .gpu_env import GpuWaypointEnv
from .vec_env import GpuVecEnv
from .obs_norm import GpuVecNormalize, ObsNormalizer
from . import baselines, ppo
from .baselines import MinSnapTrajectory, PidController, PidWaypointPolicy
from .ppo import PPO, ActorCritic, evaluate_policy, clone_pid_policy

__all__ = ["GpuWaypointEnv", "GpuVecEnv", "GpuVecNormalize", "ObsNormalizer", "PPO", "ActorCritic", "evaluate_policy", "clone_pid_policy", "ppo", "baselines", "PidController", "MinSnapTrajectory", "PidWaypointPolicy", "vec_env", "AmenvError", "_lib", "sharding"]
