"""rl_env_scaledObs -- single-environment facade for `initial-implementation-v1/rl_env_scaledObs.py` (scaled observations).

`from rl_env_scaledObs import WaypointQuadEnv` (v1/rl_train.py:4, v1/runsim_scaledObs.py:3) resolves to this file when `rl-aerial-manipulator_amd/compat/v1/` is on
PYTHONPATH.  Same class as the v2 facade (compat/rl_env_scaledObs.py) with the v1 task selected: 17-D observation
(v1/rl_env_scaledObs.py:65-83), 1..2 waypoints per episode, termination on the final waypoint, 1200-step limit.
"""
import importlib.util
import os

_base_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rl_env_scaledObs.py")
_spec = importlib.util.spec_from_file_location("_amenv_facade_v2", _base_path)   # the v2 facade under a private module name
_base = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_base)


class WaypointQuadEnv(_base.WaypointQuadEnv):
    TASK = "v1_scaled"
