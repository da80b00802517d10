"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/amenv.h declares,
its host-side helpers agree with the oracle's independent constants, and it fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rl_aerial_manipulator_amd as amd
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "amenv.h")).read()
    declared = set(re.findall(r"\b(amenv_[a-z_]+)\s*\(", hdr)) - {"amenv_stats"}
    assert declared == set(amd._lib.SYMBOLS), declared ^ set(amd._lib.SYMBOLS)
    lib = C.CDLL(amd._lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in amd._lib.load().amenv_version()


def test_struct_layout_matches_header():
    cfg = amd._lib.default_config("quad", 4096)
    assert cfg.struct_size == C.sizeof(amd._lib.Config) == C.sizeof(O.Config)
    assert cfg.abi_version == 2 and cfg.num_envs == 4096 and cfg.flags == amd._lib.FLAG_AUTO_RESET
    assert cfg.step_kernel == amd._lib.KERNEL_AUTO and cfg.task.ee_task == amd._lib.EE_TASK_BASE
    arm = amd._lib.default_config("hexa_arm", 8)
    assert arm.task.ee_task == amd._lib.EE_TASK_TOOL and list(arm.vehicle.tool_offset) == [-0.0015, 0.003, -0.125]
    od = C.c_int32(); ad = C.c_int32()
    amd._lib.load().amenv_dims(C.byref(arm), C.byref(od), C.byref(ad), None, None)
    assert (od.value, ad.value) == (29, 7)
    nf = C.c_int32()
    for nj in (1, 2):       # an n-link arm with fewer joints keeps ITS dimensions at the C ABI: 4 + n actions, 20 + 2 n + 3 observations, 2 n joint fields
        arm.vehicle.n_joints = nj
        amd._lib.load().amenv_dims(C.byref(arm), C.byref(od), C.byref(ad), C.byref(nf), None)
        assert (od.value, ad.value, nf.value) == (23 + 2 * nj, 4 + nj, 19 + 2 * nj)


def test_pid_default_params_are_the_reference_gains():
    """amenv_pid_default_params (host only) against the gains committed in `PID Controller/pid_controller.py:16-21,34` as baselines.py
    restates them; bad arguments to the launchers are refused before anything touches a device."""
    from rl_aerial_manipulator_amd.baselines import GAINS, MAX_INTEGRAL
    p = amd._lib.PidParams()
    assert amd._lib.load().amenv_pid_default_params(C.byref(p)) == 0 and amd._lib.load().amenv_pid_default_params(None) == -1
    assert (p.dt, p.mass, p.g, p.max_integral) == (0.01, 0.18, 9.81, MAX_INTEGRAL)
    for k, name in enumerate(("x", "y", "z", "phi", "theta", "psi")):
        assert tuple(p.gain[3 * k:3 * k + 3]) == GAINS[name]
    assert C.sizeof(amd._lib.PidParams) == 22 * 8 and C.sizeof(amd._lib.PidPolicyParams) == 22 * 8 + 5 * 8 + 16
    L = amd._lib.load()
    assert L.amenv_minsnap_workspace_bytes(4) == 32 * 40 * 8 and L.amenv_minsnap_workspace_bytes(0) == 0 and L.amenv_minsnap_workspace_bytes(17) == 0
    assert L.amenv_pid_run(C.byref(p), 0, None, None, None, None, None, None, 4, None) == -1
    assert L.amenv_minsnap_solve(0, 1, 1.2, None, None, None, None, None, None) == -1
    assert L.amenv_minsnap_eval(4, 0, None, None, None, None, None, None, 1, None, None) == -1
    assert L.amenv_pid_policy(None, 0, None, None, None, None, 1, None) == -1
    arm = amd._lib.default_config("hexa_arm", 1)
    assert L.amenv_arm_rhs(C.byref(arm), 2, 1, None, None, None, None, 4, None) == -1 and L.amenv_arm_rhs(C.byref(arm), 1, 1, None, None, None, None, 4, None) == -1


def test_default_quad_equals_oracle_constants():
    """Product-side constants (amenv_default_config) vs the oracle's independent restatement of params.py."""
    a = amd._lib.default_config("quad", 1)
    b = O.reference_quad_config(1)
    for f in ("n_rotors", "mass", "g", "moment_scale"):
        assert getattr(a.vehicle, f) == getattr(b.vehicle, f)
    for f in ("inertia", "inv_inertia", "alloc", "mix", "t_min", "t_max"):
        np.testing.assert_allclose(np.array(getattr(a.vehicle, f)), np.array(getattr(b.vehicle, f)), rtol=1e-13, atol=1e-15)
    for f in ("variant", "num_waypoints", "max_episode_steps", "counter_limit", "rk4_substeps", "dt"):
        assert getattr(a.task, f) == getattr(b.task, f)
    assert list(a.task.traj_sin) == list(b.task.traj_sin) and list(a.task.traj_cos) == list(b.task.traj_cos)


def test_hexa_config_is_consistent():
    c = amd._lib.default_config("hexa", 1)
    v = c.vehicle
    assert v.n_rotors == 6
    A = np.array(v.mix[:24]).reshape(4, 6); P = np.array(v.alloc[:24]).reshape(6, 4)
    np.testing.assert_allclose(A @ P, np.eye(4), atol=1e-12)        # allocation is a right inverse of the mixer
    hover = P @ np.array([v.mass * v.g, 0, 0, 0])
    np.testing.assert_allclose(hover, v.mass * v.g / 6, rtol=1e-12)  # symmetric airframe: equal thrusts at hover
    assert (hover > np.array(v.t_min[:6])).all() and (hover < np.array(v.t_max[:6])).all()
    # hover rotor speed ~459 rad/s (hexacopter_description/test_motors.py:45)
    assert abs(np.sqrt(hover[0] / 2.11e-5) - 459) < 2


def test_bytes_per_env_step_formula():
    cfg = amd._lib.default_config("quad", 1)
    L = amd._lib.load()
    assert L.amenv_bytes_per_env_step(C.byref(cfg)) == 4 * (13 + 3 + 3) + 12 + 16 + 4 * 15 + 12 + 80 + 4 + 1 + 4 == 265
    cfg.dtype = amd._lib.F64
    assert L.amenv_bytes_per_env_step(C.byref(cfg)) == 8 * 19 + 12 + 16 + 8 * 15 + 12 + 80 + 8 + 1 + 4


def test_bad_config_is_refused():
    L = amd._lib.load()
    h = C.c_void_p()
    cfg = amd._lib.default_config("quad", 16)
    cfg.struct_size = 12
    assert L.amenv_create(C.byref(cfg), 0, C.byref(h)) == -1 and b"struct_size" in L.amenv_last_error(None)
    cfg = amd._lib.default_config("quad", 16)
    cfg.task.num_waypoints = 9
    assert L.amenv_create(C.byref(cfg), 0, C.byref(h)) == -1
    # workgroup sizes that do not divide the 256-lane allocation granule are refused (192 used to be accepted and ran past the blob)
    for bs, ok in ((64, True), (128, True), (256, True), (192, False), (32, False), (512, False)):
        cfg = amd._lib.default_config("quad", 4096); cfg.block_size = bs
        rc = L.amenv_create(C.byref(cfg), 0, C.byref(h))
        assert (b"block_size" in L.amenv_last_error(None)) == (not ok), (bs, rc, L.amenv_last_error(None))
    cfg = amd._lib.default_config("quad", 16); cfg.step_kernel = 9
    assert L.amenv_create(C.byref(cfg), 0, C.byref(h)) == -1 and b"step_kernel" in L.amenv_last_error(None)
    with pytest.raises(amd.AmenvError):
        amd._lib.default_config("octo", 1)


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(amd.AmenvError, match="no CPU"):
        amd.GpuWaypointEnv(64)
