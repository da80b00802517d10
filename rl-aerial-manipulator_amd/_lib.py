"""ctypes binding of libamenv.so (include/amenv.h).  There is NO CPU fallback: if the HIP
library is missing or no gfx950 device is visible, creation fails loudly."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# AMENV_LIB lets tools/ab_kernel.py load an alternative build of the SAME library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("AMENV_LIB") or os.path.join(HERE, "libamenv.so")

MAX_ROTORS, MAX_WAYPOINTS, MAX_JOINTS = 8, 4, 3
ABI_VERSION = 2
F32, F64 = 0, 1
FLAG_AUTO_RESET, FLAG_NAN_GUARD = 1, 2
INFO_TERMINATED, INFO_TRUNCATED, INFO_SUCCESS, INFO_STOPPED, INFO_CRASHED, INFO_OOB, INFO_NONFINITE, INFO_WAS_RESET = (1 << i for i in range(8))
F_FINAL_YAW, F_LAST_DISTANCE, F_EP_RETURN, F_WP0 = 13, 14, 15, 16
I_STEP, I_COUNTER, I_FLAGS, I_EPISODE, I_NFIELDS = 0, 1, 2, 3, 4
FLAGBIT_FWR, FLAGBIT_COUNTER_ACTIVE = 256, 512
TASK_V2_SCALED20, TASK_V1_SCALED17, TASK_V1_RAW17 = 0, 1, 2
KERNEL_AUTO, KERNEL_LANE, KERNEL_HELPER, KERNEL_TEAM, KERNEL_STAGED = 0, 1, 2, 3, 4
KERNELS = {"auto": KERNEL_AUTO, "lane": KERNEL_LANE, "helper": KERNEL_HELPER, "team": KERNEL_TEAM, "staged": KERNEL_STAGED}
EE_TASK_BASE, EE_TASK_TOOL = 0, 1
TASKS = {"v2": TASK_V2_SCALED20, "v1_scaled": TASK_V1_SCALED17, "v1_raw": TASK_V1_RAW17}


class Vehicle(C.Structure):
    _fields_ = [
        ("n_rotors", C.c_int32), ("n_joints", C.c_int32), ("mass", C.c_double), ("g", C.c_double),
        ("inertia", C.c_double * 9), ("inv_inertia", C.c_double * 9),
        ("alloc", C.c_double * (MAX_ROTORS * 4)), ("mix", C.c_double * (4 * MAX_ROTORS)),
        ("t_min", C.c_double * MAX_ROTORS), ("t_max", C.c_double * MAX_ROTORS), ("moment_scale", C.c_double),
        ("joint_origin", C.c_double * (MAX_JOINTS * 3)), ("joint_axis", C.c_double * (MAX_JOINTS * 3)),
        ("link_mass", C.c_double * MAX_JOINTS), ("link_com", C.c_double * (MAX_JOINTS * 3)),
        ("link_inertia", C.c_double * (MAX_JOINTS * 9)),
        ("joint_kp", C.c_double), ("joint_kd", C.c_double), ("joint_acc_max", C.c_double), ("joint_reserved", C.c_double),
        ("joint_limit", C.c_double * (MAX_JOINTS * 2)), ("tool_offset", C.c_double * 3),
    ]


class Task(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("num_waypoints", C.c_int32), ("max_episode_steps", C.c_int32),
        ("counter_limit", C.c_int32), ("rk4_substeps", C.c_int32), ("ee_task", C.c_int32), ("dt", C.c_double),
        ("traj_sin", C.c_double * MAX_WAYPOINTS), ("traj_cos", C.c_double * MAX_WAYPOINTS),
    ]


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("num_envs", C.c_int32), ("dtype", C.c_int32),
        ("flags", C.c_uint32), ("block_size", C.c_int32), ("step_kernel", C.c_int32), ("reserved1", C.c_int32), ("seed", C.c_uint64), ("env_id_offset", C.c_int64),
        ("vehicle", Vehicle), ("task", Task),
    ]


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("steps", "episodes", "terminated", "truncated", "success", "crashed", "oob", "nonfinite", "length_sum")] + [
        ("return_sum_q10", C.c_int64)]


class PidParams(C.Structure):
    _fields_ = [("dt", C.c_double), ("mass", C.c_double), ("g", C.c_double), ("max_integral", C.c_double), ("gain", C.c_double * 18)]


class PidPolicyParams(C.Structure):
    _fields_ = [("pid", PidParams), ("speed", C.c_double), ("moment_scale", C.c_double), ("inertia_ratio", C.c_double * 3),
                ("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("tool_mode", C.c_int32), ("reserved", C.c_int32)]


class AmenvError(RuntimeError):
    pass


# every symbol include/amenv.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "amenv_version": (C.c_char_p, []),
    "amenv_default_config": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(Config)]),
    "amenv_config_set_task": (C.c_int, [C.POINTER(Config), C.c_int32]),
    "amenv_dims": (C.c_int, [C.POINTER(Config)] + [C.POINTER(C.c_int32)] * 4),
    "amenv_bytes_per_env_step": (C.c_int64, [C.POINTER(Config)]),
    "amenv_create": (C.c_int, [C.POINTER(Config), C.c_int, C.POINTER(_P)]),
    "amenv_destroy": (C.c_int, [_P]),
    "amenv_last_error": (C.c_char_p, [_P]),
    "amenv_set_seed": (C.c_int, [_P, C.c_uint64]),
    "amenv_reset": (C.c_int, [_P, _P, _P, _P]),
    "amenv_step": (C.c_int, [_P] * 10),
    "amenv_step_timed": (C.c_int, [_P] * 10 + [C.POINTER(C.c_float)]),
    "amenv_rollout": (C.c_int, [_P, C.c_int32] + [_P] * 6),
    "amenv_rollout_policy": (C.c_int, [_P, C.c_int32, _P, C.c_uint64, C.c_uint32] + [_P] * 9),
    "amenv_get_state": (C.c_int, [_P] * 4),
    "amenv_set_state": (C.c_int, [_P] * 4),
    "amenv_observe": (C.c_int, [_P, _P, _P]),
    "amenv_ee_position": (C.c_int, [_P, _P, _P]),
    "amenv_stats_read": (C.c_int, [_P, C.POINTER(Stats), C.c_int, _P]),
    "amenv_kernel_name": (C.c_char_p, [_P]),
    "amenv_calibration_copy": (C.c_int, [_P, _P, C.c_size_t, C.c_int32, _P]),
    "amenv_obsnorm_create": (C.c_int, [C.c_int32, C.c_int, C.POINTER(_P)]),
    "amenv_obsnorm_destroy": (C.c_int, [_P]),
    "amenv_obsnorm_update": (C.c_int, [_P, _P, C.c_int64, _P]),
    "amenv_obsnorm_apply": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, C.c_double, _P]),
    "amenv_obsnorm_get": (C.c_int, [_P, _P, _P, _P, _P]),
    "amenv_obsnorm_set": (C.c_int, [_P, _P, _P, C.c_double, _P]),
    "amenv_gae": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int64, C.c_float, C.c_float, _P]),
    "amenv_policy_forward": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int64, _P, _P, _P]),
    "amenv_policy_forward_mfma": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int64, _P, _P, _P, _P]),
    "amenv_ppo_workspace_bytes": (C.c_size_t, []),
    "amenv_ppo_loss_grad": (C.c_int, [_P] * 7 + [C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "amenv_ppo_mlp_workspace_bytes": (C.c_size_t, []),
    "amenv_ppo_mlp_step": (C.c_int, [_P, C.c_int32, C.c_int32] + [_P] * 6 + [C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int32, _P, _P, _P, _P]),
    "amenv_ppo_adam_step": (C.c_int, [_P] * 5 + [C.c_int64, _P, _P, _P, _P]),
    "amenv_arm_rhs": (C.c_int, [C.POINTER(Config), C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int64, _P]),
    "amenv_pid_default_params": (C.c_int, [C.POINTER(PidParams)]),
    "amenv_pid_run": (C.c_int, [C.POINTER(PidParams), C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "amenv_minsnap_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "amenv_minsnap_solve": (C.c_int, [C.c_int32, C.c_int64, C.c_double, _P, _P, _P, _P, _P, _P]),
    "amenv_minsnap_eval": (C.c_int, [C.c_int32, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P]),
    "amenv_pid_policy": (C.c_int, [C.POINTER(PidPolicyParams), C.c_int32, _P, _P, _P, _P, C.c_int64, _P]),
    "amenv_gaussian_act": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_uint64, C.c_uint32, C.c_int64, _P]),
}

_lib = None


def load():
    """dlopen libamenv.so and bind every symbol.  Raises if the library was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AmenvError(f"{LIB_PATH} not found: build it with __graft_entry__.build() "
                             f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def default_config(vehicle="quad", num_envs=1, task="v2"):
    cfg = Config()
    rc = load().amenv_default_config(vehicle.encode(), num_envs, C.byref(cfg))
    if rc == 0 and task != "v2":
        rc = load().amenv_config_set_task(C.byref(cfg), TASKS[task])
    if rc != 0:
        raise AmenvError(load().amenv_last_error(None).decode())
    return cfg
