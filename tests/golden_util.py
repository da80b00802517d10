"""Helpers shared by the oracle (CPU) and HIP (GPU) parity tests: golden-vector loading."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EPISODES = ["policy_ep0", "policy_ep1", "policy_ep2", "policy_ep3", "openloop_1000", "crash", "oob", "timelimit",
            "saturation", "reach_and_leave"]


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def teacher_forced_inputs(d):
    """Arrays, one row per step t, holding the reference env's state BEFORE step t."""
    T = d["actions"].shape[0]
    return dict(
        T=T,
        state=d["state"][:T], waypoints=d["waypoints"], final_yaw=float(d["final_yaw"]),
        last_distance=d["var_last_distance"][:T], waypoint_index=d["var_waypoint_index"][:T], fwr=d["var_fwr"][:T],
        counter=d["var_counter"][:T], counter_activated=d["var_counter_activated"][:T], current_step=d["var_current_step"][:T],
        actions=d["actions"],
    )


V1_EPISODES = ["reach1", "reach2", "flythrough", "crash", "oob", "timelimit"]


def fill_blob(fstate, istate, d, K=1, per_env_k=False):
    """Write the teacher-forced inputs of episode d into SoA blobs (column t = state before step t).
    per_env_k: v1 envs carry the episode's waypoint count in flags bits 4-7."""
    from oracle import oracle as O  # field indices only

    tf = teacher_forced_inputs(d)
    T = tf["T"]
    fstate[0:13, :T] = tf["state"].T
    fstate[O.F_FINAL_YAW, :T] = tf["final_yaw"]
    ld = tf["last_distance"].astype(np.float64)
    fstate[O.F_LAST_DISTANCE, :T] = np.where(np.isnan(ld), -1.0, ld)
    fstate[O.F_EP_RETURN, :T] = 0.0
    wp = np.asarray(tf["waypoints"], np.float64).reshape(-1)
    fstate[O.F_WP0:O.F_WP0 + wp.size, :T] = wp[:, None]
    istate[O.I_STEP, :T] = tf["current_step"]
    istate[O.I_COUNTER, :T] = tf["counter"]
    kbits = (int(np.asarray(tf["waypoints"]).reshape(-1, 3).shape[0]) << 4) if per_env_k else 0
    istate[O.I_FLAGS, :T] = kbits | (tf["waypoint_index"].astype(np.int32) & 15) | np.where(tf["fwr"], O.FLAGBIT_FWR, 0) | np.where(
        tf["counter_activated"], O.FLAGBIT_COUNTER_ACTIVE, 0)
    istate[O.I_EPISODE, :T] = 1
    return tf


def report_flips(where, flips, out_of):
    """fp32 threshold flips (a comparison such as d < 0.1 taken differently from the fp64 oracle because the oracle's own margin is below the fp32
    tolerance): the count a parity test tolerated, printed and appended to gpurun_out/threshold_flips.txt (DESIGN section 2 quotes that file)."""
    line = f"{where}: {int(flips)} of {int(out_of)}"
    print("threshold flips --", line)
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "threshold_flips.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
