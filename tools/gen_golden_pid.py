"""Golden vectors for row f4 (PID + minimum-snap baseline) from the UNMODIFIED reference modules
`initial-implementation-v2/PID Controller/{pid_controller,trajGen3D}.py` and its Quadcopter model, run the way the
reference's own `runsim.py:26-49` runs them (helix waypoints (10, 5), v = 1.2, control dt = 0.01, start (0.5, 0, 0)).

    python tools/gen_golden_pid.py        # -> tests/golden/pid_helix.npz

Stored per control step: time, desired state (pos/vel/acc/yaw/yawdot), the 13-state the controller saw, its roll/pitch/yaw,
the controller outputs F, M (before the mixer) and the state after `Quadcopter.update`; plus waypoints and the three
minimum-snap coefficient vectors.  Data only -- no reference source is copied.
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
PID_DIR = os.path.join(REF, "initial-implementation-v2", "PID Controller")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "pid_helix.npz")

if __name__ == "__main__":
    sys.path.insert(0, PID_DIR)
    import pid_controller as pid            # noqa: E402
    import trajGen3D                        # noqa: E402
    from model.quadcopter import Quadcopter  # noqa: E402

    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    quad = Quadcopter((0.5, 0, 0), (0, 0, 0))
    wps = trajGen3D.get_helix_waypoints(10, 5)
    cx, cy, cz = trajGen3D.get_MST_coefficients(wps)
    dt, t = 0.01, 0.0
    rec = {k: [] for k in ("t", "des_pos", "des_vel", "des_acc", "des_yaw", "des_yawdot", "state", "rpy", "F", "M", "state_next")}
    for _ in range(steps):
        with contextlib.redirect_stdout(io.StringIO()):     # generate_trajectory prints every call
            des = trajGen3D.generate_trajectory(t, 1.2, wps, cx, cy, cz)
        rec["t"].append(t)
        rec["des_pos"].append(np.array(des.pos, float)); rec["des_vel"].append(np.array(des.vel, float)); rec["des_acc"].append(np.array(des.acc, float))
        rec["des_yaw"].append(float(des.yaw)); rec["des_yawdot"].append(float(des.yawdot))
        rec["state"].append(quad.state.copy()); rec["rpy"].append(np.array(quad.attitude(), float))
        F, M = pid.run(quad, des, dt)
        rec["F"].append(float(F)); rec["M"].append(np.array(M, float).reshape(3))
        quad.update(dt, F, M)
        rec["state_next"].append(quad.state.copy())
        t += dt
    np.savez_compressed(OUT, waypoints=wps, coeff_x=cx, coeff_y=cy, coeff_z=cz, speed=1.2, dt=dt,
                        **{k: np.array(v) for k, v in rec.items()}, numpy_version=np.array(np.__version__))
    s = np.array(rec["state_next"])
    print("final pos", s[-1, :3], "last waypoint", wps[-1], "max |pos - des|", np.abs(s[:, :3] - np.array(rec["des_pos"])).max(), file=sys.stderr)
