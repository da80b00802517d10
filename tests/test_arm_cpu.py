"""Physics checks of the hexacopter + 3-joint arm model (BASELINE config 3) on the CPU oracle.  No reference dynamics
exist for this configuration (the reference simulates it in Gazebo), so instead of golden vectors the model is pinned
by formulation-independent invariants:
  - total linear and angular momentum (computed from positions/velocities only) are conserved without external wrench
    while the joints move;
  - arm locked  => identical to the rigid-body oracle run on the composite mass / CoM / inertia;
  - link masses -> 0 => identical to the rigid hexacopter;
  - hover thrust balances the total weight; the gravity moment of the offset arm CoM is what tilts it.
"""
import ctypes as C

import numpy as np

import rl_aerial_manipulator_amd as amd
from oracle import oracle as O


def arm_cfg(**over):
    pc = amd._lib.default_config("hexa_arm", 1)
    cfg = O.reference_quad_config(1, flags=0)
    C.memmove(C.byref(cfg.vehicle), C.byref(pc.vehicle), C.sizeof(O.Vehicle))
    cfg.task.ee_task = pc.task.ee_task               # the arm's default: the waypoint task measures from the tool point
    for k, v in over.items():
        setattr(cfg.vehicle, k, v)
    return cfg


def rot_q(q):  # rotation matrix of the normalised quaternion (body->world is its transpose in this model)
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rodrigues(a, th):
    a = np.asarray(a, float); K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def momenta(cfg, s):
    """Total linear momentum and angular momentum about the world origin, from positions and velocities only."""
    v = cfg.vehicle
    Rwb = rot_q(s[6:10]).T                               # body -> world
    om = s[10:13]
    m_links = np.array(v.link_mass[:3]); m0 = v.mass - m_links.sum()
    bodies = [(m0, np.zeros(3), np.zeros(3), np.array(v.inertia).reshape(3, 3), np.zeros(3))]  # (m, r, u, J_body, w_rel)
    R = np.eye(3); p = np.zeros(3); pd = np.zeros(3); w = np.zeros(3)
    for k in range(3):
        o = np.array(v.joint_origin[3 * k:3 * k + 3]); ax = np.array(v.joint_axis[3 * k:3 * k + 3])
        pd = pd + np.cross(w, R @ o); p = p + R @ o
        w = w + (R @ ax) * s[16 + k]
        R = R @ rodrigues(ax, s[13 + k])
        c = R @ np.array(v.link_com[3 * k:3 * k + 3])
        J = R @ np.array(v.link_inertia[9 * k:9 * k + 9]).reshape(3, 3) @ R.T
        bodies.append((m_links[k], p + c, pd + np.cross(w, c), J, w.copy()))
    P = np.zeros(3); Lw = np.zeros(3); mt = 0; xc = np.zeros(3)
    for m, r, u, J, wr in bodies:
        x = s[0:3] + Rwb @ r                                         # world position of the CoM
        vel = s[3:6] + Rwb @ (np.cross(om, r) + u)                   # world velocity of the CoM
        P += m * vel
        Lw += np.cross(x, m * vel) + Rwb @ (J @ (om + wr))
        mt += m; xc += m * x
    return P, Lw, xc / mt


def test_momentum_conservation_with_moving_joints():
    cfg = arm_cfg(g=0.0)
    rng = np.random.RandomState(0)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    s = np.concatenate([[0.3, -0.2, 1.5], [0.4, -0.1, 0.2], q, [0.5, -0.8, 0.3], [0.2, -0.4, 0.6], [0.0, 0.0, 0.0]])
    P0, L0, _ = momenta(cfg, s)
    a = np.array([0, 0, 0, 0, 0.9, -0.7, 0.8], np.float32)           # zero rotor wrench (t_min = 0 below), joints slewing
    cfg.vehicle.t_min[:] = (C.c_double * 8)(*([0.0] * 8))
    worst = 0.0; peak_rate = 0.0
    for t in range(400):
        s, w = O.arm_dynamics_step(cfg, s, a)
        peak_rate = max(peak_rate, np.abs(s[16:19]).max())
        assert w[0] == 0 and not w[1:4].any()
        P, L, _ = momenta(cfg, s)
        worst = max(worst, np.abs(P - P0).max() / np.abs(P0).max(), np.abs(L - L0).max() / np.abs(L0).max())
    assert peak_rate > 1.0 and np.abs(s[13:16] - [0.2, -0.4, 0.6]).max() > 0.5                 # the arm really moved
    assert worst < 5e-9, worst                                                                 # RK4(5 ms) integration error only


def test_momentum_rate_equals_external_wrench():
    """dP/dt = F_world + m g,  dL/dt = x_O x F_world + R M + sum x_k x m_k g  (finite differences of the same momenta)."""
    cfg = arm_cfg()
    rng = np.random.RandomState(1)
    for trial in range(5):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        s = np.concatenate([rng.uniform(-1, 1, 3), rng.normal(0, 0.5, 3), q, rng.normal(0, 1, 3), rng.uniform(-1, 1, 3), rng.normal(0, 1, 3)])
        F, M, cmd = 25.0 + rng.normal(), rng.normal(0, 0.5, 3), rng.uniform(-1, 1, 3)
        d = O.arm_rhs(cfg, s, F, M, cmd)
        eps = 1e-6
        Pp, Lp, _ = momenta(cfg, s + eps * d); Pm, Lm, _ = momenta(cfg, s - eps * d)
        _, _, xc = momenta(cfg, s)
        dP, dL = (Pp - Pm) / (2 * eps), (Lp - Lm) / (2 * eps)
        Rwb = rot_q(s[6:10]).T
        Fw = Rwb @ np.array([0, 0, F]); g = np.array([0, 0, -cfg.vehicle.g]); mt = cfg.vehicle.mass
        # the quaternion constraint term 2(1-|q|^2)q vanishes for unit q, so the finite difference is exact to O(eps^2)
        np.testing.assert_allclose(dP, Fw + mt * g, rtol=0, atol=1e-8)
        np.testing.assert_allclose(dL, np.cross(s[0:3], Fw) + Rwb @ M + np.cross(xc, mt * g), rtol=0, atol=1e-8)


def test_massless_arm_is_the_rigid_hexacopter():
    cfg = arm_cfg()
    hexa = O.reference_quad_config(1, flags=0)
    C.memmove(C.byref(hexa.vehicle), C.byref(amd._lib.default_config("hexa", 1).vehicle), C.sizeof(O.Vehicle))
    for k in range(3):
        cfg.vehicle.link_mass[k] = 1e-300
        for j in range(9):
            cfg.vehicle.link_inertia[9 * k + j] = 0.0
    cfg.vehicle.mass = hexa.vehicle.mass
    for j in range(9):
        cfg.vehicle.inertia[j] = hexa.vehicle.inertia[j]; cfg.vehicle.inv_inertia[j] = hexa.vehicle.inv_inertia[j]
    rng = np.random.RandomState(2)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    s13 = np.concatenate([[0, 0, 2.0], rng.normal(0, 1, 3), q, rng.normal(0, 1, 3)])
    sa = np.concatenate([s13, [0.3, -0.2, 0.1], [0, 0, 0]]); sr = s13.copy()
    for t in range(100):
        a = rng.uniform([0.5, -0.3, -0.3, -0.3], [1.5, 0.3, 0.3, 0.3]).astype(np.float32)
        sa, _ = O.arm_dynamics_step(cfg, sa, np.concatenate([a, [0.5, 0.5, -0.5]]).astype(np.float32))
        sr, _ = O.dynamics_step(hexa, sr, a)
        assert np.abs(sa[:13] - sr).max() < 1e-10


def test_locked_arm_equals_composite_rigid_body():
    cfg = arm_cfg(joint_kp=0.0, joint_kd=0.0)            # thdd = 0, joints stay where they are
    th = np.array([0.4, -0.7, 0.9])
    s0 = np.concatenate([[0, 0, 2.0], [0, 0, 0], [1, 0, 0, 0], [0, 0, 0], th, [0, 0, 0]])
    _, _, xc0 = momenta(cfg, s0)
    # composite inertia about the composite CoM in body axes, from the same kinematics helper
    v = cfg.vehicle
    m_links = np.array(v.link_mass[:3]); m0 = v.mass - m_links.sum()
    R = np.eye(3); p = np.zeros(3); parts = [(m0, np.zeros(3), np.array(v.inertia).reshape(3, 3))]
    for k in range(3):
        p = p + R @ np.array(v.joint_origin[3 * k:3 * k + 3]); R = R @ rodrigues(v.joint_axis[3 * k:3 * k + 3], th[k])
        parts.append((m_links[k], p + R @ np.array(v.link_com[3 * k:3 * k + 3]), R @ np.array(v.link_inertia[9 * k:9 * k + 9]).reshape(3, 3) @ R.T))
    rc = sum(m * r for m, r, _ in parts) / v.mass
    Ic = sum(J + m * (((r - rc) @ (r - rc)) * np.eye(3) - np.outer(r - rc, r - rc)) for m, r, J in parts)
    rigid = O.reference_quad_config(1, flags=0)
    C.memmove(C.byref(rigid.vehicle), C.byref(cfg.vehicle), C.sizeof(O.Vehicle))
    rigid.vehicle.n_joints = 0
    for j in range(9):
        rigid.vehicle.inertia[j] = Ic.reshape(-1)[j]; rigid.vehicle.inv_inertia[j] = np.linalg.inv(Ic).reshape(-1)[j]
    rng = np.random.RandomState(3)
    sa = s0.copy(); sr = np.concatenate([xc0, s0[3:13]])
    for t in range(150):
        a = rng.uniform([0.8, -0.2, -0.2, -0.2], [1.2, 0.2, 0.2, 0.2]).astype(np.float32)
        sa, w = O.arm_dynamics_step(cfg, sa, np.concatenate([a, [0, 0, 0]]).astype(np.float32))
        # the rigid body sees the same thrust, and the rotor moments re-expressed about the composite CoM: M_c = M - rc x (0,0,F)
        Rwb = rot_q(sr[6:10]).T
        Mc = w[1:4] - np.cross(rc, [0, 0, w[0]])
        sr = rigid_step(rigid, sr, w[0], Mc)
        _, _, xc = momenta(cfg, sa)
        assert np.abs(xc - sr[0:3]).max() < 2e-9 and np.abs(sa[6:13] - sr[6:13]).max() < 2e-9, t
    assert np.allclose(sa[13:16], th) and not sa[16:19].any()


def rigid_step(cfg, s, F, M):
    """RK4 step of the rigid oracle RHS with an explicit wrench (bypasses the mixer)."""
    import ctypes
    lib = O.lib()
    # state_dot is static in the oracle: integrate here with the same closed form (quadcopter.py:66-103)
    v = cfg.vehicle
    I = np.array(v.inertia).reshape(3, 3); J = np.array(v.inv_inertia).reshape(3, 3)

    def f(y):
        q = y[6:10]; n2 = q @ q; R = rot_q(q)
        p_, q_, r_ = y[10:13]
        acc = R.T @ np.array([0, 0, F]) / v.mass - np.array([0, 0, v.g])
        Om = np.array([[0, -p_, -q_, -r_], [p_, 0, -r_, q_], [q_, r_, 0, -p_], [r_, -q_, p_, 0]])
        qd = -0.5 * Om @ q + 2 * (1 - n2) * q
        wd = J @ (M - np.cross(y[10:13], I @ y[10:13]))
        return np.concatenate([y[3:6], acc, qd, wd])
    h = cfg.task.dt
    k1 = f(s); k2 = f(s + 0.5 * h * k1); k3 = f(s + 0.5 * h * k2); k4 = f(s + h * k3)
    s = s + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    s[6:10] /= np.linalg.norm(s[6:10])
    return s


def test_hover_balance_and_servo_tracking():
    cfg = arm_cfg()
    v = cfg.vehicle
    s = np.concatenate([[0, 0, 2.0], [0, 0, 0], [1, 0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 0, 0]])
    d = O.arm_rhs(cfg, s, v.mass * v.g, np.zeros(3), np.zeros(3))
    _, _, xc = momenta(cfg, s)
    a_com = d[3:6] + np.cross(d[10:13], xc - s[0:3])          # attitude = identity, w = 0: a_com = a_O + wd x r_c
    assert np.abs(a_com).max() < 1e-12                        # thrust = total weight: the CoM does not accelerate ...
    tau = np.cross(xc - s[0:3], [0, 0, -v.mass * v.g])        # ... but the arm's CoM offset gives a gravity moment about O
    dL = O.arm_rhs(cfg, s, v.mass * v.g, -tau, np.zeros(3))   # cancel it with rotor moments: no angular acceleration either
    assert np.abs(dL[10:13]).max() < 1e-10 and np.abs(d[10:13]).max() > 1e-3
    # servo: commanded joint positions are reached (critically damped, acceleration-limited), within limits
    a = np.array([1.0, 0, 0, 0, 0.5, -0.5, 1.0], np.float32)
    for t in range(1200):
        s, w = O.arm_dynamics_step(cfg, s, a)
    np.testing.assert_allclose(s[13:16], w[4:7], atol=2e-3)
    assert abs(w[4] - 1.57) < 1e-6 and abs(w[5] + 0.785) < 1e-6 and abs(w[6] - 1.57) < 1e-6
    assert np.abs(s[16:19]).max() < 1e-2


# ---- forward kinematics of the tool point (north_star "arm forward kinematics") -----------------------------
def fk_numpy(cfg, s):
    """Independent restatement: chain of Rodrigues rotations about the SDF joint axes, tool offset, body -> world."""
    v = cfg.vehicle
    R = np.eye(3); p = np.zeros(3)
    for k in range(3):
        p = p + R @ np.array(v.joint_origin[3 * k:3 * k + 3])
        R = R @ rodrigues(v.joint_axis[3 * k:3 * k + 3], s[13 + k])
    p = p + R @ np.array(v.tool_offset[:3])
    return s[0:3] + rot_q(s[6:10]).T @ p


def test_forward_kinematics_matches_independent_chain_and_home_invariant():
    cfg = arm_cfg()
    rng = np.random.RandomState(0)
    home = np.array(cfg.vehicle.joint_origin[:9]).reshape(3, 3).sum(0) + np.array(cfg.vehicle.tool_offset[:3])
    assert abs(home[2] + 0.103522586 + 0.106 + 0.125) < 1e-9 and np.linalg.norm(home) < 0.4     # the tool hangs ~0.33 m below the body origin
    for _ in range(200):
        s = np.zeros(19)
        s[0:3] = rng.normal(0, 2, 3); q = rng.normal(size=4); s[6:10] = q / np.linalg.norm(q)
        s[13:16] = rng.uniform([-3.14, -1.57, -1.57], [3.14, 1.57, 1.57])
        np.testing.assert_allclose(O.ee_position(cfg, s), fk_numpy(cfg, s), rtol=0, atol=1e-13)
        # arm locked at home: the tool point is the base position plus a constant offset rotated by q (VERDICT invariant)
        s[13:16] = 0
        np.testing.assert_allclose(O.ee_position(cfg, s), s[0:3] + rot_q(s[6:10]).T @ home, rtol=0, atol=1e-13)
    # joint 1 turns about body z through the joint origin: the tool stays on a circle around that axis at constant height
    s = np.zeros(19); s[6] = 1; s[14:16] = (0.7, -0.4)
    pts = []
    for th1 in np.linspace(-3, 3, 13):
        s[13] = th1; pts.append(O.ee_position(cfg, s) - np.array(cfg.vehicle.joint_origin[:3]))
    pts = np.array(pts)
    assert np.ptp(pts[:, 2]) < 1e-13 and np.ptp(np.hypot(pts[:, 0], pts[:, 1])) < 1e-13


def test_tool_point_task_measures_from_the_end_effector():
    """AMENV_EE_TASK_TOOL (the arm's default): reward distance, reach test and obs[13:16] use the tool point; _BASE the base position."""
    outs = {}
    for mode in (O.EE_TASK_TOOL, O.EE_TASK_BASE):
        cfg = arm_cfg(); cfg.num_envs = 2; cfg.task.ee_task = mode
        orc = O.OracleEnv(cfg)
        orc.reset()
        assert orc.obs_dim == 29
        ee = orc.ee_position()
        # env 0: waypoint 5 cm from the tool point (inside the 0.1 ball), ~0.3 m from the base; env 1: 5 cm from the base
        orc.fstate[O.F_WP0:O.F_WP0 + 3, 0] = ee[0] + [0.05, 0, 0]
        orc.fstate[O.F_WP0:O.F_WP0 + 3, 1] = orc.fstate[0:3, 1] + [0.05, 0, 0]
        a = np.tile(np.array([1, 0, 0, 0, 0, 0, 0], np.float32), (2, 1))
        o = orc.step(a)
        outs[mode] = o
        ee1 = orc.ee_position()
        ref = ee1 if mode == O.EE_TASK_TOOL else orc.fstate[0:3].T
        np.testing.assert_allclose(o["obs"][:, 13:16], (orc.fstate[O.F_WP0:O.F_WP0 + 3].T - ref) / 2, atol=1e-7)
        np.testing.assert_allclose(o["obs"][:, 26:29], (ee1 - orc.fstate[0:3].T) / 0.5, atol=1e-7)
    tool, base = outs[O.EE_TASK_TOOL]["info"], outs[O.EE_TASK_BASE]["info"]
    assert (tool[0] & O.INFO_SUCCESS) and not (tool[1] & O.INFO_SUCCESS)
    assert (base[1] & O.INFO_SUCCESS) and not (base[0] & O.INFO_SUCCESS)


def _staged_aggregates(v, th, thd, thdd):
    """The 30 joint-configuration sums of the stage-wave kernel (csrc/amenv_arm.hpp "Staged form"), component by component as the device
    code forms them: S, U, Aa, I_O (incl. the base inertia), G, H, Tn."""
    I0 = np.array(v.inertia).reshape(3, 3)
    S, U, Aa, H, Tn = (np.zeros(3) for _ in range(5))
    IO = I0.copy(); G = np.zeros((3, 3))
    R = np.eye(3); p = np.zeros(3); pd = np.zeros(3); pdd = np.zeros(3); w = np.zeros(3); al = np.zeros(3)
    for k in range(3):
        o = np.array(v.joint_origin[3 * k:3 * k + 3]); ax = np.array(v.joint_axis[3 * k:3 * k + 3])
        Ro = R @ o
        pd = pd + np.cross(w, Ro); pdd = pdd + np.cross(al, Ro) + np.cross(w, np.cross(w, Ro)); p = p + Ro
        z = R @ ax
        al = al + z * thdd[k] + np.cross(w, z) * thd[k]; w = w + z * thd[k]; R = R @ rodrigues(ax, th[k])
        Rc = R @ np.array(v.link_com[3 * k:3 * k + 3])
        r = p + Rc; u = pd + np.cross(w, Rc); a_ = pdd + np.cross(al, Rc) + np.cross(w, np.cross(w, Rc))
        J = R @ np.array(v.link_inertia[9 * k:9 * k + 9]).reshape(3, 3) @ R.T
        m = v.link_mass[k]
        xx, xy, xz, yy, yz, zz = J[0, 0], J[0, 1], J[0, 2], J[1, 1], J[1, 2], J[2, 2]
        wx, wy, wz = w
        E = np.array([[2 * (wz * xy - wy * xz), (wx * xz - wz * xx) + (wz * yy - wy * yz), (wy * xx - wx * xy) + (wz * yz - wy * zz)],
                      [0.0, 2 * (wx * yz - wz * xy), (wy * xy - wx * yy) + (wx * zz - wz * xz)],
                      [0.0, 0.0, 2 * (wy * xz - wx * yz)]])
        E = E + np.triu(E, 1).T                                                   # D + D^T, D = J [w]x
        S += m * r; U += m * u; Aa += m * a_
        IO += J + m * ((r @ r) * np.eye(3) - np.outer(r, r))
        G += 2 * m * ((u @ r) * np.eye(3) - np.outer(u, r)) - E
        H += J @ w
        Tn += m * np.cross(r, a_) + J @ al + np.cross(w, J @ w)
    return S, U, Aa, IO, G, H, Tn


def test_staged_form_of_the_arm_dynamics_is_the_same_right_hand_side():
    """The stage-wave kernel (step_kernel_armk) evaluates the base dynamics on joint-configuration aggregates:
        fb = w x (w x S) + 2 w x U + Aa,   nb = w x (I_O w) + G w + w x H + Tn
    instead of summing per link with the base's angular velocity inside.  The identity is exact: same derivatives as the oracle's RHS
    (which sums per link) to rounding, for random attitudes, rates, joint states, commands and wrenches."""
    cfg = arm_cfg()
    v = cfg.vehicle
    rng = np.random.RandomState(0)
    worst = 0.0
    for _ in range(200):
        s = np.zeros(19)
        s[:6] = rng.normal(size=6); s[6:10] = rng.normal(size=4); s[6:10] /= np.linalg.norm(s[6:10])
        s[10:13] = rng.normal(size=3) * 2; s[13:16] = rng.uniform(-1.5, 1.5, 3); s[16:19] = rng.normal(size=3) * 2
        F = rng.uniform(10, 50); M = rng.normal(size=3); cmd = rng.uniform(-1.5, 1.5, 3)
        d = O.arm_rhs(cfg, s, F, M, cmd)
        om = s[10:13]
        thdd = np.clip(v.joint_kp * (cmd - s[13:16]) - v.joint_kd * s[16:19], -v.joint_acc_max, v.joint_acc_max)
        S, U, Aa, IO, G, H, Tn = _staged_aggregates(v, s[13:16], s[16:19], thdd)
        Rq = rot_q(s[6:10])
        gb = -v.g * Rq[:, 2]
        fb = np.cross(om, np.cross(om, S)) + 2 * np.cross(om, U) + Aa
        nb = np.cross(om, IO @ om) + G @ om + np.cross(om, H) + Tn
        f = v.mass * gb - fb; f[2] += F
        n = M + np.cross(S, gb) - nb
        Ic = IO - ((S @ S) * np.eye(3) - np.outer(S, S)) / v.mass
        wd = np.linalg.solve(Ic, n - np.cross(S, f) / v.mass)
        vd = Rq.T @ ((f + np.cross(S, wd)) / v.mass)
        worst = max(worst, np.abs(vd - d[3:6]).max() / max(1.0, np.abs(d[3:6]).max()), np.abs(wd - d[10:13]).max() / max(1.0, np.abs(d[10:13]).max()),
                    np.abs(thdd - d[16:19]).max())
    assert worst < 1e-12, worst


def test_phantom_links_are_exact_zeros():
    """The n-link adapter of the C ABI (n_joints = 1, 2: amenv_capi.hip pad_arm_config) runs the 3-joint kernels with PHANTOM links behind the real
    ones -- zero mass / inertia / CoM / joint offset, x axes, joint limits 0.  On the oracle: the padded 3-joint vehicle and the native shorter chain
    give the same step (the phantom terms are exact zeros, their rotations the identity) and the same tool point."""
    rng = np.random.RandomState(4)
    lm = [0.082, 0.054, 0.220]
    for nj in (1, 2):
        mass = 3.2121 - sum(lm[nj:])
        short = arm_cfg(n_joints=nj, mass=mass)
        padded = arm_cfg(mass=mass)
        v = padded.vehicle
        for k in range(nj, 3):
            v.link_mass[k] = 0.0
            for j in range(3):
                v.joint_origin[3 * k + j] = 0.0; v.link_com[3 * k + j] = 0.0; v.joint_axis[3 * k + j] = 1.0 if j == 0 else 0.0
            for j in range(9):
                v.link_inertia[9 * k + j] = 0.0
            v.joint_limit[2 * k] = v.joint_limit[2 * k + 1] = 0.0
        for _ in range(50):
            s = np.zeros(19)
            s[:6] = rng.normal(size=6); q = rng.normal(size=4); s[6:10] = q / np.linalg.norm(q); s[10:13] = rng.normal(size=3) * 2
            s[13:13 + nj] = rng.uniform(-1.5, 1.5, nj); s[16:16 + nj] = rng.normal(size=nj) * 2
            a = rng.uniform(-1, 1, 7).astype(np.float32); a[0] = rng.uniform(0, 2)
            a_short = a[:4 + nj].copy()
            a[4 + nj:] = rng.uniform(-1, 1, 3 - nj)          # whatever the padded action columns hold: limits 0 -> command 0
            s1, _ = O.arm_dynamics_step(short, s, a_short)
            s2, _ = O.arm_dynamics_step(padded, s, a)
            assert np.abs(s1 - s2).max() < 1e-14 and np.all(s2[13 + nj:16] == 0) and np.all(s2[16 + nj:19] == 0)
            assert np.abs(O.ee_position(short, s1) - O.ee_position(padded, s2)).max() < 1e-15
