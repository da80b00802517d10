#!/usr/bin/env python3
"""Lumped parameters of the hexacopter + 3-joint arm (BASELINE config 3) from the repo's SDF files.

Reads (never copies) hexacopter_description/custom_hexa/model.sdf and
Manipulator/src/manipulator_description/sdf/manipulator.sdf under /root/reference and prints the constants hard-coded in
csrc/amenv_capi.hip:vehicle_hexa_arm():
  * base body = the 27 hexacopter links + the manipulator's `base_plate` lump (fixed to base_link at zero offset,
    custom_hexa_arm/model.sdf:1746-1759): mass, CoM, inertia about that CoM; the body frame origin O is that CoM;
  * joints 1..3: origin in the parent frame (joint 1: in the body frame, relative to O), axis;
  * links 1..3: mass, CoM in the link frame, inertia about the CoM; the two gripper fingers (joint_4/5, rigidly closed at 0)
    are folded into link 3.
Build-container only."""
import xml.etree.ElementTree as ET

import numpy as np

from hexa_params import main as hexa_main, pose

MAN = "/root/reference/Manipulator/src/manipulator_description/sdf/manipulator.sdf"


def inertial(link):
    i = link.find("inertial")
    c, R = pose(i.find("pose"))
    m = float(i.find("mass").text)
    it = i.find("inertia")
    g = lambda k: float(it.find(k).text)
    I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
    return m, c, R @ I @ R.T


def combine(parts):
    """[(m, c, I_about_c)] -> (M, C, I about C)"""
    M = sum(p[0] for p in parts)
    C = sum(p[0] * p[1] for p in parts) / M
    I = np.zeros((3, 3))
    for m, c, Ic in parts:
        d = c - C
        I += Ic + m * ((d @ d) * np.eye(3) - np.outer(d, d))
    return M, C, I


def main():
    np.set_printoptions(precision=10, linewidth=160)
    m_h, cog_h, I_h = hexa_main("/root/reference/hexacopter_description/custom_hexa/model.sdf")
    model = ET.parse(MAN).getroot().find("model")
    links = {l.get("name"): l for l in model.findall("link")}
    joints = {j.get("name"): j for j in model.findall("joint")}
    base_link_origin = np.array([0.0, 0.0, 0.0015])  # custom_hexa/model.sdf:10 base_link pose; base_plate is fixed to it at zero offset
    m_bp, c_bp, I_bp = inertial(links["base_plate"])
    M0, C0, I0 = combine([(m_h, cog_h, I_h), (m_bp, base_link_origin + c_bp, I_bp)])
    print("\n== base body (hexacopter + base_plate) ==")
    print(f"mass {M0:.6f}  CoM (model frame) {C0}\ninertia about CoM:\n{I0}")
    jo = {k: pose(joints[k].find("pose"))[0] for k in ("joint_1", "joint_2", "joint_3", "joint_4", "joint_5")}
    ax = {k: [float(x) for x in joints[k].find("axis").find("xyz").text.split()] for k in jo}
    lim = {k: (float(joints[k].find("axis").find("limit").find("lower").text), float(joints[k].find("axis").find("limit").find("upper").text)) for k in jo}
    o1 = base_link_origin + jo["joint_1"] - C0
    print("\n== joints ==")
    print(f"joint_1 origin in body frame (rel. O) {o1} axis {ax['joint_1']} limits {lim['joint_1']}")
    print(f"joint_2 origin in link-1 frame {jo['joint_2']} axis {ax['joint_2']} limits {lim['joint_2']}")
    print(f"joint_3 origin in link-2 frame {jo['joint_3']} axis {ax['joint_3']} limits {lim['joint_3']}")
    m1, c1, I1 = inertial(links["arm_motor_2"])
    m2, c2, I2 = inertial(links["h_arm"])
    m3, c3, I3 = inertial(links["arm_motor_3"])
    mL, cL, IL = inertial(links["gripper_L"]); mR, cR, IR = inertial(links["gripper_R"])
    M3, C3, I3c = combine([(m3, c3, I3), (mL, jo["joint_4"] + cL, IL), (mR, jo["joint_5"] + cR, IR)])
    print("\n== links (mass, CoM in link frame, inertia about CoM) ==")
    for n, (m, c, I) in (("link1 arm_motor_2", (m1, c1, I1)), ("link2 h_arm", (m2, c2, I2)), ("link3 arm_motor_3 + closed gripper", (M3, C3, I3c))):
        print(f"{n}: m={m:.6f} c={c}\n{I}")
    print(f"tool point in link-3 frame (midpoint of the joint_4 / joint_5 origins, :371,450): {0.5 * (jo['joint_4'] + jo['joint_5'])}")
    print(f"\ntotal mass {M0 + m1 + m2 + M3:.6f} kg (hexa {m_h:.4f} + arm {m_bp + m1 + m2 + M3:.4f})")
    # rotor positions relative to O (x, y matter; thrust is along body z)
    print(f"rotor xy offsets from O: dx={-C0[0]:.3e} dy={-C0[1]:.3e}")


if __name__ == "__main__":
    main()
