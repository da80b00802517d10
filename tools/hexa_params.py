#!/usr/bin/env python3
"""Derive the lumped rigid-body parameters of the repo's hexacopter from its SDF.

Reads (never copies) /root/reference/hexacopter_description/custom_hexa/model.sdf, composes every
link's inertial (mass, CoM pose, inertia tensor) into the model frame and prints total mass, centre of
gravity and the composite inertia about the CoG (parallel-axis theorem).  The printed numbers are the
constants hard-coded in csrc/amenv_capi.hip:vehicle_hexa() and in DESIGN.md.  Build-container only.
"""
import sys
import xml.etree.ElementTree as ET

import numpy as np


def rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def pose(el):
    if el is None or el.text is None:
        return np.zeros(3), np.eye(3)
    v = [float(x) for x in el.text.split()]
    return np.array(v[:3]), rpy(*v[3:6])


def main(path):
    root = ET.parse(path).getroot()
    model = root.find("model")
    tot_m, first, parts = 0.0, np.zeros(3), []
    for link in model.findall("link"):
        inert = link.find("inertial")
        if inert is None:
            continue
        pl, Rl = pose(link.find("pose"))
        pi, Ri = pose(inert.find("pose"))
        m = float(inert.find("mass").text)
        I = np.zeros((3, 3))
        it = inert.find("inertia")
        if it is not None:
            g = lambda k: float(it.find(k).text) if it.find(k) is not None else 0.0
            I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
        c = pl + Rl @ pi
        R = Rl @ Ri
        parts.append((link.get("name"), m, c, R @ I @ R.T))
        tot_m += m
        first += m * c
    cog = first / tot_m
    Ic = np.zeros((3, 3))
    for _, m, c, I in parts:
        d = c - cog
        Ic += I + m * ((d @ d) * np.eye(3) - np.outer(d, d))
    np.set_printoptions(precision=10, suppress=False, linewidth=160)
    print(f"links with inertial: {len(parts)}")
    print(f"total mass  = {tot_m:.6f} kg")
    print(f"CoG (model) = {cog}")
    print("composite inertia about CoG (model axes):")
    print(Ic)
    return tot_m, cog, Ic


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/hexacopter_description/custom_hexa/model.sdf")
