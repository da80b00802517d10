// amenv_team_math.hpp -- the arithmetic of the lane-team kernels (hexacopter + z,x,x arm), written ONCE for three lane value types X:
//
//   float   the product: one env = one DPP row of 16 lanes, cross-lane operands through DPP modifiers (amenv_team.hpp)
//   double  the fp64 logic-gate build of the SAME kernel on the GPU (DPP on register pairs): every selector, row sum and stage hand-over of the
//           product's plumbing is then held to <= 1e-12 against the fp64 oracle (tests/test_gpu_arm.py)
//   HV      a host emulation (16 explicit lanes, fp64): tests/emu/team_emu.cpp compiles this header with g++ so that the formulation is
//           checked against the oracle in the CPU suite, without a GPU (tests/test_team_emu_cpu.py).  Test infrastructure only.
//
// The includer defines, for its X, before this header: qp<P0,P1,P2,P3>(X) (quad permute), row_ror<N>(X), row_shr<N>(X) (lane i reads lane
// i - N of its 16-lane row, 0 where that lane does not exist), fma_, rcp_, rsqrt_, clamp_, sincos_t, sel(mask, a, b), scale_action_f32,
// joint_cmd_f32, and AMENV_FN (function qualifiers).
//
// Layout of one env row: 4 quads x 4 lanes.  3-vectors live with ONE component per lane (lanes 0..2 of a quad), 3x3 matrices as three
// column registers with one ROW per lane, quaternions in the 4 lanes of a quad.  The base state (P, V, Q, W, joints) is replicated in the
// four quads; inside the RK4 quad s works on STAGE s.
//
// RK4 of the base on joint-configuration aggregates (round 3 form).  The joint servos do not feel the base, so the joint state of all
// four stages is known up front; quad s forms stage s's joint configuration and reduces it to what the base dynamics needs:
//     S = sum m r, U = sum m u, Aa = sum m a                       (first moments of position / relative velocity / relative acceleration)
//     I_c = I_O - (|S|^2 1 - S S^T) / mtot, C = I_c^-1              (composite inertia about the system CoM)
//     L  = -sum_k [ 2 m ((u.r) 1 - u r^T) + [w_k]x (2 J - tr(J) 1) ] + (2 / mtot) ((S.U) 1 - U S^T)
//     c0 = M - Tn + S x (Aa - F e_z) / mtot,  Tn = sum m r x a + J al + w_k x (J w_k)
// With them the base's angular acceleration is Euler's equation about the system CoM,
//     wd = C (c0 + L w - w x (I_c w)),
// which is the oracle's  I_c wd = n - S x f / mtot  with the terms sorted by their power of w: gravity drops out (it exerts no moment
// about the CoM: S x g - S x (mtot g) / mtot), the quadratic terms collapse to w x (I_c w), and the link terms J (w x w_k) + w x (J w_k) +
// w_k x (J w) are [w_k]x (2 J - tr(J) 1) w for a symmetric J (tr(J) = tr(I_link), a constant).  So the w chain of the four stages does not
// depend on the attitude at all; it runs SYSTOLICALLY: every round all four quads evaluate wd and dq with their own aggregates on their
// own copy of the stage state, and quad s + 1 takes  y + c_s * (quad s's derivative)  through one row_shr:4 -- after round s quad s + 1
// holds the true stage s + 1 state and keeps recomputing the same numbers from then on (its upstream is steady), no selects, no row sums.
// The translational acceleration is off the chain: after round 4 quad s holds stage s's (Q, w, wd) and computes ITS stage's vd, all four at
// once; the RK4 combination is one weighted row sum per state register (sum_bodies: bit-identical in the four quads).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace amenv_dev {

template <typename X> struct LaneTraits { using T = X; using M = bool; };   // scalar type, lane-mask type (the host emulation specialises)

// Per-lane constants: a table built by amenv_create (amenv_team_host.hpp), constant k of lane column 4 * quad + component.
enum TeamConst {
  // shared with the rigid lane-quad kernel (amenv_quad.hpp): three 16-byte pieces it loads anyway + one of its own
  TC_E0 = 0, TC_E1, TC_E2, TC_GV,                      // 1 if component == j; gravity (0, 0, -g)
  TC_SP, TC_SQ, TC_SR, TC_O0,                          // signs of the quaternion kinematics (incl. the 1/2); joint-1 origin (component per lane)
  TC_ACT1, TC_ACT2, TC_JHALF, TC_JMID,                 // action scaling u = (a * ACT1) * ACT2: (mass, ms, ms, ms), (g, 1, 1, 1) -- fp32, left to right; joint command = fma(action, half, mid)
  TC_ALLOC0, TC_MIX0 = TC_ALLOC0 + 6,                  // lane-quad kernel only: alloc[r][component], mix[component][r], r = 0..5 (three pieces the team kernel never loads)
  // lane-team kernels
  TC_GV1 = TC_MIX0 + 6, TC_GV2,                        // the two rotations of the gravity vector
  TC_E01, TC_ZX,                                       // (1, 1, 0, 0); (-1, 1, 0, 0): e_z x v = ZX * quad_perm[1,0,2,3](v)
  TC_I0C0, TC_I0C1, TC_I0C2,                           // base body inertia about its CoM: column j, row = component
  TC_I1C0, TC_I1C1, TC_I1C2,                           // link 1 inertia (link frame), column j
  TC_LC0, TC_O1,                                       // link 1 CoM in its frame; joint-2 origin in link 1's frame (component per lane)
  TC_NTR0, TC_NTR_END = TC_NTR0 + 9,                   // -tr(I_link k) e_j at TC_NTR0 + 3 k + j
  TC_ALC0 = TC_NTR_END, TC_ALC1, TC_ALC2, TC_ALC3,     // mixer, lane = rotor (lanes 0..5 of the row): alloc[rotor][0..3]
  TC_TMIN, TC_TMAX,                                    // ... its thrust limits
  TC_MXQ0, TC_MXQ1, TC_MXQ2, TC_MXQ3,                  // re-mix: mix[component][4 quad + j] (0 where there is no such rotor)
  TC_OBS_A, TC_OBS_B, TC_OBS_C,                        // observation scalings of the three row segments this lane writes
  TC_HSTEP,                                            // RK4 hand-over: stage state of this quad = y + HSTEP * (previous quad's derivative): 0, h/2, h/2, h
  TC_WGT,                                              // RK4 weights of this quad's stage: h/6 * (1, 2, 2, 1)
  TC_PW,                                               // position: h * h/6 * (1, 1, 1, 0)
  kTeamConsts
};
constexpr int kTeamFirstPiece = TC_GV1 / 4;            // the team kernels load pieces 0..2 and from here on
static_assert(TC_ALLOC0 % 4 == 0 && TC_GV1 % 4 == 0, "the lane-quad kernel's mixer constants fill whole 16-byte pieces");

template <typename T>
struct TeamParamsT {          // wave-uniform (SGPRs)
  T o1[3], o2[3];            // joint-2 / joint-3 origins in their parent frames
  T tool[3];
  T kp, kd, amax, mtot, inv_mtot, g, h;
  T ee_home[3];
  T lm[3], lcm[3][3], li[3][6];   // link masses, CoMs (link frame), inertias (xx xy xz yy yz zz)
  int32_t substeps, max_steps, counter_limit, ee_task, K;   // K = 1 (the task code reads it)
  uint32_t flags;
  const void* consts;        // per-lane constant table (layout: amenv_team_host.hpp)
};

// Device block of the step kernel: the per-lane table, then the wave-uniform parameters.  The step kernel reads the parameters from THERE
// with scalar loads (cached in L2 like any data) instead of from its kernel arguments: a trip to the kernel-argument segment is served from
// memory on every launch and costs ~600 clocks per batch, which would sit in front of the RK4 for every value it needs.
constexpr int kTeamTablePieces = (kTeamConsts + 3) / 4;
template <typename T> constexpr size_t team_table_bytes() { return size_t(kTeamTablePieces) * 4 * 16 * sizeof(T); }
template <typename T> constexpr size_t team_block_bytes() { return team_table_bytes<T>() + sizeof(TeamParamsT<T>); }

template <typename X> struct TeamStateT { X P, V, Q, W, TH, THD; };   // one register each: position, velocity (component per lane), quaternion (4 lanes), body rates, joints

// ---- cross-lane helpers on top of the includer's qp / row_ror / row_shr ----------------------------------------------------------
template <int J, typename X> AMENV_FN X bc(X v) { return qp<J, J, J, J>(v); }          // component J to the whole quad
template <typename X> AMENV_FN X rot1(X v) { return qp<1, 2, 0, 3>(v); }               // v[(c+1)%3]
template <typename X> AMENV_FN X rot2(X v) { return qp<2, 0, 1, 3>(v); }               // v[(c+2)%3]
// Sum over the 4 quads of the row, BIT-IDENTICAL in all of them: opposite quads first (x0 + x2 and x2 + x0 are the same number), then
// the two pair sums (a + b = b + a).  With neighbours first each quad would associate the four terms differently, the replicated base
// state would drift apart between the quads by rounding, and a threshold of the task step could then be taken differently inside one row.
template <typename X> AMENV_FN X sum_bodies(X v) { v = v + row_ror<8>(v); return v + row_ror<4>(v); }
template <typename X> AMENV_FN X sum4(X p) { const X t = p + qp<1, 0, 3, 2>(p); return t + qp<2, 3, 0, 1>(t); }   // all 4 lanes valid in, all out
template <typename X> AMENV_FN X dot3(X a, X b) { const X p = a * b; return (p + rot1(p)) + rot2(p); }            // lanes 0..2

template <typename X> struct X3 { X v, r1, r2; };   // a 3-vector (component per lane) with its two rotations cached
template <typename X> AMENV_FN X3<X> x3(X v) { return X3<X>{v, rot1(v), rot2(v)}; }
template <typename X> AMENV_FN X cross(const X3<X>& a, const X3<X>& b) { return fma_(a.r1, b.r2, -(a.r2 * b.r1)); }
// cross product a x b when only `a` has its rotations cached: the other operand's rotations ride as DPP modifiers of the two
// multiplies (v_mul_f32_dpp; an FMA cannot carry one), so no v_mov_b32_dpp is spent on `b`
template <typename X> AMENV_FN X cross_c(const X3<X>& a, X b) { return a.r1 * rot2(b) - a.r2 * rot1(b); }
template <typename X> struct TM { X c0, c1, c2; };  // 3x3 matrix: lane i holds row i, one register per column
template <typename X> AMENV_FN X matvec(const TM<X>& M, X v) { return fma_(M.c0, bc<0>(v), fma_(M.c1, bc<1>(v), M.c2 * bc<2>(v))); }

// R <- R Rot(column AX, angle): the two other columns mix, lane-wise (row per lane)
template <int AX, typename X>
AMENV_FN void rotate_cols(TM<X>& R, X s, X co) {
  X& a = AX == 0 ? R.c1 : (AX == 1 ? R.c2 : R.c0);
  X& b = AX == 0 ? R.c2 : (AX == 1 ? R.c0 : R.c1);
  const X ra = a, rb = b;
  a = fma_(co, ra, s * rb);
  b = fma_(co, rb, -(s * ra));
}

template <typename X, typename PT> AMENV_FN X joint_accel(const PT& P, X cmd, X th, X thd) {   // servo, joint k in lane k
  return clamp_(fma_(P.kp, cmd - th, -(P.kd * thd)), -P.amax, P.amax);
}

// What a joint configuration (one RK4 stage) hands to the base dynamics: 19 registers.
template <typename X> struct TeamStage {
  X3<X> S, U;        // sum m r, sum m u (component per lane, rotations cached)
  X Aa;              // sum m a
  TM<X> Ic, C, L;    // composite inertia about the system CoM, its inverse, the operator of the terms linear in w
  X c0;              // the terms free of w
};

// Joint configuration (TH, THD, thdd: joint k in lane k) -> aggregates.  Fe2 = thrust along body z (component per lane), Mv = rotor moments.
template <typename X, typename PT>
AMENV_FN TeamStage<X> team_kin_stage(const PT& P, const X* c, X TH, X THD, X thdd, X Fe2, X Mv) {
  using T = typename LaneTraits<X>::T;
  const X e0 = c[TC_E0], e1 = c[TC_E1], e2 = c[TC_E2];
  X S, U, Aa, Tn;
  TM<X> IO{c[TC_I0C0], c[TC_I0C1], c[TC_I0C2]};          // base body: r = 0, J = I0
  TM<X> G{T(0), T(0), T(0)};
  X sIO = T(0), sG = T(0);                               // the isotropic parts m |r|^2 and 2 m (u . r): added to the diagonals once, after the links
  TM<X> R;
  X p, pd, pdd, w, al;
  {   // joint 1 about z at the start of the chain and link 1 behind it: R = Rz(th), w = thd e_z, al = thdd e_z, the frame origin at rest
    X s, co;
    sincos_t(bc<0>(TH), s, co);
    const X td = bc<0>(THD), tdd = bc<0>(thdd);
    const X cov = fma_(co, c[TC_E01], e2), sv = s * c[TC_ZX];
    auto zrot = [&](X v) { return fma_(cov, v, sv * qp<1, 0, 2, 3>(v)); };   // Rz v
    auto zx = [&](X v) { return c[TC_ZX] * qp<1, 0, 2, 3>(v); };             // e_z x v
    R.c0 = fma_(co, e0, s * e1); R.c1 = fma_(co, e1, -(s * e0)); R.c2 = e2;
    p = c[TC_O0];
    {
      const T m = P.lm[0];
      const X Rc = zrot(c[TC_LC0]), zRc = zx(Rc);
      const X r = p + Rc, u = td * zRc;                   // u = w x Rc
      const X a_ = fma_(tdd, zRc, td * zx(u));            // al x Rc + w x (w x Rc)
      const X A0 = zrot(c[TC_I1C0]), A1 = zrot(c[TC_I1C1]), A2 = zrot(c[TC_I1C2]);   // Rz I, columns
      const TM<X> J{fma_(co, A0, -(s * A1)), fma_(s, A0, co * A1), A2};              // (Rz I) Rz^T
      S = m * r; U = m * u; Aa = m * a_;
      const X mr = m * r;
      sIO = m * dot3(r, r);
      const X r0 = bc<0>(r), r1 = bc<1>(r), r2 = bc<2>(r);
      IO.c0 = IO.c0 + fma_(-mr, r0, J.c0); IO.c1 = IO.c1 + fma_(-mr, r1, J.c1); IO.c2 = IO.c2 + fma_(-mr, r2, J.c2);
      const T m2 = m + m;
      const X mu = m2 * u;
      sG = m2 * dot3(u, r);
      G.c0 = fma_(td, zx(fma_(T(2), J.c0, c[TC_NTR0 + 0])), -(mu * r0));      // [w]x (2 J - tr(J) 1) = thd [e_z]x (...)
      G.c1 = fma_(td, zx(fma_(T(2), J.c1, c[TC_NTR0 + 1])), -(mu * r1));
      G.c2 = fma_(td, zx(fma_(T(2), J.c2, c[TC_NTR0 + 2])), -(mu * r2));
      Tn = fma_(m, cross_c(x3(r), a_), fma_(tdd, J.c2, (td * td) * zx(J.c2)));   // m r x a + J al + w x (J w)
    }
    {   // across joint 2 (about link 1's x axis = column 0 of R) with the parent at w = thd e_z
      X s2, co2;
      sincos_t(bc<1>(TH), s2, co2);
      const X td1 = bc<1>(THD), tdd1 = bc<1>(thdd);
      const X Ro = zrot(c[TC_O1]), zRo = zx(Ro);
      pd = td * zRo;
      pdd = fma_(tdd, zRo, td * zx(pd));
      p = p + Ro;
      const X z = R.c0;
      al = fma_(td1, td * zx(z), fma_(tdd1, z, tdd * e2));    // al + z thdd + (w x z) thd
      w = fma_(td1, z, td * e2);
      rotate_cols<0>(R, s2, co2);
    }
  }
  auto link = [&](int k) {   // link k behind the joints advanced so far: CoM motion relative to the body frame, inertia in body axes, sums
    const T m = P.lm[k];
    const X Rc = fma_(R.c0, P.lcm[k][0], fma_(R.c1, P.lcm[k][1], R.c2 * P.lcm[k][2]));
    const X3<X> xw = x3(w), xRc = x3(Rc);
    const X wRc = cross(xw, xRc);
    const X r = p + Rc, u = pd + wRc;
    const X a_ = pdd + cross_c(xw, wRc) - cross_c(xRc, al);
    const X RI0 = fma_(R.c0, P.li[k][0], fma_(R.c1, P.li[k][1], R.c2 * P.li[k][2]));
    const X RI1 = fma_(R.c0, P.li[k][1], fma_(R.c1, P.li[k][3], R.c2 * P.li[k][4]));
    const X RI2 = fma_(R.c0, P.li[k][2], fma_(R.c1, P.li[k][4], R.c2 * P.li[k][5]));
    const TM<X> J{fma_(RI0, bc<0>(R.c0), fma_(RI1, bc<0>(R.c1), RI2 * bc<0>(R.c2))), fma_(RI0, bc<1>(R.c0), fma_(RI1, bc<1>(R.c1), RI2 * bc<1>(R.c2))),
                  fma_(RI0, bc<2>(R.c0), fma_(RI1, bc<2>(R.c1), RI2 * bc<2>(R.c2)))};
    S = fma_(m, r, S); U = fma_(m, u, U); Aa = fma_(m, a_, Aa);
    const X mr = m * r;
    sIO = fma_(m, dot3(r, r), sIO);
    const X r0 = bc<0>(r), r1 = bc<1>(r), r2 = bc<2>(r);
    IO.c0 = IO.c0 + fma_(-mr, r0, J.c0); IO.c1 = IO.c1 + fma_(-mr, r1, J.c1); IO.c2 = IO.c2 + fma_(-mr, r2, J.c2);
    // G += 2 m ((u . r) 1 - u r^T) + [w]x (2 J - tr(J) 1): the point mass's 2 m r x (W x u) and the link's J (W x w) + W x (J w) + w x (J W)
    const T m2 = m + m;
    const X mu = m2 * u;
    sG = fma_(m2, dot3(u, r), sG);
    G.c0 = G.c0 + (cross_c(xw, fma_(T(2), J.c0, c[TC_NTR0 + 3 * k + 0])) - mu * r0);
    G.c1 = G.c1 + (cross_c(xw, fma_(T(2), J.c1, c[TC_NTR0 + 3 * k + 1])) - mu * r1);
    G.c2 = G.c2 + (cross_c(xw, fma_(T(2), J.c2, c[TC_NTR0 + 3 * k + 2])) - mu * r2);
    const X Jw = matvec(J, w);
    Tn = Tn + (fma_(m, cross_c(x3(r), a_), matvec(J, al)) + cross_c(xw, Jw));
  };
  auto advance_x = [&](X th, X td, X tdd, const T* o) {   // across a joint about its frame's x axis (column 0 of R)
    X s, co;
    sincos_t(th, s, co);
    const X Ro = fma_(R.c0, o[0], fma_(R.c1, o[1], R.c2 * o[2]));
    const X3<X> xRo = x3(Ro), xw = x3(w);
    const X wRo = cross(xw, xRo);
    pd = pd + wRo;
    pdd = pdd + (cross_c(xw, wRo) - cross_c(xRo, al));      // al x Ro + w x (w x Ro)
    p = p + Ro;
    const X z = R.c0;
    const X wz = cross_c(xw, z);
    al = fma_(td, wz, fma_(tdd, z, al));
    w = fma_(td, z, w);
    rotate_cols<0>(R, s, co);
  };
  link(1);
  advance_x(bc<2>(TH), bc<2>(THD), bc<2>(thdd), P.o2);
  link(2);
  TeamStage<X> k;
  // composite inertia about the system CoM, I_c = I_O - (|S|^2 1 - S S^T) / mtot, its inverse = adjugate (columns = cross products of columns) / det
  const T im = P.inv_mtot;
  const X imS2 = im * dot3(S, S), imS = im * S;
  const X S0 = bc<0>(S), S1 = bc<1>(S), S2 = bc<2>(S);
  const X dI = sIO - imS2;
  k.Ic = TM<X>{fma_(imS, S0, fma_(dI, e0, IO.c0)), fma_(imS, S1, fma_(dI, e1, IO.c1)), fma_(imS, S2, fma_(dI, e2, IO.c2))};
  const X3<X> x0 = x3(k.Ic.c0), x1 = x3(k.Ic.c1), x2 = x3(k.Ic.c2);
  const TM<X> A{cross(x1, x2), cross(x2, x0), cross(x0, x1)};
  const X idet = rcp_(dot3(x0.v, A.c0));
  k.C = TM<X>{idet * A.c0, idet * A.c1, idet * A.c2};
  // L = -G + (2 / mtot) ((S . U) 1 - U S^T)
  const T im2 = im + im;
  const X imU2 = im2 * U, dL = fma_(im2, dot3(S, U), -sG);
  k.L = TM<X>{fma_(-imU2, S0, fma_(dL, e0, -G.c0)), fma_(-imU2, S1, fma_(dL, e1, -G.c1)), fma_(-imU2, S2, fma_(dL, e2, -G.c2))};
  k.S = x3(S); k.U = x3(U); k.Aa = Aa;
  k.c0 = fma_(im, cross_c(k.S, Aa - Fe2), Mv - Tn);
  return k;
}

// base angular acceleration of a stage: wd = C (c0 + L w - w x (I_c w))
template <typename X> AMENV_FN X team_wdot(const TeamStage<X>& k, X W) {
  const X b0 = bc<0>(W), b1 = bc<1>(W), b2 = bc<2>(W);
  const X IcW = fma_(k.Ic.c0, b0, fma_(k.Ic.c1, b1, k.Ic.c2 * b2));
  const X LW = fma_(k.L.c0, b0, fma_(k.L.c1, b1, fma_(k.L.c2, b2, k.c0)));
  return matvec(k.C, LW - cross_c(x3(W), IcW));
}
// quaternion kinematics (4 lanes): -1/2 Omega(w) q + 2 (1 - |q|^2) q
template <typename X> AMENV_FN X team_qdot(const X* c, X Q, X W) {
  using T = typename LaneTraits<X>::T;
  const X n2 = sum4(Q * Q);
  X dq = fma_(T(-2), n2, T(2)) * Q;
  dq = fma_(c[TC_SP] * bc<0>(W), qp<1, 0, 3, 2>(Q), dq);
  dq = fma_(c[TC_SQ] * bc<1>(W), qp<2, 3, 0, 1>(Q), dq);
  dq = fma_(c[TC_SR] * bc<2>(W), qp<3, 2, 1, 0>(Q), dq);
  return dq;
}
// world acceleration of the body origin for a stage whose angular acceleration wd is known:
//   fb = w x (w x S) + 2 w x U + Aa, f = F e_z + mtot g_body - fb, A = (f + S x wd) / mtot, vd = Rq^T A
template <typename X, typename PT> AMENV_FN X team_vdot(const PT& P, const X* c, const TeamStage<X>& k, X Q, X W, X wd, X Fe2) {
  using T = typename LaneTraits<X>::T;
  const X n2 = sum4(Q * Q);
  const X two_in2 = T(2) * rcp_(n2);
  const X3<X> qv{qp<1, 2, 3, 3>(Q), qp<2, 3, 1, 3>(Q), qp<3, 1, 2, 3>(Q)};     // vector part in component layout, with its rotations
  const X qw = bc<0>(Q);
  const X3<X> om = x3(W);
  const X3<X> gv{c[TC_GV], c[TC_GV1], c[TC_GV2]};
  // gravity in body components: Rq (0,0,-g) = v + (2/|q|^2) qv x (qv x v + qw v)
  const X gb = fma_(two_in2, cross_c(qv, fma_(qw, gv.v, cross(qv, gv))), gv.v);
  const X fb = fma_(T(2), cross(om, k.U), cross_c(om, cross(om, k.S))) + k.Aa;
  const X f = Fe2 + fma_(P.mtot, gb, -fb);
  const X Aacc = P.inv_mtot * (f + cross_c(k.S, wd));
  // Rq^T A = A + (2/|q|^2) qv x (qv x A - qw A)
  return fma_(two_in2, cross_c(qv, fma_(-qw, Aacc, cross_c(qv, Aacc))), Aacc);
}

// tool point relative to the body origin, world axes, of a (unit-quaternion) state
template <typename X, typename PT> AMENV_FN X team_tool_offset(const PT& P, const X* c, const TeamStateT<X>& y) {
  using T = typename LaneTraits<X>::T;
  const X e0 = c[TC_E0], e1 = c[TC_E1], e2 = c[TC_E2];
  X s, co;
  sincos_t(bc<0>(y.TH), s, co);
  TM<X> R{fma_(co, e0, s * e1), fma_(co, e1, -(s * e0)), e2};
  X p = c[TC_O0];
  p = p + fma_(R.c0, P.o1[0], fma_(R.c1, P.o1[1], R.c2 * P.o1[2]));
  sincos_t(bc<1>(y.TH), s, co);
  rotate_cols<0>(R, s, co);
  p = p + fma_(R.c0, P.o2[0], fma_(R.c1, P.o2[1], R.c2 * P.o2[2]));
  sincos_t(bc<2>(y.TH), s, co);
  rotate_cols<0>(R, s, co);
  p = p + fma_(R.c0, P.tool[0], fma_(R.c1, P.tool[1], R.c2 * P.tool[2]));
  // world = Rq^T body (|q| = 1)
  const X3<X> qv{qp<1, 2, 3, 3>(y.Q), qp<2, 3, 1, 3>(y.Q), qp<3, 1, 2, 3>(y.Q)};
  const X qw = bc<0>(y.Q);
  return fma_(T(2), cross_c(qv, fma_(-qw, p, cross_c(qv, p))), p);
}

// What the four rounds leave in quad s: stage s's derivatives (dV, dQ, dW), for tests of the plumbing (amenv_team_rhs).
template <typename X> struct TeamStageDeriv { X dV, dQ, dW, Qs, Ws; };

// One control step of the dynamics of one env row, state in registers: mixer -> per-rotor clamp -> re-mix (quadcopter.py:109-112), joint
// commands, RK4 (sub-steps), renormalisation.  act: wrench action a0..a3 (one per lane), actj: joint actions (joint per lane).
// q0..q2: "this lane sits in quad 0 / 1 / 2" (quad s takes RK4 stage s + 1).  Returns the tool offset of the new state.
template <typename X, typename PT, typename M>
AMENV_FN X team_dynamics(const PT& P, const X* c, M q0, M q1, M q2, TeamStateT<X>& y, X act, X actj, TeamStageDeriv<X>* probe = nullptr) {
  using T = typename LaneTraits<X>::T;
  // mixer: lane 4 q + j of the row takes rotor 4 q + j (rotors 0..5 in quads 0 and 1): thrust = alloc[rotor] . u, clamp, then wrench
  // entry c = sum over rotors of mix[c][rotor] * thrust: per quad over its own four lanes, then over the quads of the row
  const X uu = scale_action_f32(act, c[TC_ACT1], c[TC_ACT2]);
  X t = fma_(c[TC_ALC0], bc<0>(uu), fma_(c[TC_ALC1], bc<1>(uu), fma_(c[TC_ALC2], bc<2>(uu), c[TC_ALC3] * bc<3>(uu))));
  t = clamp_(t, c[TC_TMIN], c[TC_TMAX]);
  const X wr = sum_bodies(fma_(c[TC_MXQ0], bc<0>(t), fma_(c[TC_MXQ1], bc<1>(t), fma_(c[TC_MXQ2], bc<2>(t), c[TC_MXQ3] * bc<3>(t)))));
  const X Fe2 = bc<0>(wr) * c[TC_E2], Mv = qp<1, 2, 3, 3>(wr);
  const X cmd = joint_cmd_f32(actj, c[TC_JHALF], c[TC_JMID]);
  const T h = P.h, hh = T(0.5) * h, h6 = h * T(1.0 / 6.0);
  const X hs = c[TC_HSTEP], wgt = c[TC_WGT], pw = c[TC_PW];
  int it = 0;
  do {
    // joint stage states (the servos do not feel the base): every lane, joint k in lane k
    const X a1 = joint_accel(P, cmd, y.TH, y.THD);
    const X TH2 = fma_(hh, y.THD, y.TH), THD2 = fma_(hh, a1, y.THD), a2 = joint_accel(P, cmd, TH2, THD2);
    const X TH3 = fma_(hh, THD2, y.TH), THD3 = fma_(hh, a2, y.THD), a3 = joint_accel(P, cmd, TH3, THD3);
    const X TH4 = fma_(h, THD3, y.TH), THD4 = fma_(h, a3, y.THD), a4 = joint_accel(P, cmd, TH4, THD4);
    // quad s of the row takes RK4 stage s: its joint state, then the aggregates of that joint configuration (all four stages at once)
    const X THs = sel(q0, y.TH, sel(q1, TH2, sel(q2, TH3, TH4))), THDs = sel(q0, y.THD, sel(q1, THD2, sel(q2, THD3, THD4)));
    const X as = sel(q0, a1, sel(q1, a2, sel(q2, a3, a4)));
    const TeamStage<X> k = team_kin_stage(P, c, THs, THDs, as, Fe2, Mv);
    // four systolic rounds: quad s + 1 takes y + hs * (quad s's derivative); quad 0 reads nothing (hs = 0, the shifted-in value is 0)
    X Wc = y.W, Qc = y.Q, wd, dq;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      wd = team_wdot(k, Wc);
      dq = team_qdot(c, Qc, Wc);
      if (s < 3) { Wc = fma_(hs, row_shr<4>(wd), y.W); Qc = fma_(hs, row_shr<4>(dq), y.Q); }
    }
    const X vd = team_vdot(P, c, k, Qc, Wc, wd, Fe2);
    if (probe) { probe->dV = vd; probe->dQ = dq; probe->dW = wd; probe->Qs = Qc; probe->Ws = Wc; }
    // RK4 combination: weighted row sums (position: kP_s = V + hs vd_{s-1}  ->  P + h V + h h/6 (vd1 + vd2 + vd3))
    y.P = fma_(h, y.V, y.P) + sum_bodies(pw * vd);
    y.V = y.V + sum_bodies(wgt * vd);
    y.Q = y.Q + sum_bodies(wgt * dq);
    y.W = y.W + sum_bodies(wgt * wd);
    const X nTH = fma_(h6, fma_(T(2), THD3, fma_(T(2), THD2, y.THD)) + THD4, y.TH);
    y.THD = fma_(h6, fma_(T(2), a3, fma_(T(2), a2, a1)) + a4, y.THD);
    y.TH = nTH;
  } while (++it < P.substeps);
  y.Q = y.Q * rsqrt_(sum4(y.Q * y.Q));
  return team_tool_offset(P, c, y);
}

}  // namespace amenv_dev
