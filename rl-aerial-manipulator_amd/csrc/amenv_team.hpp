// amenv_team.hpp -- lane-TEAM step kernel for the hexacopter + z,x,x arm (BASELINE config 3) in the latency regime.
//
// Why.  At 4096 envs a one-lane-per-env launch is 64 wavefronts on a chip with 1024 SIMDs, and its length is ONE lane's instruction
// stream: ~4000 dependent-ish VALU instructions at ~4 clocks each (a lone wave issues one VALU instruction per 4 clocks).  The
// arithmetic itself would take the whole chip 0.2 us.  So the step is spread over more lanes instead of more envs:
//
//   one env = one DPP ROW of 16 lanes = 4 bodies (base, link 1, link 2, link 3) x 4 components (x, y, z, spare / quaternion w-z)
//   one wavefront = 4 envs, 4096 envs = 1024 wavefronts = one per SIMD
//
//   * 3-vectors live with ONE component per lane (lanes c = 0..2 of a quad), 3x3 matrices as three column registers with one ROW per
//     lane, quaternions in the 4 lanes of a quad.  Cross products, matrix-vector and matrix-matrix products, R I R^T, the adjugate
//     solve all become 3-5 instructions instead of 6-45: the operands of the other components come through DPP quad_perm modifiers
//     (folded into v_mul / v_add by the compiler, v_mov_b32_dpp otherwise) -- no LDS, no barrier, no readlane.
//   * the four bodies of the multibody system run the SAME generic code in the four quads of the row (per-lane constant registers hold
//     each body's mass / CoM / inertia; the chain across joint k is masked by (body > k) through multiplications by 0 / 1), and the
//     Newton-Euler sums over bodies are two DPP row_ror adds per register.
//   * per-env scalars (reward logic, state machine, reset) are computed redundantly by all 16 lanes from broadcast copies of the state,
//     with the SAME task_step / reset code as the one-lane kernels (amenv_model.hpp): every branch is uniform within a row, which is
//     what keeps DPP legal inside it (DPP reads of EXEC-disabled lanes return 0).
//
//   * (round 2) what the four quads of a row are spent on inside the RK4 changed: the product build gives quad s RK4 STAGE s of the joint-
//     configuration work (team_kin_stage: chain + three links, unmasked, aggregated to 14 registers) and evaluates the base dynamics on the
//     aggregates (team_dyn_agg) -- see "Stage-parallel form" below; the body-parallel form described in the two points above (team_kin /
//     team_dyn: quad b = body b, masked chain, sums over bodies per stage) is kept for A/B builds (-DAMENV_TEAM_BODY_PARALLEL).  Loads, stores,
//     mixer, task step, reset and the helper wave are the same in both.
//
// Same model, same expressions as amenv_arm.hpp; sums are associated differently (trees over lanes), so results agree with the
// one-lane kernel to rounding (tests: <= 2e-6 rel per step against the fp64 oracle, the same gate), not bit for bit.
#pragma once
#include "amenv_kernels.hpp"

namespace amenv_dev {

// ---- DPP helpers -------------------------------------------------------------------------------------------------------------
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ float qp(float v) {   // lane c of every quad reads lane P_c of its quad
  constexpr int ctrl = P0 | (P1 << 2) | (P2 << 4) | (P3 << 6);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true));
}
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ int qpi(int v) {
  constexpr int ctrl = P0 | (P1 << 2) | (P2 << 4) | (P3 << 6);
  return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xF, 0xF, true);
}
template <int N> __device__ __forceinline__ float row_ror(float v) {   // rotate within the 16-lane row
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, true));
}
template <int J> __device__ __forceinline__ float bc(float v) { return qp<J, J, J, J>(v); }          // component J to the whole quad
__device__ __forceinline__ float rot1(float v) { return qp<1, 2, 0, 3>(v); }                          // v[(c+1)%3]
__device__ __forceinline__ float rot2(float v) { return qp<2, 0, 1, 3>(v); }                          // v[(c+2)%3]
// Sum over the 4 quads of the row, BIT-IDENTICAL in all of them: opposite quads first (x0 + x2 and x2 + x0 are the same number), then
// the two pair sums (a + b = b + a).  With neighbours first each quad would associate the four terms differently, the replicated base
// state would drift apart between the quads by rounding, and a threshold of the task step could then be taken differently inside one row.
__device__ __forceinline__ float sum_bodies(float v) { v = v + row_ror<8>(v); return v + row_ror<4>(v); }
__device__ __forceinline__ float sum4(float p) { const float t = p + qp<1, 0, 3, 2>(p); return t + qp<2, 3, 0, 1>(t); }   // all 4 lanes valid in, all out
__device__ __forceinline__ float dot3(float a, float b) { const float p = a * b; return (p + rot1(p)) + rot2(p); }        // lanes 0..2
__device__ __forceinline__ float dot_all(float a, float b) { const float p = a * b; return (bc<0>(p) + bc<1>(p)) + bc<2>(p); }   // all 4 lanes

struct X3 { float v, r1, r2; };   // a 3-vector (component per lane) with its two rotations cached
__device__ __forceinline__ X3 x3(float v) { return X3{v, rot1(v), rot2(v)}; }
__device__ __forceinline__ float cross(const X3& a, const X3& b) { return fma_(a.r1, b.r2, -(a.r2 * b.r1)); }
struct TM { float c0, c1, c2; };  // 3x3 matrix: lane i holds row i, one register per column
__device__ __forceinline__ float matvec(const TM& M, float v) { return fma_(M.c0, bc<0>(v), fma_(M.c1, bc<1>(v), M.c2 * bc<2>(v))); }

// ---- parameters -------------------------------------------------------------------------------------------------------------
// Per-lane constants: a table in HBM (built by amenv_create), constant k of lane column 4*body + component; each lane loads its column
// once at kernel entry as 16-byte pieces (the loads overlap the state loads).
enum TeamConst {
  TC_I00 = 0, TC_I01, TC_I02, TC_I11, TC_I12, TC_I22,   // this body's inertia about its CoM, body-fixed frame (uniform within the quad)
  TC_LCX, TC_LCY, TC_LCZ,                              // this body's CoM in its frame (0 for the base)
  TC_MASS,                                             // this body's mass
  TC_MK0, TC_MK1, TC_MK2,                              // 1 if this body sits behind joint k (body > k), else 0
  TC_E0, TC_E1, TC_E2,                                 // 1 if component == j
  TC_ALLOC0, TC_MIX0 = TC_ALLOC0 + 6,                  // alloc[r][component], mix[component][r] (component = wrench entry F, Mx, My, Mz), r = 0..5
  TC_SP = TC_MIX0 + 6, TC_SQ, TC_SR,                   // signs of the quaternion kinematics (incl. the 1/2)
  TC_ACT1, TC_ACT2,                                    // action scaling u = (a * ACT1) * ACT2: (mass, ms, ms, ms), (g, 1, 1, 1) -- fp32, left to right
  TC_JHALF, TC_JMID,                                   // joint command = fma(action, half, mid), joint = component
  TC_O0,                                               // joint-1 origin (component per lane)
  TC_GV, TC_GV1, TC_GV2,                               // (0, 0, -g) and its two rotations
  TC_OBS_A, TC_OBS_B, TC_OBS_C,                        // observation scalings of the three row segments this lane writes
  kTeamConsts
};

struct TeamParams {          // wave-uniform (SGPRs)
  float o1[3], o2[3];        // joint-2 / joint-3 origins in their parent frames
  float tool[3];
  float kp, kd, amax, mtot, inv_mtot, g, h;
  float tmin[6], tmax[6];
  float ee_home[3];
  float lm[3], lcm[3][3], li[3][6], I0[6];   // stage-parallel form: link masses, CoMs (link frame), inertias (xx xy xz yy yz zz), base inertia
  int32_t substeps, max_steps, counter_limit, ee_task, K;   // K = 1 (the task code reads it)
  uint32_t flags;
  const float4* consts;      // [(kTeamConsts + 3) / 4][16] float4: constants 4k..4k+3 of lane column l at [k][l]
};

struct TeamState { float P, V, Q, W, TH, THD; };   // one register each: position, velocity (component per lane), quaternion (4 lanes), body rates, joints

__device__ __forceinline__ void sincos_t(float x, float& s, float& c) { s = __sinf(x); c = __cosf(x); }

// R <- R Rot(column AX, angle): the two other columns mix, lane-wise (row per lane)
template <int AX>
__device__ __forceinline__ void rotate_cols(TM& R, float s, float co) {
  float& a = AX == 0 ? R.c1 : (AX == 1 ? R.c2 : R.c0);
  float& b = AX == 0 ? R.c2 : (AX == 1 ? R.c0 : R.c1);
  const float ra = a, rb = b;
  a = fma_(co, ra, s * rb);
  b = fma_(co, rb, -(s * ra));
}

// cross product a x b when only `a` has its rotations cached: the other operand's rotations ride as DPP modifiers of the two
// multiplies (v_mul_f32_dpp; an FMA cannot carry one), so no v_mov_b32_dpp is spent on `b`
__device__ __forceinline__ float cross_c(const X3& a, float b) { return a.r1 * rot2(b) - a.r2 * rot1(b); }

// One RK4 stage is evaluated in two parts.  The joint servos do not feel the base, so the joint states of all four stages are known
// up front, and with them everything that depends on the joint configuration only (team_kin): chain kinematics, each body's CoM motion
// relative to the body frame, inertias in body axes, the composite inertia about the system CoM and its adjugate.  The four team_kin
// evaluations are independent of each other -- the compiler interleaves them, which is what a lone wavefront needs to issue at its
// full rate (one VALU instruction per 4 clocks when independent, ~5.5 when each waits for its predecessor) -- and only team_dyn (the
// terms with the base's attitude and rates, ~1/3 of the work) forms the serial chain stage 1 -> 2 -> 3 -> 4.
struct TeamKin {
  X3 r, u, w;        // this body: CoM position, velocity (relative to the body frame), angular velocity relative to the base
  float a_, al;      // CoM acceleration, angular acceleration (relative)
  TM J;              // inertia about the CoM, body axes
  X3 S;              // sum over bodies of m r
  TM C;              // adjugate of the composite inertia I_c about the system CoM (columns)
  float idet;        // 1 / det I_c
};

__device__ __forceinline__ float joint_accel(const TeamParams& P, float cmd, float th, float thd) {   // servo, joint k in lane k
  return clamp_(fma_(P.kp, cmd - th, -(P.kd * thd)), -P.amax, P.amax);
}

__device__ __forceinline__ TeamKin team_kin(const TeamParams& P, const float* c, float TH, float THD, float thdd) {
  TeamKin k;
  const float e0 = c[TC_E0], e1 = c[TC_E1], e2 = c[TC_E2];
  // ---- chain across the joints this body sits behind (masks mk: 1 behind joint k, else 0 -> angle, rate, acceleration, offset vanish)
  TM R;
  float p, pd, pdd, w, al;
  {   // joint 1 about z at the start of the chain: R = 1, p = pd = pdd = w = al = 0
    const float mk = c[TC_MK0];
    const float th = bc<0>(TH) * mk, td = bc<0>(THD) * mk, tdd = bc<0>(thdd) * mk;
    float s, co;
    sincos_t(th, s, co);
    p = mk * c[TC_O0]; pd = 0.0f; pdd = 0.0f;
    w = td * e2; al = tdd * e2;
    R.c0 = fma_(co, e0, s * e1); R.c1 = fma_(co, e1, -(s * e0)); R.c2 = e2;
  }
  auto advance_x = [&](float mk, float th, float td, float tdd, const float* o) {   // joint about its frame's x axis (column 0)
    float s, co;
    sincos_t(th, s, co);
    const float Ro = fma_(R.c0, o[0], fma_(R.c1, o[1], R.c2 * o[2]));
    const X3 xRo = x3(Ro), xw = x3(w);
    const float wRo = cross(xw, xRo);
    pd = fma_(mk, wRo, pd);
    pdd = fma_(mk, cross_c(xw, wRo) - cross_c(xRo, al), pdd);      // al x Ro + w x (w x Ro)
    p = fma_(mk, Ro, p);
    const float z = R.c0;
    const float wz = cross_c(xw, z);
    al = fma_(td, wz, fma_(tdd, z, al));
    w = fma_(td, z, w);
    rotate_cols<0>(R, s, co);
  };
  advance_x(c[TC_MK1], bc<1>(TH) * c[TC_MK1], bc<1>(THD) * c[TC_MK1], bc<1>(thdd) * c[TC_MK1], P.o1);
  advance_x(c[TC_MK2], bc<2>(TH) * c[TC_MK2], bc<2>(THD) * c[TC_MK2], bc<2>(thdd) * c[TC_MK2], P.o2);
  // ---- this body: CoM motion relative to the body frame, inertia in body axes
  const float m = c[TC_MASS];
  const float Rc = fma_(R.c0, c[TC_LCX], fma_(R.c1, c[TC_LCY], R.c2 * c[TC_LCZ]));
  k.w = x3(w); k.al = al;
  const X3 xRc = x3(Rc);
  const float wRc = cross(k.w, xRc);
  const float r = p + Rc;
  k.r = x3(r);
  k.u = x3(pd + wRc);
  k.a_ = pdd + cross_c(k.w, wRc) - cross_c(xRc, al);
  {   // J = R I R^T
    const float RI0 = fma_(R.c0, c[TC_I00], fma_(R.c1, c[TC_I01], R.c2 * c[TC_I02]));
    const float RI1 = fma_(R.c0, c[TC_I01], fma_(R.c1, c[TC_I11], R.c2 * c[TC_I12]));
    const float RI2 = fma_(R.c0, c[TC_I02], fma_(R.c1, c[TC_I12], R.c2 * c[TC_I22]));
    k.J.c0 = fma_(RI0, bc<0>(R.c0), fma_(RI1, bc<0>(R.c1), RI2 * bc<0>(R.c2)));
    k.J.c1 = fma_(RI0, bc<1>(R.c0), fma_(RI1, bc<1>(R.c1), RI2 * bc<1>(R.c2)));
    k.J.c2 = fma_(RI0, bc<2>(R.c0), fma_(RI1, bc<2>(R.c1), RI2 * bc<2>(R.c2)));
  }
  // ---- sums over the four bodies of the row: S = sum m r, I_O = sum J + m (|r|^2 1 - r r^T) (column by column)
  const float mr2 = m * dot3(r, r);
  const float S = sum_bodies(m * r);
  const float IO0 = sum_bodies(fma_(-m, r * bc<0>(r), fma_(mr2, e0, k.J.c0)));
  const float IO1 = sum_bodies(fma_(-m, r * bc<1>(r), fma_(mr2, e1, k.J.c1)));
  const float IO2 = sum_bodies(fma_(-m, r * bc<2>(r), fma_(mr2, e2, k.J.c2)));
  k.S = x3(S);
  // composite inertia about the system CoM, I_c = I_O - (|S|^2 1 - S S^T) / mtot, and its adjugate (columns = cross products of columns)
  const float im = P.inv_mtot;
  const float imS2 = im * dot3(S, S), imS = im * S;
  const X3 x0 = x3(fma_(imS, bc<0>(S), fma_(-imS2, e0, IO0)));
  const X3 x1 = x3(fma_(imS, bc<1>(S), fma_(-imS2, e1, IO1)));
  const X3 x2 = x3(fma_(imS, bc<2>(S), fma_(-imS2, e2, IO2)));
  k.C = TM{cross(x1, x2), cross(x2, x0), cross(x0, x1)};
  k.idet = rcp_(dot3(x0.v, k.C.c0));
  return k;
}

struct TeamDeriv { float V, Q, W; };   // derivatives of velocity, quaternion, body rates (position' = velocity, joints: known up front)

__device__ __forceinline__ TeamDeriv team_dyn(const TeamParams& P, const float* c, const TeamKin& k, float Q, float W, float F, float Mv) {
  TeamDeriv d;
  const float e2 = c[TC_E2], m = c[TC_MASS], im = P.inv_mtot;
  // attitude: |q|^2, vector part of q in component layout (with rotations), scalar part
  const float n2 = sum4(Q * Q);
  const float two_in2 = 2.0f * rcp_(n2);
  const X3 qv{qp<1, 2, 3, 3>(Q), qp<2, 3, 1, 3>(Q), qp<3, 1, 2, 3>(Q)};
  const float qw = bc<0>(Q);
  const X3 om = x3(W);
  // gravity in body components: Rq (0,0,-g) = v + (2/|q|^2) qv x (qv x v + qw v)
  const X3 gv{c[TC_GV], c[TC_GV1], c[TC_GV2]};
  const float gb = fma_(two_in2, cross_c(qv, fma_(qw, gv.v, cross(qv, gv))), gv.v);
  // this body's Newton-Euler terms that involve the base's angular velocity
  const float b_ = fma_(2.0f, cross(om, k.u), cross_c(om, cross(om, k.r))) + k.a_;      // w x (w x r) + 2 w x u + a
  const float aa = k.al + cross(om, k.w), Om = W + k.w.v;
  const float JOm = matvec(k.J, Om);
  float nb = fma_(m, cross_c(k.r, b_), matvec(k.J, aa)) + cross_c(x3(Om), JOm);
  float fb = m * b_;
  fb = sum_bodies(fb); nb = sum_bodies(nb);
  // external wrench about O, 3x3 solve with the prepared adjugate
  const float f = fma_(F, e2, fma_(P.mtot, gb, -fb));
  const float n = Mv + cross_c(k.S, gb) - nb;
  const float rhs = fma_(-im, cross_c(k.S, f), n);
  const float wd = k.idet * matvec(k.C, rhs);
  const float Aacc = im * (f + cross_c(k.S, wd));
  // world acceleration of O: Rq^T A = A + (2/|q|^2) qv x (qv x A - qw A)
  const float vd = fma_(two_in2, cross_c(qv, fma_(-qw, Aacc, cross_c(qv, Aacc))), Aacc);
  // quaternion kinematics (4 lanes): -1/2 Omega(w) q + 2 (1 - |q|^2) q
  float dq = fma_(-2.0f, n2, 2.0f) * Q;
  dq = fma_(c[TC_SP] * bc<0>(W), qp<1, 0, 3, 2>(Q), dq);
  dq = fma_(c[TC_SQ] * bc<1>(W), qp<2, 3, 0, 1>(Q), dq);
  dq = fma_(c[TC_SR] * bc<2>(W), qp<3, 2, 1, 0>(Q), dq);
  d.V = vd; d.Q = dq; d.W = wd;
  return d;
}

// ---- Stage-parallel form (round 2) -------------------------------------------------------------------------------------------------
// The body-parallel form above spends the four quads of a row on the four BODIES and walks the four RK4 stages' joint configurations
// one after the other (4 x team_kin), with masked chain steps (every quad executes all joint advances) and two DPP adds per summed
// register and stage.  With the aggregated form of the base dynamics (amenv_arm.hpp "Staged form": fb = w x (w x S) + 2 w x U + Aa,
// nb = w x (I_O w) + G w + w x H + Tn) a stage's joint configuration reaches the base dynamics as 14 registers that do not depend on
// the base's rates -- so the four quads take the four STAGES instead: quad s forms stage s's joint state and runs the chain and the
// three links for it, unmasked, all four stages at once (~450 instructions instead of 4 x ~200); then the serial part: every quad
// evaluates team_dyn_agg with ITS OWN aggregates on the common stage state, stage s's derivative is the one quad s computed, picked and
// broadcast with a multiply by the quad's 0 / 1 selector and the bit-identical row sum (x + 0 + 0 + 0 is exact in every quad).
struct TeamAgg {
  X3 S, U, H;        // sum m r, sum m u, sum J w_k (component per lane, rotations cached)
  float Aa, Tn;      // sum m a, sum m r x a + J al + w_k x (J w_k)
  TM IO, G, C;       // I_O, G (row per lane), adjugate of the composite inertia I_c divided by its determinant
};

__device__ __forceinline__ TeamAgg team_kin_stage(const TeamParams& P, const float* c, float TH, float THD, float thdd) {
  const float e0 = c[TC_E0], e1 = c[TC_E1], e2 = c[TC_E2];
  float S = 0.0f, U = 0.0f, Aa = 0.0f, H = 0.0f, Tn = 0.0f;
  TM IO{fma_(e0, P.I0[0], fma_(e1, P.I0[1], e2 * P.I0[2])), fma_(e0, P.I0[1], fma_(e1, P.I0[3], e2 * P.I0[4])),
        fma_(e0, P.I0[2], fma_(e1, P.I0[4], e2 * P.I0[5]))};          // base body: r = 0, J = I0
  TM G{0.0f, 0.0f, 0.0f};
  TM R;
  float p, pd = 0.0f, pdd = 0.0f, w, al;
  {   // joint 1 about z at the start of the chain
    float s, co;
    sincos_t(bc<0>(TH), s, co);
    p = c[TC_O0];
    w = bc<0>(THD) * e2; al = bc<0>(thdd) * e2;
    R.c0 = fma_(co, e0, s * e1); R.c1 = fma_(co, e1, -(s * e0)); R.c2 = e2;
  }
  auto link = [&](int k) {   // link k behind the joints advanced so far: CoM motion relative to the body frame, inertia in body axes, sums
    const float m = P.lm[k];
    const float Rc = fma_(R.c0, P.lcm[k][0], fma_(R.c1, P.lcm[k][1], R.c2 * P.lcm[k][2]));
    const X3 xw = x3(w), xRc = x3(Rc);
    const float wRc = cross(xw, xRc);
    const float r = p + Rc, u = pd + wRc;
    const float a_ = pdd + cross_c(xw, wRc) - cross_c(xRc, al);
    const float RI0 = fma_(R.c0, P.li[k][0], fma_(R.c1, P.li[k][1], R.c2 * P.li[k][2]));
    const float RI1 = fma_(R.c0, P.li[k][1], fma_(R.c1, P.li[k][3], R.c2 * P.li[k][4]));
    const float RI2 = fma_(R.c0, P.li[k][2], fma_(R.c1, P.li[k][4], R.c2 * P.li[k][5]));
    const TM J{fma_(RI0, bc<0>(R.c0), fma_(RI1, bc<0>(R.c1), RI2 * bc<0>(R.c2))), fma_(RI0, bc<1>(R.c0), fma_(RI1, bc<1>(R.c1), RI2 * bc<1>(R.c2))),
               fma_(RI0, bc<2>(R.c0), fma_(RI1, bc<2>(R.c1), RI2 * bc<2>(R.c2)))};
    S = fma_(m, r, S); U = fma_(m, u, U); Aa = fma_(m, a_, Aa);
    const float mr2 = m * dot3(r, r), mr = m * r;
    IO.c0 += fma_(-mr, bc<0>(r), fma_(mr2, e0, J.c0));
    IO.c1 += fma_(-mr, bc<1>(r), fma_(mr2, e1, J.c1));
    IO.c2 += fma_(-mr, bc<2>(r), fma_(mr2, e2, J.c2));
    // G += 2 m ((u . r) 1 - u r^T) - (D + D^T), D = J [w]x: column j of D is J (w x e_j), column j of D^T is -w x (column j of J)
    const float m2 = m + m, d = m2 * dot3(u, r), mu = m2 * u;
    const float wx = bc<0>(w), wy = bc<1>(w), wz = bc<2>(w);
    G.c0 += fma_(-mu, bc<0>(r), d * e0) - (fma_(wz, J.c1, -(wy * J.c2)) - cross_c(xw, J.c0));
    G.c1 += fma_(-mu, bc<1>(r), d * e1) - (fma_(wx, J.c2, -(wz * J.c0)) - cross_c(xw, J.c1));
    G.c2 += fma_(-mu, bc<2>(r), d * e2) - (fma_(wy, J.c0, -(wx * J.c1)) - cross_c(xw, J.c2));
    const float Jw = matvec(J, w);
    H += Jw;
    Tn += fma_(m, cross_c(x3(r), a_), matvec(J, al)) + cross_c(xw, Jw);
  };
  auto advance_x = [&](float th, float td, float tdd, const float* o) {   // across a joint about its frame's x axis (column 0 of R)
    float s, co;
    sincos_t(th, s, co);
    const float Ro = fma_(R.c0, o[0], fma_(R.c1, o[1], R.c2 * o[2]));
    const X3 xRo = x3(Ro), xw = x3(w);
    const float wRo = cross(xw, xRo);
    pd += wRo;
    pdd += cross_c(xw, wRo) - cross_c(xRo, al);      // al x Ro + w x (w x Ro)
    p += Ro;
    const float z = R.c0;
    const float wz = cross_c(xw, z);
    al = fma_(td, wz, fma_(tdd, z, al));
    w = fma_(td, z, w);
    rotate_cols<0>(R, s, co);
  };
  link(0);
  advance_x(bc<1>(TH), bc<1>(THD), bc<1>(thdd), P.o1);
  link(1);
  advance_x(bc<2>(TH), bc<2>(THD), bc<2>(thdd), P.o2);
  link(2);
  TeamAgg k;
  k.S = x3(S); k.U = x3(U); k.H = x3(H); k.Aa = Aa; k.Tn = Tn; k.IO = IO; k.G = G;
  // composite inertia about the system CoM, I_c = I_O - (|S|^2 1 - S S^T) / mtot, its adjugate (columns = cross products of columns) / det
  const float im = P.inv_mtot;
  const float imS2 = im * dot3(S, S), imS = im * S;
  const X3 x0 = x3(fma_(imS, bc<0>(S), fma_(-imS2, e0, IO.c0)));
  const X3 x1 = x3(fma_(imS, bc<1>(S), fma_(-imS2, e1, IO.c1)));
  const X3 x2 = x3(fma_(imS, bc<2>(S), fma_(-imS2, e2, IO.c2)));
  const TM A{cross(x1, x2), cross(x2, x0), cross(x0, x1)};
  const float idet = rcp_(dot3(x0.v, A.c0));
  k.C = TM{idet * A.c0, idet * A.c1, idet * A.c2};
  return k;
}

__device__ __forceinline__ TeamDeriv team_dyn_agg(const TeamParams& P, const float* c, const TeamAgg& k, float Q, float W, float F, float Mv) {
  TeamDeriv d;
  const float e2 = c[TC_E2], im = P.inv_mtot;
  const float n2 = sum4(Q * Q);
  const float two_in2 = 2.0f * rcp_(n2);
  const X3 qv{qp<1, 2, 3, 3>(Q), qp<2, 3, 1, 3>(Q), qp<3, 1, 2, 3>(Q)};
  const float qw = bc<0>(Q);
  const X3 om = x3(W);
  const X3 gv{c[TC_GV], c[TC_GV1], c[TC_GV2]};
  const float gb = fma_(two_in2, cross_c(qv, fma_(qw, gv.v, cross(qv, gv))), gv.v);
  const float fb = fma_(2.0f, cross(om, k.U), cross_c(om, cross(om, k.S))) + k.Aa;
  const float nb = (cross_c(om, matvec(k.IO, W)) + matvec(k.G, W)) + (cross(om, k.H) + k.Tn);
  const float f = fma_(F, e2, fma_(P.mtot, gb, -fb));
  const float n = Mv + cross_c(k.S, gb) - nb;
  const float rhs = fma_(-im, cross_c(k.S, f), n);
  const float wd = matvec(k.C, rhs);
  const float Aacc = im * (f + cross_c(k.S, wd));
  const float vd = fma_(two_in2, cross_c(qv, fma_(-qw, Aacc, cross_c(qv, Aacc))), Aacc);
  float dq = fma_(-2.0f, n2, 2.0f) * Q;
  dq = fma_(c[TC_SP] * bc<0>(W), qp<1, 0, 3, 2>(Q), dq);
  dq = fma_(c[TC_SQ] * bc<1>(W), qp<2, 3, 0, 1>(Q), dq);
  dq = fma_(c[TC_SR] * bc<2>(W), qp<3, 2, 1, 0>(Q), dq);
  d.V = vd; d.Q = dq; d.W = wd;
  return d;
}
// the derivative quad `sel` (1 in that quad, 0 elsewhere) computed, in all four quads of the row, bit-identical
__device__ __forceinline__ TeamDeriv team_pick(const TeamDeriv& d, float sel) {
  return TeamDeriv{sum_bodies(d.V * sel), sum_bodies(d.Q * sel), sum_bodies(d.W * sel)};
}

// tool point relative to the body origin, world axes, of a (unit-quaternion) state: chain positions in the link-3 quad, summed over the row
__device__ __forceinline__ float team_tool_offset(const TeamParams& P, const float* c, const TeamState& y) {
  const float e0 = c[TC_E0], e1 = c[TC_E1], e2 = c[TC_E2];
  float s, co;
  sincos_t(bc<0>(y.TH), s, co);
  TM R{fma_(co, e0, s * e1), fma_(co, e1, -(s * e0)), e2};
  float p = c[TC_O0];
  p = p + fma_(R.c0, P.o1[0], fma_(R.c1, P.o1[1], R.c2 * P.o1[2]));
  sincos_t(bc<1>(y.TH), s, co);
  rotate_cols<0>(R, s, co);
  p = p + fma_(R.c0, P.o2[0], fma_(R.c1, P.o2[1], R.c2 * P.o2[2]));
  sincos_t(bc<2>(y.TH), s, co);
  rotate_cols<0>(R, s, co);
  p = p + fma_(R.c0, P.tool[0], fma_(R.c1, P.tool[1], R.c2 * P.tool[2]));
  // world = Rq^T body (|q| = 1)
  const X3 qv{qp<1, 2, 3, 3>(y.Q), qp<2, 3, 1, 3>(y.Q), qp<3, 1, 2, 3>(y.Q)};
  const float qw = bc<0>(y.Q);
  return fma_(2.0f, cross_c(qv, fma_(-qw, p, cross_c(qv, p))), p);
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, each XCD has its own L2).  A team wavefront touches
// 64 B of every state group; with the identity mapping the two halves of each 128-B line would be fetched by two different XCDs.
// Blocks that share an XCD therefore take CONSECUTIVE 4-env groups (speed only: any mapping is correct).  gridDim.x is a multiple of 64.
__device__ __forceinline__ int team_group_of_block(int b, int nblocks) { return (b & 7) * (nblocks >> 3) + (b >> 3); }

// ---- one env row's registers, and the per-lane role of a lane in its team -------------------------------------------------------------
struct TeamEnv {
  TeamState y;                                   // dynamic state (component per lane, replicated in the four quads)
  float WP;                                      // waypoint (component per lane)
  float final_yaw, last_distance, ep_return;     // per-env scalars, replicated in all 16 lanes
  int32_t step, counter, flags, episode;
};

struct TeamLane {
  int lane, cc, bb;
  bool q0, q1, q2, lead;
  uint32_t offA, offB, offC;                     // observation columns this lane writes in the three row segments
  bool okA, okB, okC;
  float c[((kTeamConsts + 3) / 4) * 4];          // this lane's constants
  __device__ __forceinline__ void init(const TeamParams& P) {
    lane = int(threadIdx.x) & 63; cc = lane & 3; bb = (lane >> 2) & 3;
    q0 = bb == 0; q1 = bb == 1; q2 = bb == 2; lead = (lane & 15) == 0;
    // A = [p/10 | v/5 | q | w/5] by quad, B = [(wp - task point)/2 | 0 | yaw/pi | th/pi], C = [thd/5 | tool offset*2]
    offA = q0 ? cc : (q1 ? 3 + cc : (q2 ? 6 + cc : 10 + cc));
    offB = q0 ? 13 + cc : (q1 ? 16 + cc : (q2 ? 19 : 20 + cc));
    offC = q0 ? 23 + cc : 26 + cc;
    okA = cc < 3 || q2; okB = q2 ? cc == 0 : cc < 3; okC = cc < 3 && bb < 2;
    constexpr int NC4 = (kTeamConsts + 3) / 4;
#pragma unroll
    for (int k = 0; k < NC4; k++) {
      const float4 v = P.consts[k * 16 + (lane & 15)];
      c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
    }
  }
};

// state of env i from its tile: lane c of every quad reads component c of each group; the per-env scalars ride in slot 3 of the
// p / v / w groups
__device__ __forceinline__ void team_load(const char* tile, int i, const TeamLane& L, TeamEnv& E) {
  const uint32_t eoff = uint32_t(i & 63) * 16u + uint32_t(L.cc) * 4u;
  auto gload = [&](int g) { return *reinterpret_cast<const float*>(tile + kIntBytes + uint32_t(g) * 1024u + eoff); };
  E.y = TeamState{gload(0), gload(1), gload(2), gload(3), gload(5), gload(6)};
  E.WP = gload(4);
  const int4 iv = *(reinterpret_cast<const int4*>(tile) + (i & 63));
  E.final_yaw = bc<3>(E.y.P); E.last_distance = bc<3>(E.y.V); E.ep_return = bc<3>(E.y.W);
  E.step = iv.x; E.counter = iv.y; E.flags = iv.z; E.episode = iv.w;
}

// Two stores: (1) quad b writes group b (p|yaw, v|last_distance, q, w|return); (2) quad 0 the joint angles, quad 1 the joint rates, quad 2
// the int plane, quad 3 the waypoint group -- per-lane offsets; the step kernel masks quad 3 off between resets (16 B per env-step of write
// traffic; the rollout kernels store once per launch).
__device__ __forceinline__ void team_store(char* tile, int i, const TeamLane& L, const TeamEnv& E, bool store_wp = true) {
  const uint32_t eoff = uint32_t(i & 63) * 16u + uint32_t(L.cc) * 4u;
  const bool l3 = L.cc == 3;
  const float g0 = l3 ? E.final_yaw : E.y.P, g1 = l3 ? E.last_distance : E.y.V, g3 = l3 ? E.ep_return : E.y.W;
  const float sv = L.q0 ? g0 : (L.q1 ? g1 : (L.q2 ? E.y.Q : g3));
  *reinterpret_cast<float*>(tile + kIntBytes + uint32_t(L.bb) * 1024u + eoff) = sv;
  const int ival = L.cc == 0 ? E.step : (L.cc == 1 ? E.counter : (L.cc == 2 ? E.flags : E.episode));
  const float s2 = L.q0 ? E.y.TH : (L.q1 ? E.y.THD : (L.q2 ? __int_as_float(ival) : E.WP));
  const uint32_t off2 = L.q2 ? eoff : kIntBytes + (L.q0 ? 5u : (L.q1 ? 6u : 4u)) * 1024u + eoff;   // (the int plane has the same 16 B per env)
  if (L.q0 || L.q1 || L.q2 || store_wp) *reinterpret_cast<float*>(tile + off2) = s2;   // the waypoint group (quad 3) changes only at a reset (uniform within the row)
}

struct TeamOut { float reward; uint32_t bits; float vA, vB, vC; bool ended; int ep_len; float ep_ret; };

// What a reset leaves in this lane's registers: position / waypoint component, final yaw, and the three observation values of the reset
// state (at rest, level, arm at home).  Lane c < 3 of every quad computes Philox block c; the 12 words are then broadcast inside the quad.
// Pure function of (seed, global env id, episode): the step kernel's helper wave evaluates it for every row while the main wave integrates.
struct TeamReset { float P, WP, final_yaw, vA, vB, vC; };
__device__ __forceinline__ TeamReset team_reset(const TeamParams& P, const ColdParams& C, const TeamLane& L, int32_t episode, int i) {
  const float* c = L.c;
  uint32_t wds[4];
  const int64_t gid = C.gid0 + i;
  philox4x32_10(C.seed_lo, C.seed_hi, uint32_t(uint64_t(gid)), uint32_t(uint64_t(gid) >> 32), uint32_t(episode), uint32_t(L.cc), wds);
  uint32_t r[12];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    r[k] = uint32_t(qpi<0, 0, 0, 0>(int(wds[k]))); r[4 + k] = uint32_t(qpi<1, 1, 1, 1>(int(wds[k]))); r[8 + k] = uint32_t(qpi<2, 2, 2, 2>(int(wds[k])));
  }
  Env<float, 1> e;
  e.episode = episode;
  reset_from_words<float, 1>(C, 1, e, r);
  const float e0 = c[TC_E0], e1 = c[TC_E1], e2 = c[TC_E2];
  TeamReset R;
  R.P = fma_(e0, e.px, fma_(e1, e.py, e2 * e.pz));
  R.WP = fma_(e0, e.wp[0][0], fma_(e1, e.wp[0][1], e2 * e.wp[0][2]));
  R.final_yaw = e.final_yaw;
  const float eo = fma_(e0, P.ee_home[0], fma_(e1, P.ee_home[1], e2 * P.ee_home[2]));
  const float zero = 0.0f, Q = L.cc == 0 ? 1.0f : 0.0f;
  R.vA = (L.q0 ? R.P : (L.q1 ? zero : (L.q2 ? Q : zero))) * c[TC_OBS_A];
  const float tp = P.ee_task != 0 ? R.P + eo : R.P;
  R.vB = (L.q0 ? R.WP - tp : (L.q1 ? 0.0f : (L.q2 ? R.final_yaw : zero))) * c[TC_OBS_B];
  R.vC = (L.q0 ? zero : eo) * c[TC_OBS_C];
  return R;
}

// One control step of one env row, state in registers: mixer -> RK4 -> forward kinematics -> task step -> (episode end: Monitor outputs,
// reset) -> observation values.  act: wrench action a0..a3 (one per lane), actj: joint commands (joint per lane).
// A reset row's registers from team_reset's values; the rest of WaypointQuadEnv.reset's result is constant (at rest, level, arm at home,
// counters cleared, episode + 1).  `e`: the scalar copies team_advance writes back into E afterwards.
__device__ __forceinline__ void team_apply_reset(const TeamLane& L, const TeamReset& R, TeamEnv& E, Env<float, 1>& e, TeamOut& o) {
  TeamState& y = E.y;
  y.P = R.P; y.V = 0.0f; y.W = 0.0f; y.TH = 0.0f; y.THD = 0.0f;
  y.Q = L.cc == 0 ? 1.0f : 0.0f;
  E.WP = R.WP; E.final_yaw = R.final_yaw;
  e.last_distance = -1.0f; e.ep_return = 0.0f;
  e.step = 0; e.counter = 0; e.flags = 0; e.episode += 1;
  o.vA = R.vA; o.vB = R.vB; o.vC = R.vC;
  o.bits |= AMENV_INFO_WAS_RESET;
}

// HELPED: the caller's helper wave does the episode-end work (step kernel); this function then stops after the task step and the
// observation values, with o.ended / o.ep_len / o.ep_ret set.
template <int NROT, bool HELPED = false>
__device__ __forceinline__ TeamOut team_advance(const TeamParams& P, const ColdParams& C, const TeamLane& L, TeamEnv& E, float act, float actj, int i, bool active,
                                                float* terminal_obs, float* ep_return_out, int32_t* ep_len_out) {
  constexpr int OD = 29;
  const float* c = L.c;
  TeamState& y = E.y;
  // mixer -> per-rotor clamp -> re-mix (quadcopter.py:109-112); wrench entry per lane
  const float uu = (act * c[TC_ACT1]) * c[TC_ACT2];
  float wr = 0.0f;
#pragma unroll
  for (int r = 0; r < NROT; r++) {
    float t = sum4(c[TC_ALLOC0 + r] * uu);
    t = clamp_(t, P.tmin[r], P.tmax[r]);
    wr = fma_(c[TC_MIX0 + r], t, wr);
  }
  const float F = bc<0>(wr), Mv = qp<1, 2, 3, 3>(wr);
  const float cmd = __builtin_fmaf(actj, c[TC_JHALF], c[TC_JMID]);
  // RK4 (running weighted sum).  Joint stage states first (the servos do not feel the base), then the four joint-configuration parts,
  // then the serial chain of base-dependent parts.
  const float h = P.h, hh = 0.5f * h, h6 = h * (1.0f / 6.0f);
  int it = 0;
  do {
    const float a1 = joint_accel(P, cmd, y.TH, y.THD);
    const float TH2 = fma_(hh, y.THD, y.TH), THD2 = fma_(hh, a1, y.THD), a2 = joint_accel(P, cmd, TH2, THD2);
    const float TH3 = fma_(hh, THD2, y.TH), THD3 = fma_(hh, a2, y.THD), a3 = joint_accel(P, cmd, TH3, THD3);
    const float TH4 = fma_(h, THD3, y.TH), THD4 = fma_(h, a3, y.THD), a4 = joint_accel(P, cmd, TH4, THD4);
#ifdef AMENV_TEAM_BODY_PARALLEL   // A/B build (tools/build_variant.py): the body-parallel form
    const TeamKin k1 = team_kin(P, c, y.TH, y.THD, a1), k2 = team_kin(P, c, TH2, THD2, a2), k3 = team_kin(P, c, TH3, THD3, a3),
                  k4 = team_kin(P, c, TH4, THD4, a4);
    TeamDeriv d = team_dyn(P, c, k1, y.Q, y.W, F, Mv);
    float aP = y.V, aV = d.V, aQ = d.Q, aW = d.W;                       // acc = k1
    float sV = fma_(hh, d.V, y.V), sQ = fma_(hh, d.Q, y.Q), sW = fma_(hh, d.W, y.W);
    d = team_dyn(P, c, k2, sQ, sW, F, Mv);
    aP = fma_(2.0f, sV, aP); aV = fma_(2.0f, d.V, aV); aQ = fma_(2.0f, d.Q, aQ); aW = fma_(2.0f, d.W, aW);
    sV = fma_(hh, d.V, y.V); sQ = fma_(hh, d.Q, y.Q); sW = fma_(hh, d.W, y.W);
    d = team_dyn(P, c, k3, sQ, sW, F, Mv);
    aP = fma_(2.0f, sV, aP); aV = fma_(2.0f, d.V, aV); aQ = fma_(2.0f, d.Q, aQ); aW = fma_(2.0f, d.W, aW);
    sV = fma_(h, d.V, y.V); sQ = fma_(h, d.Q, y.Q); sW = fma_(h, d.W, y.W);
    d = team_dyn(P, c, k4, sQ, sW, F, Mv);
#else
    // quad s of the row takes RK4 stage s: its joint state, then the aggregates of that joint configuration (all four stages at once)
    const float s0 = L.q0 ? 1.0f : 0.0f, s1 = L.q1 ? 1.0f : 0.0f, s2 = L.q2 ? 1.0f : 0.0f, s3 = (L.q0 || L.q1 || L.q2) ? 0.0f : 1.0f;
    const float THs = L.q0 ? y.TH : (L.q1 ? TH2 : (L.q2 ? TH3 : TH4)), THDs = L.q0 ? y.THD : (L.q1 ? THD2 : (L.q2 ? THD3 : THD4));
    const float as = L.q0 ? a1 : (L.q1 ? a2 : (L.q2 ? a3 : a4));
    const TeamAgg k = team_kin_stage(P, c, THs, THDs, as);
    TeamDeriv d = team_pick(team_dyn_agg(P, c, k, y.Q, y.W, F, Mv), s0);
    float aP = y.V, aV = d.V, aQ = d.Q, aW = d.W;                       // acc = k1
    float sV = fma_(hh, d.V, y.V), sQ = fma_(hh, d.Q, y.Q), sW = fma_(hh, d.W, y.W);
    d = team_pick(team_dyn_agg(P, c, k, sQ, sW, F, Mv), s1);
    aP = fma_(2.0f, sV, aP); aV = fma_(2.0f, d.V, aV); aQ = fma_(2.0f, d.Q, aQ); aW = fma_(2.0f, d.W, aW);
    sV = fma_(hh, d.V, y.V); sQ = fma_(hh, d.Q, y.Q); sW = fma_(hh, d.W, y.W);
    d = team_pick(team_dyn_agg(P, c, k, sQ, sW, F, Mv), s2);
    aP = fma_(2.0f, sV, aP); aV = fma_(2.0f, d.V, aV); aQ = fma_(2.0f, d.Q, aQ); aW = fma_(2.0f, d.W, aW);
    sV = fma_(h, d.V, y.V); sQ = fma_(h, d.Q, y.Q); sW = fma_(h, d.W, y.W);
    d = team_pick(team_dyn_agg(P, c, k, sQ, sW, F, Mv), s3);
#endif
    y.P = fma_(h6, aP + sV, y.P); y.V = fma_(h6, aV + d.V, y.V); y.Q = fma_(h6, aQ + d.Q, y.Q); y.W = fma_(h6, aW + d.W, y.W);
    const float nTH = fma_(h6, fma_(2.0f, THD3, fma_(2.0f, THD2, y.THD)) + THD4, y.TH);
    y.THD = fma_(h6, fma_(2.0f, a3, fma_(2.0f, a2, a1)) + a4, y.THD);
    y.TH = nTH;
  } while (++it < P.substeps);
  y.Q = y.Q * rsqrt_(sum4(y.Q * y.Q));
  const float EO = team_tool_offset(P, c, y);
  // ---- task step: the one-lane kernels' code on broadcast copies of the state (identical in the 16 lanes of a row)
  Env<float, 1> e;
  e.px = bc<0>(y.P); e.py = bc<1>(y.P); e.pz = bc<2>(y.P);
  e.vx = bc<0>(y.V); e.vy = bc<1>(y.V); e.vz = bc<2>(y.V);
  e.qw = bc<0>(y.Q); e.qx = bc<1>(y.Q); e.qy = bc<2>(y.Q); e.qz = bc<3>(y.Q);
  e.wx = bc<0>(y.W); e.wy = bc<1>(y.W); e.wz = bc<2>(y.W);
  e.wp[0][0] = bc<0>(E.WP); e.wp[0][1] = bc<1>(E.WP); e.wp[0][2] = bc<2>(E.WP);
  e.eox = bc<0>(EO); e.eoy = bc<1>(EO); e.eoz = bc<2>(EO);
#pragma unroll
  for (int k = 0; k < 3; k++) { e.th[k] = 0.0f; e.thd[k] = 0.0f; }    // (joints stay in team registers; the task code does not read them)
  e.final_yaw = E.final_yaw; e.last_distance = E.last_distance; e.ep_return = E.ep_return;
  e.step = E.step; e.counter = E.counter; e.flags = E.flags; e.episode = E.episode;
  TeamOut o;
  o.bits = task_step<float, 1, true>(P, e, o.reward);
  e.ep_return += o.reward;
  auto obs_vals = [&](const TeamState& z, float wpv, float eo, float fyaw) {
    o.vA = (L.q0 ? z.P : (L.q1 ? z.V : (L.q2 ? z.Q : z.W))) * c[TC_OBS_A];
    const float tp = P.ee_task != 0 ? z.P + eo : z.P;
    o.vB = (L.q0 ? wpv - tp : (L.q1 ? 0.0f : (L.q2 ? fyaw : z.TH))) * c[TC_OBS_B];
    o.vC = (L.q0 ? z.THD : eo) * c[TC_OBS_C];
  };
  obs_vals(y, E.WP, EO, E.final_yaw);
  o.ended = (o.bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
  const bool resets = o.ended && (P.flags & AMENV_FLAG_AUTO_RESET);
  o.ep_len = 0; o.ep_ret = 0.0f;
  if (o.ended) {   // uniform within the row; SB3 DummyVecEnv + Monitor contract
    o.ep_len = e.step; o.ep_ret = e.ep_return;
    if constexpr (!HELPED) {
      if (active) {
        if (terminal_obs) {
          const uint32_t row = uint32_t(i) * OD;
          if (L.okA) terminal_obs[row + L.offA] = o.vA;
          if (L.okB) terminal_obs[row + L.offB] = o.vB;
          if (L.okC) terminal_obs[row + L.offC] = o.vC;
        }
        if (L.lead) {
          if (ep_return_out) ep_return_out[i] = o.ep_ret;
          if (ep_len_out) ep_len_out[i] = o.ep_len;
        }
      }
      if (resets) {
        const TeamReset R = team_reset(P, C, L, e.episode, i);
        team_apply_reset(L, R, E, e, o);
      }
    }
  }
  E.last_distance = e.last_distance; E.ep_return = e.ep_return;
  E.step = e.step; E.counter = e.counter; E.flags = e.flags; E.episode = e.episode;
  return o;
}

// per-step outputs; OPT = the caller may pass null pointers (rollout kernels), else all are present (step kernel: no pointer tests)
template <bool OPT>
__device__ __forceinline__ void team_store_outputs(const TeamLane& L, const TeamOut& o, uint32_t i, bool active, float* __restrict__ obs, float* __restrict__ reward_out,
                                                   uint8_t* __restrict__ done, uint32_t* __restrict__ info) {
  const uint32_t row = i * 29u;
  const bool ob = active && (!OPT || obs != nullptr);
  if (ob && L.okA) obs[row + L.offA] = o.vA;
  if (ob && L.okB) obs[row + L.offB] = o.vB;
  if (ob && L.okC) obs[row + L.offC] = o.vC;
  if (active && L.lead) {
    if (!OPT || reward_out) reward_out[i] = o.reward;
    if (!OPT || done) done[i] = o.ended ? 1 : 0;
    if (!OPT || info) info[i] = o.bits;
  }
}

// One control step of 4 envs per main wavefront.  grid = n_tiles * 16 workgroups of 128 threads: wave 0 integrates, wave 1 helps with
// episode ends.  At 4096 envs ~10 of the 1024 main waves see an episode end in every launch, the launch is as slow as its slowest wave,
// and episode-end code (rarely run on any one CU) costs ~7 clocks per instruction: with everything inline those waves ran 2,100 clocks
// (0.9 us of 5.8) longer than the rest (tools/stamp_team.py).  So the helper wave evaluates team_reset for all four rows while the main
// wave integrates, leaves the values in LDS and loads the workgroup's replica of the Monitor totals; after the kernel's ONE barrier the
// main wave's episode-end path is six LDS reads (+ three terminal-observation stores after its regular stores), and the helper writes
// Monitor's return / length and adds the ended episodes to its replica (owned for the launch when the grid has at most kStatsReplicas
// workgroups: plain read-modify-write; otherwise atomics).
template <int NROT>
__global__ __launch_bounds__(128) void step_kernel_team(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, const float* __restrict__ actions,
                                                        float* __restrict__ obs, float* __restrict__ reward_out, uint8_t* __restrict__ done,
                                                        uint32_t* __restrict__ info, const StepTail tl, const ColdParams C, const TeamParams P) {
  static_assert(NROT == 6, "team kernel: 6-rotor airframe");
  constexpr int AD = 7, OD = 29;
  __shared__ float rst[6][64];                     // team_reset's values, lane for lane
  __shared__ uint32_t fl[4][4];                    // per row: bit 0 ended on a real env, bit 1 reset | info bits | length | return
  __shared__ unsigned long long acc[S_COUNT];      // this launch's additions to the Monitor totals
#ifdef AMENV_STAMPS
  unsigned long long stamps_[kStampSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const int role = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  TeamLane L;
  L.init(P);
  const int row = L.lane >> 4;
  const int i = team_group_of_block(int(blockIdx.x), int(gridDim.x)) * 4 + row;   // env of this row
  const bool active = i < n_envs;
  char* tile = static_cast<char*>(blob) + size_t(i >> 6) * tile_bytes;
  if (role == 1) {
    const bool owned = gridDim.x <= kStatsReplicas;   // wave-uniform
    unsigned long long* totals = tl.stats + size_t(blockIdx.x & (kStatsReplicas - 1)) * kStatsStride;
    unsigned long long mine = 0ull;
    if (L.lane < S_COUNT) { if (owned) mine = totals[L.lane]; acc[L.lane] = 0ull; }
    const int32_t episode = (reinterpret_cast<const int4*>(tile) + (i & 63))->w;
    const TeamReset R = team_reset(P, C, L, episode, i);
    rst[0][L.lane] = R.P; rst[1][L.lane] = R.WP; rst[2][L.lane] = R.final_yaw;
    rst[3][L.lane] = R.vA; rst[4][L.lane] = R.vB; rst[5][L.lane] = R.vC;
    __syncthreads();
    const bool ended = (fl[row][0] & 1u) != 0 && L.lead;   // one lane per ended row
    if (__ballot(ended) != 0ull) {   // wave-uniform
      if (ended) {
        const uint32_t bits = fl[row][1];
        const int ep_len = int(fl[row][2]);
        const float ep_ret = __uint_as_float(fl[row][3]);
        if (tl.ep_return) tl.ep_return[i] = ep_ret;
        if (tl.ep_len) tl.ep_len[i] = ep_len;
        if (owned) accumulate_stats_lane(acc, 0, bits, ep_len, ep_ret);               // LDS adds
        else accumulate_stats_lane(tl.stats, int(blockIdx.x), bits, ep_len, ep_ret);  // global atomics
      }
      if (owned) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (L.lane < S_COUNT) totals[L.lane] = mine + acc[L.lane];
      }
    }
    return;
  }
  AMENV_STAMP(0);
  const int ia = active ? i : n_envs - 1;                   // rows past the end redo the last env's arithmetic (their outputs are masked)
  TeamEnv E;
  team_load(tile, i, L, E);
  const float act = actions[size_t(ia) * AD + L.cc];                                     // a0..a3: one per lane
  const float actj = actions[size_t(ia) * AD + 4 + (L.cc < 3 ? L.cc : 2)];                // joint commands a4..a6
  AMENV_STAMP(1);          // loads issued
  AMENV_STAMP_DRAIN();
  AMENV_STAMP(2);          // loads landed
  TeamOut o = team_advance<NROT, true>(P, C, L, E, act, actj, i, active, nullptr, nullptr, nullptr);
  const bool resets = o.ended && (P.flags & AMENV_FLAG_AUTO_RESET);
  if (L.lead) {
    fl[row][0] = (o.ended && active ? 1u : 0u) | (resets ? 2u : 0u);
    if (o.ended) { fl[row][1] = o.bits; fl[row][2] = uint32_t(o.ep_len); fl[row][3] = __float_as_uint(o.ep_ret); }
  }
  AMENV_STAMP(4);          // mixer + RK4 + forward kinematics + task step
#ifdef AMENV_STAMPS
  stamps_[3] = __ballot(o.ended && active) != 0ull ? 1ull : 0ull;   // (not a time) did one of this wave's envs end its episode?
#endif
  __syncthreads();
  const float tA = o.vA, tB = o.vB, tC = o.vC;              // the terminal observation of a row that ended
  if (resets) {
    const TeamReset R{rst[0][L.lane], rst[1][L.lane], rst[2][L.lane], rst[3][L.lane], rst[4][L.lane], rst[5][L.lane]};
    Env<float, 1> e;
    e.episode = E.episode;
    team_apply_reset(L, R, E, e, o);
    E.last_distance = e.last_distance; E.ep_return = e.ep_return;
    E.step = e.step; E.counter = e.counter; E.flags = e.flags; E.episode = e.episode;
  }
  AMENV_STAMP(5);          // barrier + reset values
  team_store(tile, i, L, E, resets);
  team_store_outputs<false>(L, o, uint32_t(i), active, obs, reward_out, done, info);
  if (o.ended && active && tl.terminal_obs) {
    const uint32_t r0 = uint32_t(i) * OD;
    if (L.okA) tl.terminal_obs[r0 + L.offA] = tA;
    if (L.okB) tl.terminal_obs[r0 + L.offB] = tB;
    if (L.okC) tl.terminal_obs[r0 + L.offC] = tC;
  }
  AMENV_STAMP(6);          // stores issued
#ifdef AMENV_STAMPS
  AMENV_STAMP_DRAIN();
  AMENV_STAMP(7);          // stores acknowledged
  if (L.lane == 0 && blockIdx.x < kStampWaves)
    for (int kk = 0; kk < kStampSlots; kk++) tl.stats[kStampBase + blockIdx.x * kStampSlots + kk] = stamps_[kk];
#endif
}

// n_steps control steps per launch with open-loop actions [T][N][7]; per-step outputs [T][N]... (any of them may be null).  State and
// constants stay in registers across steps: no launch boundary, no prologue, no state traffic between steps.
template <int NROT>
__global__ __launch_bounds__(64) void rollout_kernel_team(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, const float* __restrict__ actions,
                                                          float* __restrict__ obs, float* __restrict__ reward_out, uint8_t* __restrict__ done,
                                                          uint32_t* __restrict__ info, int n_steps, const StepTail tl, const ColdParams C, const TeamParams P) {
  constexpr int AD = 7, OD = 29;
  TeamLane L;
  L.init(P);
  const int i = team_group_of_block(int(blockIdx.x), int(gridDim.x)) * 4 + (L.lane >> 4);
  const bool active = i < n_envs;
  const int ia = active ? i : n_envs - 1;
  char* tile = static_cast<char*>(blob) + size_t(i >> 6) * tile_bytes;
  TeamEnv E;
  team_load(tile, i, L, E);
  const size_t n = size_t(n_envs);
  const uint32_t ja = uint32_t(L.cc < 3 ? L.cc : 2);
  const float* ap = actions + size_t(ia) * AD;
  float act = ap[L.cc], actj = ap[4 + ja];
  for (int t = 0; t < n_steps; t++) {
    const float a_now = act, aj_now = actj;
    if (t + 1 < n_steps) { ap += n * AD; act = ap[L.cc]; actj = ap[4 + ja]; }   // next step's action: in flight during this step
    const TeamOut o = team_advance<NROT>(P, C, L, E, a_now, aj_now, i, active, nullptr, nullptr, nullptr);
    accumulate_stats(tl.stats, int(blockIdx.x), o.bits, active && o.ended && L.lead, o.ep_len, o.ep_ret);
    const size_t tn = size_t(t) * n;
    team_store_outputs<true>(L, o, uint32_t(i), active, obs ? obs + tn * OD : nullptr, reward_out ? reward_out + tn : nullptr, done ? done + tn : nullptr,
                       info ? info + tn : nullptr);
  }
  team_store(tile, i, L, E);
}

}  // namespace amenv_dev
