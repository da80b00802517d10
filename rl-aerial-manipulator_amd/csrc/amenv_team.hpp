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
__device__ __forceinline__ float sum_bodies(float v) { v = v + row_ror<4>(v); return v + row_ror<8>(v); }   // over the 4 quads of the row
__device__ __forceinline__ float sum4(float p) { const float t = p + qp<1, 0, 3, 2>(p); return t + qp<2, 3, 0, 1>(t); }   // all 4 lanes valid in, all out
__device__ __forceinline__ float dot3(float a, float b) { const float p = a * b; return (p + rot1(p)) + rot2(p); }        // lanes 0..2
__device__ __forceinline__ float dot_all(float a, float b) { const float p = a * b; return (bc<0>(p) + bc<1>(p)) + bc<2>(p); }   // all 4 lanes

struct X3 { float v, r1, r2; };   // a 3-vector (component per lane) with its two rotations cached
__device__ __forceinline__ X3 x3(float v) { return X3{v, rot1(v), rot2(v)}; }
__device__ __forceinline__ float cross(const X3& a, const X3& b) { return fma_(a.r1, b.r2, -(a.r2 * b.r1)); }
struct TM { float c0, c1, c2; };  // 3x3 matrix: lane i holds row i, one register per column
__device__ __forceinline__ float matvec(const TM& M, float v) { return fma_(M.c0, bc<0>(v), fma_(M.c1, bc<1>(v), M.c2 * bc<2>(v))); }

// ---- parameters -------------------------------------------------------------------------------------------------------------
// Per-lane constants: a [kTeamConsts][16] fp32 table in HBM (built by amenv_create), entry [k][4*body + component]; each lane loads its
// column once at kernel entry (the loads overlap the state loads).
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
  int32_t substeps, max_steps, counter_limit, ee_task;
  uint32_t flags;
  const float* consts;       // [kTeamConsts][16]
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

// 19 derivatives of the arm vehicle in team form.  Same model as arm_rhs_body (amenv_arm.hpp).  F: total thrust (all lanes), Mv: rotor
// moments, cmd: joint commands (component per lane).  c[]: this lane's constants.
__device__ __forceinline__ TeamState team_rhs(const TeamParams& P, const float* c, const TeamState& y, float F, float Mv, float cmd) {
  TeamState d;
  const float e0 = c[TC_E0], e1 = c[TC_E1], e2 = c[TC_E2];
  // attitude: |q|^2, vector part of q in component layout (with rotations), scalar part
  const float n2 = sum4(y.Q * y.Q);
  const float two_in2 = 2.0f * rcp_(n2);
  const X3 qv{qp<1, 2, 3, 3>(y.Q), qp<2, 3, 1, 3>(y.Q), qp<3, 1, 2, 3>(y.Q)};
  const float qw = bc<0>(y.Q);
  const X3 om = x3(y.W);
  // gravity in body components: Rq (0,0,-g) = v + (2/|q|^2) qv x (qv x v + qw v)
  float gb;
  {
    const X3 gv{c[TC_GV], c[TC_GV1], c[TC_GV2]};
    const float t = fma_(qw, gv.v, cross(qv, gv));
    gb = fma_(two_in2, cross(qv, x3(t)), gv.v);
  }
  // joint servos (joint k in lane k)
  const float thdd = clamp_(fma_(P.kp, cmd - y.TH, -(P.kd * y.THD)), -P.amax, P.amax);
  // ---- chain across the joints this body sits behind (masks mk: 1 behind joint k, else 0 -> angle, rate, acceleration, offset vanish)
  TM R;
  float p, pd, pdd, w, al;
  {   // joint 1 about z at the start of the chain: R = 1, p = pd = pdd = w = al = 0
    const float mk = c[TC_MK0];
    const float th = bc<0>(y.TH) * mk, td = bc<0>(y.THD) * mk, tdd = bc<0>(thdd) * mk;
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
    const X3 xRo = x3(Ro), xw = x3(w), xal = x3(al);
    const float wRo = cross(xw, xRo);
    pd = fma_(mk, wRo, pd);
    pdd = fma_(mk, cross(xal, xRo) + cross(xw, x3(wRo)), pdd);
    p = fma_(mk, Ro, p);
    const float z = R.c0;
    const float wz = cross(xw, x3(z));
    al = fma_(td, wz, fma_(tdd, z, al));
    w = fma_(td, z, w);
    rotate_cols<0>(R, s, co);
  };
  advance_x(c[TC_MK1], bc<1>(y.TH) * c[TC_MK1], bc<1>(y.THD) * c[TC_MK1], bc<1>(thdd) * c[TC_MK1], P.o1);
  advance_x(c[TC_MK2], bc<2>(y.TH) * c[TC_MK2], bc<2>(y.THD) * c[TC_MK2], bc<2>(thdd) * c[TC_MK2], P.o2);
  // ---- this body: CoM motion relative to the body frame, inertia in body axes, its Newton-Euler terms
  const float m = c[TC_MASS];
  const float Rc = fma_(R.c0, c[TC_LCX], fma_(R.c1, c[TC_LCY], R.c2 * c[TC_LCZ]));
  const X3 xw = x3(w), xal = x3(al), xRc = x3(Rc);
  const float wRc = cross(xw, xRc);
  const float r = p + Rc, u = pd + wRc;
  const float a_ = pdd + cross(xal, xRc) + cross(xw, x3(wRc));
  TM J;   // J = R I R^T
  {
    const float RI0 = fma_(R.c0, c[TC_I00], fma_(R.c1, c[TC_I01], R.c2 * c[TC_I02]));
    const float RI1 = fma_(R.c0, c[TC_I01], fma_(R.c1, c[TC_I11], R.c2 * c[TC_I12]));
    const float RI2 = fma_(R.c0, c[TC_I02], fma_(R.c1, c[TC_I12], R.c2 * c[TC_I22]));
    J.c0 = fma_(RI0, bc<0>(R.c0), fma_(RI1, bc<0>(R.c1), RI2 * bc<0>(R.c2)));
    J.c1 = fma_(RI0, bc<1>(R.c0), fma_(RI1, bc<1>(R.c1), RI2 * bc<1>(R.c2)));
    J.c2 = fma_(RI0, bc<2>(R.c0), fma_(RI1, bc<2>(R.c1), RI2 * bc<2>(R.c2)));
  }
  const X3 xr = x3(r);
  const float b_ = cross(om, x3(cross(om, xr))) + 2.0f * cross(om, x3(u)) + a_;     // w x (w x r) + 2 w x u + a
  const float aa = al + cross(om, xw), Om = y.W + w;
  float S = m * r, fb = m * b_;
  float nb = m * cross(xr, x3(b_)) + matvec(J, aa) + cross(x3(Om), x3(matvec(J, Om)));
  const float r2 = dot3(r, r);
  const float mr2 = m * r2;
  float IO0 = fma_(-m, r * bc<0>(r), fma_(mr2, e0, J.c0));       // I_O += J + m (|r|^2 1 - r r^T), column by column
  float IO1 = fma_(-m, r * bc<1>(r), fma_(mr2, e1, J.c1));
  float IO2 = fma_(-m, r * bc<2>(r), fma_(mr2, e2, J.c2));
  // ---- sums over the four bodies of the row
  S = sum_bodies(S); fb = sum_bodies(fb); nb = sum_bodies(nb);
  IO0 = sum_bodies(IO0); IO1 = sum_bodies(IO1); IO2 = sum_bodies(IO2);
  // ---- external wrench about O, composite inertia about the system CoM, 3x3 solve by the adjugate (columns = cross products)
  const float f = fma_(F, e2, fma_(P.mtot, gb, -fb));
  const X3 xS = x3(S);
  const float n = Mv + cross(xS, x3(gb)) - nb;
  const float im = P.inv_mtot;
  const float imS2 = im * dot3(S, S);
  const float imS = im * S;
  const float a0 = fma_(imS, bc<0>(S), fma_(-imS2, e0, IO0));    // I_c = I_O - (|S|^2 1 - S S^T) / mtot
  const float a1 = fma_(imS, bc<1>(S), fma_(-imS2, e1, IO1));
  const float a2 = fma_(imS, bc<2>(S), fma_(-imS2, e2, IO2));
  const float rhs = n - im * cross(xS, x3(f));
  const X3 x0 = x3(a0), x1 = x3(a1), x2 = x3(a2);
  const TM C{cross(x1, x2), cross(x2, x0), cross(x0, x1)};
  const float idet = rcp_(dot3(a0, C.c0));
  const float wd = idet * matvec(C, rhs);
  const float Aacc = im * (f + cross(xS, x3(wd)));
  // world acceleration of O: Rq^T A = A + (2/|q|^2) qv x (qv x A - qw A)
  const float t = fma_(-qw, Aacc, cross(qv, x3(Aacc)));
  const float vd = fma_(two_in2, cross(qv, x3(t)), Aacc);
  // quaternion kinematics (4 lanes): -1/2 Omega(w) q + 2 (1 - |q|^2) q
  const float kq = fma_(-2.0f, n2, 2.0f);
  float dq = kq * y.Q;
  dq = fma_(c[TC_SP] * bc<0>(y.W), qp<1, 0, 3, 2>(y.Q), dq);
  dq = fma_(c[TC_SQ] * bc<1>(y.W), qp<2, 3, 0, 1>(y.Q), dq);
  dq = fma_(c[TC_SR] * bc<2>(y.W), qp<3, 2, 1, 0>(y.Q), dq);
  d.P = y.V; d.V = vd; d.Q = dq; d.W = wd; d.TH = y.THD; d.THD = thdd;
  return d;
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
  const float t = fma_(-qw, p, cross(qv, x3(p)));
  return fma_(2.0f, cross(qv, x3(t)), p);
}

// One control step of 4 envs per wavefront.  grid = ceil(n_tiles * 64 / 4) workgroups of 64 threads.
template <int NROT>
__global__ __launch_bounds__(64) void step_kernel_team(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, const float* __restrict__ actions,
                                                       float* __restrict__ obs, float* __restrict__ reward_out, uint8_t* __restrict__ done,
                                                       uint32_t* __restrict__ info, const StepTail tl, const HotParams<float, NROT> HP, const ColdParams C,
                                                       const TeamParams P) {
  static_assert(NROT == 6, "team kernel: 6-rotor airframe");
  constexpr int OD = 29, AD = 7;
  const int lane = int(threadIdx.x);
  const int cc = lane & 3, bb = (lane >> 2) & 3;
  const int i = int(blockIdx.x) * 4 + (lane >> 4);          // env of this row
  const bool active = i < n_envs;
  const int ia = active ? i : n_envs - 1;                   // rows past the end redo the last env's arithmetic (their stores are masked)
  char* tile = static_cast<char*>(blob) + size_t(i >> 6) * tile_bytes;
  const uint32_t eoff = uint32_t(i & 63) * 16u + uint32_t(cc) * 4u;
  auto gload = [&](int g) { return *reinterpret_cast<const float*>(tile + kIntBytes + uint32_t(g) * 1024u + eoff); };
  // per-lane constants
  float c[kTeamConsts];
#pragma unroll
  for (int k = 0; k < kTeamConsts; k++) c[k] = P.consts[k * 16 + (lane & 15)];
  // state (every quad of the row holds a copy), per-env scalars ride in lane 3 of the p / v / w groups
  TeamState y{gload(0), gload(1), gload(2), gload(3), gload(5), gload(6)};
  const float WPv = gload(4);
  const int4 iv = *(reinterpret_cast<const int4*>(tile) + (i & 63));
  const float act = actions[size_t(ia) * AD + cc];                                     // a0..a3: one per lane
  const float actj = actions[size_t(ia) * AD + 4 + (cc < 3 ? cc : 2)];                  // joint commands a4..a6
  const float final_yaw = bc<3>(y.P), last_distance = bc<3>(y.V), ep_return0 = bc<3>(y.W);
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
  // RK4 (running weighted sum)
  const float h = P.h, hh = 0.5f * h, h6 = h * (1.0f / 6.0f);
  int it = 0;
  do {
    TeamState k = team_rhs(P, c, y, F, Mv, cmd), acc = k, s;
#define AMENV_TS(OP) OP(P) OP(V) OP(Q) OP(W) OP(TH) OP(THD)
#define ST1(f) s.f = fma_(hh, k.f, y.f);
    AMENV_TS(ST1)
    k = team_rhs(P, c, s, F, Mv, cmd);
#define ST2(f) acc.f = fma_(2.0f, k.f, acc.f); s.f = fma_(hh, k.f, y.f);
    AMENV_TS(ST2)
    k = team_rhs(P, c, s, F, Mv, cmd);
#define ST3(f) acc.f = fma_(2.0f, k.f, acc.f); s.f = fma_(h, k.f, y.f);
    AMENV_TS(ST3)
    k = team_rhs(P, c, s, F, Mv, cmd);
#define ST4(f) y.f = fma_(h6, acc.f + k.f, y.f);
    AMENV_TS(ST4)
#undef ST1
#undef ST2
#undef ST3
#undef ST4
  } while (++it < P.substeps);
  y.Q = y.Q * rsqrt_(sum4(y.Q * y.Q));
  const float EO = team_tool_offset(P, c, y);
  // ---- task step: the one-lane kernels' code on broadcast copies of the state (identical in the 16 lanes of a row)
  Env<float, 1> e;
  e.px = bc<0>(y.P); e.py = bc<1>(y.P); e.pz = bc<2>(y.P);
  e.vx = bc<0>(y.V); e.vy = bc<1>(y.V); e.vz = bc<2>(y.V);
  e.qw = bc<0>(y.Q); e.qx = bc<1>(y.Q); e.qy = bc<2>(y.Q); e.qz = bc<3>(y.Q);
  e.wx = bc<0>(y.W); e.wy = bc<1>(y.W); e.wz = bc<2>(y.W);
  e.wp[0][0] = bc<0>(WPv); e.wp[0][1] = bc<1>(WPv); e.wp[0][2] = bc<2>(WPv);
  e.eox = bc<0>(EO); e.eoy = bc<1>(EO); e.eoz = bc<2>(EO);
#pragma unroll
  for (int k = 0; k < 3; k++) { e.th[k] = 0.0f; e.thd[k] = 0.0f; }    // (joints stay in team registers; the task code does not read them)
  e.final_yaw = final_yaw; e.last_distance = last_distance; e.ep_return = ep_return0;
  e.step = iv.x; e.counter = iv.y; e.flags = iv.z; e.episode = iv.w;
  float reward;
  uint32_t bits = task_step<float, 1, true>(HP, e, reward);
  e.ep_return += reward;
  // observation segments of this lane: A = [p/10 | v/5 | q | w/5] by quad, B = [(wp - task point)/2 | 0 | yaw/pi | th/pi], C = [thd/5 | tool offset*2]
  const bool q0 = bb == 0, q1 = bb == 1, q2 = bb == 2;
  auto obs_vals = [&](const TeamState& z, float wpv, float eo, float fyaw, float& vA, float& vB, float& vC) {
    vA = (q0 ? z.P : (q1 ? z.V : (q2 ? z.Q : z.W))) * c[TC_OBS_A];
    const float tp = HP.ee_task != 0 ? z.P + eo : z.P;
    vB = (q0 ? wpv - tp : (q1 ? 0.0f : (q2 ? fyaw : z.TH))) * c[TC_OBS_B];
    vC = (q0 ? z.THD : eo) * c[TC_OBS_C];
  };
  const int offA = q0 ? cc : (q1 ? 3 + cc : (q2 ? 6 + cc : 10 + cc));
  const int offB = q0 ? 13 + cc : (q1 ? 16 + cc : (q2 ? 19 : 20 + cc));
  const int offC = q0 ? 23 + cc : 26 + cc;
  const bool okA = cc < 3 || q2, okB = q2 ? cc == 0 : cc < 3, okC = cc < 3 && bb < 2;
  float vA, vB, vC;
  obs_vals(y, WPv, EO, final_yaw, vA, vB, vC);
  const bool ended = (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
  const bool resets = ended && (HP.flags & AMENV_FLAG_AUTO_RESET);
  const bool lead = (lane & 15) == 0;
  const bool is_done = active && ended;
  int ep_len = 0; float ep_ret = 0.0f;
  float WPn = WPv; float fy_n = final_yaw;
  if (ended) {   // uniform within the row; SB3 DummyVecEnv + Monitor contract
    ep_len = e.step; ep_ret = e.ep_return;
    if (active) {
      if (tl.terminal_obs) {
        float* t = tl.terminal_obs + size_t(i) * OD;
        if (okA) t[offA] = vA;
        if (okB) t[offB] = vB;
        if (okC) t[offC] = vC;
      }
      if (lead) {
        if (tl.ep_return) tl.ep_return[i] = ep_ret;
        if (tl.ep_len) tl.ep_len[i] = ep_len;
      }
    }
    if (resets) {
      // reset RNG: lane c < 3 of every quad computes Philox block c; the 12 words are then broadcast inside the quad
      uint32_t wds[4];
      const int64_t gid = C.gid0 + i;
      philox4x32_10(C.seed_lo, C.seed_hi, uint32_t(uint64_t(gid)), uint32_t(uint64_t(gid) >> 32), uint32_t(e.episode), uint32_t(cc), wds);
      uint32_t r[12];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        r[k] = uint32_t(qpi<0, 0, 0, 0>(int(wds[k]))); r[4 + k] = uint32_t(qpi<1, 1, 1, 1>(int(wds[k]))); r[8 + k] = uint32_t(qpi<2, 2, 2, 2>(int(wds[k])));
      }
      reset_from_words<float, 1>(C, 1, e, r);
      e.eox = P.ee_home[0]; e.eoy = P.ee_home[1]; e.eoz = P.ee_home[2];
      const float e0 = c[TC_E0], e1 = c[TC_E1], e2 = c[TC_E2];
      y.P = fma_(e0, e.px, fma_(e1, e.py, e2 * e.pz));
      y.V = 0.0f; y.W = 0.0f; y.TH = 0.0f; y.THD = 0.0f;
      y.Q = cc == 0 ? 1.0f : 0.0f;
      WPn = fma_(e0, e.wp[0][0], fma_(e1, e.wp[0][1], e2 * e.wp[0][2]));
      fy_n = e.final_yaw;
      const float EOn = fma_(e0, e.eox, fma_(e1, e.eoy, e2 * e.eoz));
      obs_vals(y, WPn, EOn, fy_n, vA, vB, vC);
      bits |= AMENV_INFO_WAS_RESET;
    }
  }
  accumulate_stats(tl.stats, int(blockIdx.x), bits, is_done && lead, ep_len, ep_ret);
  // ---- stores.  State: quad b writes group b (p|yaw, v|last_distance, q, w|return), then joints (quads 0, 1) and the int plane (quad 2)
  {
    const bool l3 = cc == 3;
    const float g0 = l3 ? fy_n : y.P, g1 = l3 ? e.last_distance : y.V, g3 = l3 ? e.ep_return : y.W;
    const float sv = q0 ? g0 : (q1 ? g1 : (q2 ? y.Q : g3));
    *reinterpret_cast<float*>(tile + kIntBytes + uint32_t(bb) * 1024u + eoff) = sv;
    if (bb < 2) *reinterpret_cast<float*>(tile + kIntBytes + uint32_t(5 + bb) * 1024u + eoff) = q0 ? y.TH : y.THD;
    if (q2) {
      const int ival = cc == 0 ? e.step : (cc == 1 ? e.counter : (cc == 2 ? e.flags : e.episode));
      *(reinterpret_cast<int*>(tile) + (i & 63) * 4 + cc) = ival;
    }
    if (bb == 3 && (bits & AMENV_INFO_WAS_RESET)) *reinterpret_cast<float*>(tile + kIntBytes + 4u * 1024u + eoff) = WPn;   // per-episode: waypoint
  }
  if (active) {
    float* o = obs + size_t(i) * OD;
    if (okA) o[offA] = vA;
    if (okB) o[offB] = vB;
    if (okC) o[offC] = vC;
    if (lead) {
      reward_out[i] = reward;
      done[i] = is_done ? 1 : 0;
      info[i] = bits;
    }
  }
}

}  // namespace amenv_dev
