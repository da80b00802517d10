// amenv_model.hpp -- per-environment device code for gfx950 (CDNA4): everything one lane does
// for one environment in one control step.  One lane = one environment; all of an
// environment's episode state lives in that lane's registers for the whole step.
//
// Semantics follow the reference (paths relative to the reference root,
// v2 = initial-implementation-v2):
//   mixer / clamp / re-mix        v2/simul_files/model/quadcopter.py:105-112
//   ODE right-hand side           v2/simul_files/model/quadcopter.py:66-103
//   R(q/|q|) third column         v2/simul_files/utils/quaternion.py:46-77 (closed form)
//   integrate + renormalise       v2/simul_files/model/quadcopter.py:113-114 (RK4 for odeint)
//   reward                        v2/rl_env_scaledObs.py:198-231
//   step state machine            v2/rl_env_scaledObs.py:135-196
//   observation                   v2/rl_env_scaledObs.py:98-121
//   quaternion -> roll/pitch/yaw  v2/utils2/utils.py:4-9
//   reset + waypoint generators   v2/rl_env_scaledObs.py:40-79, v2/utils2/utils.py:12-95
//
// Templated on the arithmetic type T: float is the product path, double is the logic-check
// build of the SAME code (tests compare it with the fp64 CPU oracle at ~1e-12).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/amenv.h"

namespace amenv_dev {

constexpr int kActDim = 4;
enum TaskVar { VAR_V2 = 0, VAR_V1 = 1 };                 // v2/rl_env_scaledObs.py | v1/rl_env_scaledObs.py + v1/rl_env.py
// v2 + arm: + joint angle, rate per joint + (tool point - base position) / 0.5 in world axes (forward kinematics)
template <int VAR, int NJ = 0> struct ObsDim { static constexpr int value = VAR == VAR_V1 ? 17 : 20 + 2 * NJ + (NJ > 0 ? 3 : 0); };
constexpr int kObsDimMax = 29;

// ---- kernel-argument constants (uniform: they live in SGPRs) ----------------------------------
// Hot parameters are kept COMPACT (rotor-count-sized mixer, symmetric inertia) so that one batch of
// scalar loads at kernel entry brings all of them into the ~100 available SGPRs: with 64 lone
// wavefronts per launch every extra dependent scalar-load phase is ~0.3 us of pure latency.
template <typename T, int NR>
struct HotParams {
  T alloc[NR][4];   // rotor thrusts = alloc . [F,Mx,My,Mz]            (quadcopter.py:109)
  T mixm[3][NR];    // [Mx,My,Mz] = mixm . thrusts; F = sum(thrusts)    (quadcopter.py:111-112)
  T tmin[NR], tmax[NR];
  T Ixx, Ixy, Ixz, Iyy, Iyz, Izz;   // inertia (symmetric)
  T Jxx, Jxy, Jxz, Jyy, Jyz, Jzz;   // inverse inertia (symmetric)
  T inv_mass, g, h;                 // h = dt / substeps
  float mass_f, g_f, mscale_f;      // action scaling is fp32 in the reference (NumPy >= 2 promotion)
  int32_t n_rotors;                 // used only by the generic (NR = AMENV_MAX_ROTORS) instantiation
  int32_t substeps, max_steps, counter_limit;
  uint32_t flags;
  int32_t K;                        // waypoints per episode (<= KW of the instantiation); v1: storage bound
  int32_t raw_obs;                  // v1 only: 1 = v1/rl_env.py (unscaled observation)
  int32_t ee_task;                  // arm vehicles: 1 = the waypoint task measures from the tool point (AMENV_EE_TASK_TOOL)
};

// Parameters only the reset path needs (cold: loaded when a lane actually resets).
struct ColdParams {
  float traj_sin[AMENV_MAX_WAYPOINTS], traj_cos[AMENV_MAX_WAYPOINTS];
  uint32_t seed_lo, seed_hi;
  int64_t gid0;                     // global id of local env 0
};

// ---- math helpers: one definition per arithmetic type ------------------------------------
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }   // 1 ulp, 1 instr
__device__ __forceinline__ double rcp_(double x) { return 1.0 / x; }
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32, 1 ulp, 1 instr
__device__ __forceinline__ double sqrt_(double x) { return sqrt(x); }
__device__ __forceinline__ float rsqrt_(float x) {                                     // v_rsq + one Newton step
  float y = __builtin_amdgcn_rsqf(x);
  return y * __builtin_fmaf(-0.5f * x, y * y, 1.5f);
}
__device__ __forceinline__ double rsqrt_(double x) { return 1.0 / sqrt(x); }
// fp32 atan2 / asin for the hold-phase bonuses (rl_env_scaledObs.py:163,169-171).  The device libm versions cost
// ~430 cycles each for a lone wavefront and the hold phase is where a trained policy spends ~500 steps per episode,
// so they are replaced by a compact form: octant reduction + the cephes atanf minimax polynomial on |t| <= tan(pi/8)
// (abs error <= 3e-7 rad incl. the two v_rcp_f32; the bonuses scale angles by <= 50, far inside the 1e-5 gate).
__device__ __forceinline__ float atan_reduced_(float t) {  // |t| <= 0.41421356
  const float z = t * t;
  float p = __builtin_fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = __builtin_fmaf(p, z, 1.99777106478e-1f);
  p = __builtin_fmaf(p, z, -3.33329491539e-1f);
  return __builtin_fmaf(p * z, t, t);
}
__device__ __forceinline__ float atan2_(float y, float x) {
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const bool swap = ay > ax;
  const float mx = swap ? ay : ax, mn = swap ? ax : ay;
  const float t = mn * __builtin_amdgcn_rcpf(mx);                       // [0, 1]
  const bool big = t > 0.41421356237f;
  const float tr = big ? (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f) : t;
  float a = atan_reduced_(tr);
  a = big ? a + 0.78539816339744831f : a;
  a = swap ? 1.57079632679489662f - a : a;
  a = x < 0.0f ? 3.14159265358979324f - a : a;
  a = mx == 0.0f ? 0.0f : a;
  return __builtin_copysignf(a, y);
}
__device__ __forceinline__ double atan2_(double a, double b) { return atan2(a, b); }
__device__ __forceinline__ float asin_(float a) { return atan2_(a, __builtin_amdgcn_sqrtf((1.0f - a) * (1.0f + a))); }
__device__ __forceinline__ double asin_(double a) { return asin(a); }
__device__ __forceinline__ float clamp_(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }  // v_med3_f32
__device__ __forceinline__ double clamp_(double x, double lo, double hi) { return __builtin_fmax(__builtin_fmin(x, hi), lo); }
__device__ __forceinline__ float abs_(float a) { return fabsf(a); }
__device__ __forceinline__ double abs_(double a) { return fabs(a); }
__device__ __forceinline__ bool finite_(float a) { return __builtin_isfinite(a); }
__device__ __forceinline__ bool finite_(double a) { return __builtin_isfinite(a); }

// ---- per-env registers ------------------------------------------------------------------------
template <typename T, int KW>
struct Env {
  T px, py, pz, vx, vy, vz, qw, qx, qy, qz, wx, wy, wz;
  T wp[KW][3];
  T final_yaw, last_distance, ep_return;
  T th[AMENV_MAX_JOINTS], thd[AMENV_MAX_JOINTS];   // arm joint angles / rates (unused, and eliminated, without an arm)
  T eox, eoy, eoz;                                 // arm: tool point - base position, world axes (forward kinematics of the post-step
                                                   // state; derived, not stored)
  int32_t step, counter, flags, episode;
};

template <typename T>
struct Deriv { T vx, vy, vz, ax, ay, az, dqw, dqx, dqy, dqz, dwx, dwy, dwz; };

// All floating-point contraction is explicit: the library is compiled with -ffp-contract=off and every
// fused multiply-add below is written as fma_().  The step kernel, the rollout kernel and any future
// variant therefore produce bit-identical trajectories (tests/test_gpu_parity.py::test_rollout_*).
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> __device__ __forceinline__ T dot3_(T a0, T a1, T a2, T b0, T b1, T b2) { return fma_(a0, b0, fma_(a1, b1, a2 * b2)); }

// ODE right-hand side (quadcopter.py:66-103).  F, M: post-mixer wrench, constant over the step.
template <typename T, typename PT>
__device__ __forceinline__ Deriv<T> rhs(const PT& P, T vx, T vy, T vz, T qw, T qx, T qy, T qz, T p, T q, T r, T Fm,
                                        T Mx, T My, T Mz) {
  Deriv<T> d;
  // third column of wRb for the NORMALISED quaternion: quadratic in q/|q| => divide by |q|^2 once
  const T n2 = fma_(qw, qw, fma_(qx, qx, fma_(qy, qy, qz * qz)));
  const T in2 = rcp_(n2);
  const T s = (Fm + Fm) * in2;                                   // 2 F / (m |q|^2)
  d.vx = vx; d.vy = vy; d.vz = vz;                               // :89-91
  d.ax = s * fma_(qx, qz, -(qw * qy));                           // :73-74
  d.ay = s * fma_(qy, qz, qw * qx);
  d.az = fma_(-s, fma_(qx, qx, qy * qy), Fm) - P.g;
  // qdot = -1/2 Omega(p,q,r) q + 2 (1-|q|^2) q                  :77-82
  const T k = fma_(T(-2), n2, T(2));
  d.dqw = fma_(T(0.5), fma_(p, qx, fma_(q, qy, r * qz)), k * qw);
  d.dqx = fma_(T(-0.5), fma_(p, qw, fma_(q, qz, -(r * qy))), k * qx);
  d.dqy = fma_(T(-0.5), fma_(q, qw, fma_(r, qx, -(p * qz))), k * qy);
  d.dqz = fma_(T(-0.5), fma_(r, qw, fma_(p, qy, -(q * qx))), k * qz);
  // pqrdot = invI (M - w x (I w))                               :86-87
  const T i0 = dot3_(P.Ixx, P.Ixy, P.Ixz, p, q, r);
  const T i1 = dot3_(P.Ixy, P.Iyy, P.Iyz, p, q, r);
  const T i2 = dot3_(P.Ixz, P.Iyz, P.Izz, p, q, r);
  const T t0 = Mx - fma_(q, i2, -(r * i1));
  const T t1 = My - fma_(r, i0, -(p * i2));
  const T t2 = Mz - fma_(p, i1, -(q * i0));
  d.dwx = dot3_(P.Jxx, P.Jxy, P.Jxz, t0, t1, t2);
  d.dwy = dot3_(P.Jxy, P.Jyy, P.Jyz, t0, t1, t2);
  d.dwz = dot3_(P.Jxz, P.Jyz, P.Jzz, t0, t1, t2);
  return d;
}

// Quadcopter.update (quadcopter.py:105-114) with RK4 in place of odeint.
template <typename T, int NROT, int KW>
__device__ __forceinline__ void dynamics(const HotParams<T, NROT>& P, Env<T, KW>& e, float a0, float a1, float a2, float a3) {
  // action scaling in fp32, left to right (rl_env_scaledObs.py:125-126)
  const float Ff = (a0 * P.mass_f) * P.g_f;
  const T u0 = T(Ff), u1 = T(a1 * P.mscale_f), u2 = T(a2 * P.mscale_f), u3 = T(a3 * P.mscale_f);
  // mixer -> per-rotor clamp -> re-mix (:109-112)
  T F = T(0), Mx = T(0), My = T(0), Mz = T(0);
#pragma unroll
  for (int r = 0; r < NROT; r++) {
    if (NROT == AMENV_MAX_ROTORS && r >= P.n_rotors) break;  // generic instantiation: runtime rotor count
    T t = fma_(P.alloc[r][0], u0, fma_(P.alloc[r][1], u1, fma_(P.alloc[r][2], u2, P.alloc[r][3] * u3)));
    t = clamp_(t, P.tmin[r], P.tmax[r]);   // np.maximum(np.minimum(t, max), min), quadcopter.py:110
    F = F + t; Mx = fma_(P.mixm[0][r], t, Mx); My = fma_(P.mixm[1][r], t, My); Mz = fma_(P.mixm[2][r], t, Mz);
  }
  const T Fm = F * P.inv_mass;
  const T h = P.h, hh = T(0.5) * h, h6 = h * T(1.0 / 6.0);
  int it = 0;
  do {  // substeps >= 1 (validated at create): no loop guard in front of the first stage
    const Deriv<T> k1 = rhs<T>(P, e.vx, e.vy, e.vz, e.qw, e.qx, e.qy, e.qz, e.wx, e.wy, e.wz, Fm, Mx, My, Mz);
    const Deriv<T> k2 = rhs<T>(P, fma_(hh, k1.ax, e.vx), fma_(hh, k1.ay, e.vy), fma_(hh, k1.az, e.vz), fma_(hh, k1.dqw, e.qw),
                            fma_(hh, k1.dqx, e.qx), fma_(hh, k1.dqy, e.qy), fma_(hh, k1.dqz, e.qz), fma_(hh, k1.dwx, e.wx),
                            fma_(hh, k1.dwy, e.wy), fma_(hh, k1.dwz, e.wz), Fm, Mx, My, Mz);
    const Deriv<T> k3 = rhs<T>(P, fma_(hh, k2.ax, e.vx), fma_(hh, k2.ay, e.vy), fma_(hh, k2.az, e.vz), fma_(hh, k2.dqw, e.qw),
                            fma_(hh, k2.dqx, e.qx), fma_(hh, k2.dqy, e.qy), fma_(hh, k2.dqz, e.qz), fma_(hh, k2.dwx, e.wx),
                            fma_(hh, k2.dwy, e.wy), fma_(hh, k2.dwz, e.wz), Fm, Mx, My, Mz);
    const Deriv<T> k4 = rhs<T>(P, fma_(h, k3.ax, e.vx), fma_(h, k3.ay, e.vy), fma_(h, k3.az, e.vz), fma_(h, k3.dqw, e.qw),
                            fma_(h, k3.dqx, e.qx), fma_(h, k3.dqy, e.qy), fma_(h, k3.dqz, e.qz), fma_(h, k3.dwx, e.wx),
                            fma_(h, k3.dwy, e.wy), fma_(h, k3.dwz, e.wz), Fm, Mx, My, Mz);
    // y += h/6 (k1 + 2 k2 + 2 k3 + k4)
#define AMENV_RK4(x, f) e.x = fma_(h6, fma_(T(2), k2.f + k3.f, k1.f + k4.f), e.x)
    AMENV_RK4(px, vx); AMENV_RK4(py, vy); AMENV_RK4(pz, vz);
    AMENV_RK4(vx, ax); AMENV_RK4(vy, ay); AMENV_RK4(vz, az);
    AMENV_RK4(qw, dqw); AMENV_RK4(qx, dqx); AMENV_RK4(qy, dqy); AMENV_RK4(qz, dqz);
    AMENV_RK4(wx, dwx); AMENV_RK4(wy, dwy); AMENV_RK4(wz, dwz);
#undef AMENV_RK4
  } while (++it < P.substeps);
  const T rn = rsqrt_(fma_(e.qw, e.qw, fma_(e.qx, e.qx, fma_(e.qy, e.qy, e.qz * e.qz))));   // :114
  e.qw *= rn; e.qx *= rn; e.qy *= rn; e.qz *= rn;
}

template <typename T, int KW>
__device__ __forceinline__ void current_waypoint(int K, const Env<T, KW>& e, int idx, T& cx, T& cy, T& cz) {
  cx = e.wp[0][0]; cy = e.wp[0][1]; cz = e.wp[0][2];
#pragma unroll
  for (int k = 1; k < KW; k++)
    if (k < K && idx >= k) { cx = e.wp[k][0]; cy = e.wp[k][1]; cz = e.wp[k][2]; }
}

// The point the waypoint task measures from: the base position (the reference), or with EE the arm's tool point.
template <bool EE, typename T, int KW>
__device__ __forceinline__ void task_point(const Env<T, KW>& e, bool ee_task, T& tx, T& ty, T& tz) {
  tx = e.px; ty = e.py; tz = e.pz;
  if constexpr (EE) {
    if (ee_task) { tx = e.px + e.eox; ty = e.py + e.eoy; tz = e.pz + e.eoz; }
  }
}

// _get_observation (rl_env_scaledObs.py:98-121)
template <typename T, int KW, bool EE = false>
__device__ __forceinline__ void observe(int K, const Env<T, KW>& e, float* o, bool ee_task = false) {
  const int idx = e.flags & 15;
  T cx, cy, cz;
  current_waypoint(K, e, idx, cx, cy, cz);
  T tx, ty, tz;
  task_point<EE>(e, ee_task, tx, ty, tz);
  // scalings as multiplications by the rounded reciprocal (<= 1 ulp from the reference's divisions)
  const T c10 = T(0.1), c5 = T(0.2), c2 = T(0.5);
  o[0] = float(e.px * c10); o[1] = float(e.py * c10); o[2] = float(e.pz * c10);
  o[3] = float(e.vx * c5); o[4] = float(e.vy * c5); o[5] = float(e.vz * c5);
  o[6] = float(e.qw); o[7] = float(e.qx); o[8] = float(e.qy); o[9] = float(e.qz);
  o[10] = float(e.wx * c5); o[11] = float(e.wy * c5); o[12] = float(e.wz * c5);
  o[13] = float((cx - tx) * c2); o[14] = float((cy - ty) * c2); o[15] = float((cz - tz) * c2);
  T nx = T(0), ny = T(0), nz = T(0);
#pragma unroll
  for (int k = 1; k < KW; k++)
    if (k < K && idx == k - 1) { nx = e.wp[k][0] - cx; ny = e.wp[k][1] - cy; nz = e.wp[k][2] - cz; }
  o[16] = float(nx * c2); o[17] = float(ny * c2); o[18] = float(nz * c2);
  o[19] = float(e.final_yaw * T(0.31830988618379067154));
}

// One WaypointQuadEnv.step (rl_env_scaledObs.py:123-196) after the dynamics update.
// Returns info bits; reward in `reward`.  Mutates the episode registers.
template <typename T, int KW, bool EE = false, typename PT>
__device__ __forceinline__ uint32_t task_step(const PT& P, Env<T, KW>& e, T& reward) {
  const int K = KW == 1 ? 1 : P.K;
  uint32_t bits = 0;
  int idx = e.flags & 15;
  bool fwr = (e.flags & AMENV_FLAGBIT_FWR) != 0;
  bool cact = (e.flags & AMENV_FLAGBIT_COUNTER_ACTIVE) != 0;
  const bool truncated = e.step >= P.max_steps;                         // :144
  e.step += 1;                                                          // :145
  if (truncated) bits |= AMENV_INFO_TRUNCATED;

  if (P.flags & AMENV_FLAG_NAN_GUARD) {                                 // deviation, see DESIGN.md
    const T sum = e.px + e.py + e.pz + e.vx + e.vy + e.vz + e.qw + e.qx + e.qy + e.qz + e.wx + e.wy + e.wz;
    if (!finite_(sum)) { reward = T(-100); return bits | AMENV_INFO_TERMINATED | AMENV_INFO_NONFINITE; }
  }
  T cx, cy, cz;
  current_waypoint(K, e, idx, cx, cy, cz);
  // ---- _calculate_reward (:198-231)
  T tx, ty, tz;
  task_point<EE>(e, P.ee_task != 0, tx, ty, tz);
  const T dx = tx - cx, dy = ty - cy, dz = tz - cz;
  const T dist = sqrt_(dot3_(dx, dy, dz, dx, dy, dz));                    // :204
  T r_dist = T(-10) * dist;                                             // :207
  const T v2 = dot3_(e.vx, e.vy, e.vz, e.vx, e.vy, e.vz);
  const T w2 = dot3_(e.wx, e.wy, e.wz, e.wx, e.wy, e.wz);
  const T vn = sqrt_(v2), wn = sqrt_(w2);
  T r_speed = T(-0.1) * v2;                                             // :208
  if (wn > T(0.1)) r_speed -= T(0.01) * w2;                             // :209-210
  T r_time = T(-0.1);                                                   // :211
  T r_prog = T(0);
  if (e.last_distance >= T(0)) {                                        // :214-220
    r_prog = T(20) * (e.last_distance - dist);
    if (r_prog > T(0)) r_prog += T(2);
  }
  e.last_distance = dist;                                               // :222
  if (fwr) {                                                            // :224-228
    r_prog = T(0); r_time = T(0);
    if (dist < T(0.1)) r_dist = T(1);
  }
  reward = ((r_dist + r_speed) + r_time) + r_prog;                      // :231

  bool returned = false;
  if (dist < T(0.1)) {                                                  // :147
    if (!fwr) { idx += 1; reward += T(100); }                           // :148-150
    if (idx >= K) {                                                     // :153 (else: next waypoint, fall through)
      idx = K;                                                          // waypoint_index == len(list)
      // roll, pitch, yaw (utils2/utils.py:4-9 closed form); only the hold-phase bonuses use them
      const T roll = atan2_(T(2) * fma_(e.qw, e.qx, e.qy * e.qz), fma_(T(-2), fma_(e.qx, e.qx, e.qy * e.qy), T(1)));
      T sp = T(2) * fma_(e.qw, e.qy, -(e.qz * e.qx));
      sp = sp > T(1) ? T(1) : (sp < T(-1) ? T(-1) : sp);
      const T pitch = asin_(sp);
      const T yaw = atan2_(T(2) * fma_(e.qw, e.qz, e.qx * e.qy), fma_(T(-2), fma_(e.qy, e.qy, e.qz * e.qz), T(1)));
      // bonuses: divisions by constants as multiplications by the rounded reciprocal (<= 1 ulp), selects not branches
      const T two_pi = T(6.28318530717958647692), inv_two_pi = T(0.15915494309189533577);
      const T dyaw = abs_(yaw - e.final_yaw);
      const T yaw_f = dyaw < two_pi ? T(1) - dyaw * inv_two_pi : T(0);     // (1 - |dyaw|/2pi) or 0   :163,169
      bits |= AMENV_INFO_SUCCESS;
      if (vn < T(0.1) && wn < T(0.1)) bits |= AMENV_INFO_STOPPED;
      const T ar = abs_(roll), ap = abs_(pitch);
      const T roll_b = ar < T(0.2) ? T(10) * (T(1) - ar * T(5)) : T(-0.1) * ar;      // :170,176
      const T pitch_b = ap < T(0.2) ? T(10) * (T(1) - ap * T(5)) : T(-0.1) * ap;     // :171,177
      const T stop_b = vn < T(1) ? T(150) * (T(1) - v2) : T(0);                      // :162
      const T r_first = ((reward + T(200)) + stop_b) + T(100) * yaw_f;               // :164 first arrival
      const T r_hold = ((reward + T(30) * yaw_f) + roll_b) + pitch_b;                // :173,179 holding
      reward = fwr ? r_hold : r_first;
      if (fwr) {                                                        // :165-179 holding
        if (e.counter <= P.counter_limit) e.counter += 1;               // :166-167
        else bits |= AMENV_INFO_TERMINATED;                             // :174-179
      }
      cact = true; fwr = true;                                          // :157-158 (no-ops when already holding)
      returned = true;
    }
  }
  if (!returned) {
    if (cact) e.counter += 1;                                           // :181-182
    if (e.pz < T(0.1)) {                                                // :188-192
      reward -= T(100);
      if (e.vz < T(0)) reward += e.vz * T(100);
      bits |= AMENV_INFO_TERMINATED | AMENV_INFO_CRASHED;
    } else if (sqrt_(dot3_(e.px, e.py, e.pz, e.px, e.py, e.pz)) > T(10)) { // :193-195
      reward -= T(100);
      bits |= AMENV_INFO_TERMINATED | AMENV_INFO_OOB;
    }
  }
  e.flags = (idx & 15) | (fwr ? AMENV_FLAGBIT_FWR : 0) | (cact ? AMENV_FLAGBIT_COUNTER_ACTIVE : 0);
  return bits;
}

__device__ __forceinline__ float u01(uint32_t r) { return float(r >> 8) * 5.9604644775390625e-08f; }   // [0,1), 24 bits, exact

// ---- v1 task (v1/rl_env_scaledObs.py, v1/rl_env.py): 17-D observation, per-episode waypoint count in flags bits 4-7 ----
template <typename T, int KW>
__device__ __forceinline__ void observe_v1(bool raw, const Env<T, KW>& e, float* o) {        // :65-83
  const int idx = e.flags & 15, K = (e.flags >> 4) & 15;
  T cx, cy, cz;
  current_waypoint(K, e, idx, cx, cy, cz);
  const T c10 = raw ? T(1) : T(0.1), c5 = raw ? T(1) : T(0.2), c2 = raw ? T(1) : T(0.5);
  o[0] = float(e.px * c10); o[1] = float(e.py * c10); o[2] = float(e.pz * c10);
  o[3] = float(e.vx * c5); o[4] = float(e.vy * c5); o[5] = float(e.vz * c5);
  o[6] = float(e.qw); o[7] = float(e.qx); o[8] = float(e.qy); o[9] = float(e.qz);
  o[10] = float(e.wx * c5); o[11] = float(e.wy * c5); o[12] = float(e.wz * c5);
  o[13] = float((cx - e.px) * c2); o[14] = float((cy - e.py) * c2); o[15] = float((cz - e.pz) * c2);
  o[16] = idx >= K - 1 ? 1.0f : 0.0f;                                                         // :80 is_final
}

// One v1 step after the dynamics update: v1/rl_env_scaledObs.py:93-140 (+ _calculate_reward :142-168).
template <typename T, int KW, typename PT>
__device__ __forceinline__ uint32_t task_step_v1(const PT& P, Env<T, KW>& e, T& reward) {
  uint32_t bits = 0;
  int idx = e.flags & 15;
  const int K = (e.flags >> 4) & 15;
  if (P.flags & AMENV_FLAG_NAN_GUARD) {
    const T sum = e.px + e.py + e.pz + e.vx + e.vy + e.vz + e.qw + e.qx + e.qy + e.qz + e.wx + e.wy + e.wz;
    if (!finite_(sum)) {
      if (e.step >= P.max_steps) bits |= AMENV_INFO_TRUNCATED;
      e.step += 1; reward = T(-100);
      return bits | AMENV_INFO_TERMINATED | AMENV_INFO_NONFINITE;
    }
  }
  T cx, cy, cz;
  current_waypoint(K, e, idx, cx, cy, cz);
  const T dx = e.px - cx, dy = e.py - cy, dz = e.pz - cz;
  const T dist = sqrt_(dot3_(dx, dy, dz, dx, dy, dz));                  // :148
  const T v2 = dot3_(e.vx, e.vy, e.vz, e.vx, e.vy, e.vz);
  const T w2 = dot3_(e.wx, e.wy, e.wz, e.wx, e.wy, e.wz);
  const T vn = sqrt_(v2), wn = sqrt_(w2);
  T r_speed = T(-0.1) * v2;                                             // :152
  if (wn > T(0.1)) r_speed -= T(0.01) * w2;                             // :153-154
  T r_prog = T(0);
  if (e.last_distance >= T(0)) {                                        // :158-164
    r_prog = T(20) * (e.last_distance - dist);
    if (r_prog > T(0)) r_prog += T(2);
  }
  e.last_distance = dist;                                               // :166
  reward = ((T(-2) * dist + r_speed) + T(-0.1)) + r_prog;               // :151,155,168
  // approach shaping: velocity component toward the waypoint (:102-110); dist == 0 -> NaN -> neither branch
  const T vtw = -dot3_(e.vx, e.vy, e.vz, dx, dy, dz) * rcp_(dist);
  if (dist < T(0.5)) {
    if (vtw > T(0.1)) reward += T(10);
    else if (vtw < T(0.1)) reward -= T(10);
  }
  bool final_reach = false;
  if (dist < T(0.1)) {                                                  // :111-124
    reward += T(100); idx += 1;
    if (idx >= K) {
      const T rot_b = wn < T(0.1) ? T(100) : T(-20) * wn;               // :122
      const T stop_b = vn < T(0.1) ? T(100) : T(-10) * vn;              // :123
      reward = ((reward + T(400)) + stop_b) + rot_b;                    // :124: returns BEFORE current_step += 1, truncated False
      bits = AMENV_INFO_TERMINATED | AMENV_INFO_SUCCESS | (vn < T(0.1) ? AMENV_INFO_STOPPED : 0u);
      final_reach = true;
    }
  }
  if (!final_reach) {
    if (e.step >= P.max_steps) bits |= AMENV_INFO_TRUNCATED;           // :126
    e.step += 1;                                                        // :127
    if (e.pz < T(0.1)) {                                                // :131-136
      reward -= T(100);
      if (e.vz < T(0)) reward += e.vz * T(100);
      bits |= AMENV_INFO_TERMINATED | AMENV_INFO_CRASHED;
    } else if (sqrt_(dot3_(e.px, e.py, e.pz, e.px, e.py, e.pz)) > T(10)) {  // :137-139
      reward -= T(100);
      bits |= AMENV_INFO_TERMINATED | AMENV_INFO_OOB;
    }
  }
  e.flags = (idx & 15) | (K << 4);
  return bits;
}

// v1 reset (v1/rl_env_scaledObs.py:32-63) from the 12 Philox words; DESIGN.md draw table "v1".
template <typename T, int KW>
__device__ __forceinline__ void reset_from_words_v1(int Kmax, Env<T, KW>& e, const uint32_t* r) {
  int K = 1 + int(r[2] >> 31);                                          // :38 randint(1,3)
  K = K < Kmax ? K : Kmax;
  e.px = T(fmaf(2.0f, u01(r[0]), -1.0f)); e.py = T(fmaf(2.0f, u01(r[1]), -1.0f)); e.pz = T(fmaf(1.0f, u01(r[3]), 1.0f));  // :36-37
  e.vx = e.vy = e.vz = T(0);
  e.qw = T(1); e.qx = e.qy = e.qz = T(0);
  e.wx = e.wy = e.wz = T(0);
#pragma unroll
  for (int k = 0; k < KW; k++) {
    const bool on = k < K && k < 2;
    const int b = 4 + 3 * (k < 2 ? k : 0);
    e.wp[k][0] = on ? T(fmaf(2.0f, u01(r[b]), -1.0f)) : T(0);          // :53-63
    e.wp[k][1] = on ? T(fmaf(2.0f, u01(r[b + 1]), -1.0f)) : T(0);
    e.wp[k][2] = on ? T(fmaf(2.0f, u01(r[b + 2]), 1.0f)) : T(0);
  }
  e.final_yaw = T(0); e.last_distance = T(-1); e.ep_return = T(0);
#pragma unroll
  for (int k = 0; k < AMENV_MAX_JOINTS; k++) { e.th[k] = T(0); e.thd[k] = T(0); }
  e.step = 0; e.counter = 0; e.flags = K << 4;
  e.episode += 1;
}

// ---- counter-based reset RNG: Philox4x32-10, key = seed, counter = (global env id, episode, block)
__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t* out) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = uint64_t(0xD2511F53u) * c0, p1 = uint64_t(0xCD9E8D57u) * c2;  // one v_mad_u64_u32 each
    const uint32_t h0 = uint32_t(p0 >> 32), l0 = uint32_t(p0), h1 = uint32_t(p1 >> 32), l1 = uint32_t(p1);
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}


// WaypointQuadEnv.reset (rl_env_scaledObs.py:40-79) with the DESIGN.md draw table.  All draws
// are formed in fp32 with explicit fmaf so the CPU oracle reproduces them bit for bit.
// The 12 Philox words of one reset, computed by the resetting lane itself (three blocks in sequence).
__device__ __forceinline__ void reset_words_serial(const ColdParams& P, int64_t gid, int32_t episode, uint32_t* r) {
#pragma unroll
  for (uint32_t b = 0; b < 3; b++)
    philox4x32_10(P.seed_lo, P.seed_hi, uint32_t(uint64_t(gid)), uint32_t(uint64_t(gid) >> 32), uint32_t(episode), b, &r[4 * b]);
}

// Same 12 words, wave-cooperative: episode ends are rare, so when a wave enters the reset path usually ONE lane
// needs words while 63 idle.  For each resetting lane (scalar loop) lanes 0..2 each run one of the three Philox
// blocks (block index = lane id) and the 12 results are handed to the owner through v_readlane: one block of
// latency instead of three (the 32x32->64 multiplies are quarter rate).  Must be called by ALL lanes of the wave.
__device__ __forceinline__ void reset_words_wave(const ColdParams& P, bool need, int64_t gid, int32_t episode, uint32_t* r) {
  unsigned long long m = __ballot(need);
  const int lane = int(threadIdx.x & 63);
  const uint32_t g_lo = uint32_t(uint64_t(gid)), g_hi = uint32_t(uint64_t(gid) >> 32);
  while (m) {  // wave-uniform
    const int l = __builtin_ctzll(m);
    m &= m - 1;
    const uint32_t c0 = __builtin_amdgcn_readlane(g_lo, l), c1 = __builtin_amdgcn_readlane(g_hi, l);
    const uint32_t c2 = uint32_t(__builtin_amdgcn_readlane(episode, l));
    uint32_t w[4];
    philox4x32_10(P.seed_lo, P.seed_hi, c0, c1, c2, uint32_t(lane), w);   // lanes 0,1,2 hold blocks 0,1,2
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t v = __builtin_amdgcn_readlane(w[k], b);
        if (lane == l) r[4 * b + k] = v;
      }
  }
}

// WaypointQuadEnv.reset (rl_env_scaledObs.py:40-79) from its 12 Philox words r[] (DESIGN.md draw table).  All draws
// are formed in fp32 with explicit fmaf so the CPU oracle reproduces them bit for bit.
template <typename T, int KW>
__device__ __forceinline__ void reset_from_words(const ColdParams& P, int K, Env<T, KW>& e, const uint32_t* r) {
  const float PIF = 3.14159274101257324f;
  const float sx = fmaf(2.0f, u01(r[0]), -1.0f);                        // :44
  const float sy = fmaf(2.0f, u01(r[1]), -1.0f);
  const float sz = fmaf(1.0f, u01(r[3]), 1.0f);                         // :45
  const float sel0 = u01(r[4]), sel1 = u01(r[5]);                       // :63,65
  const float ex = fmaf(2.0f, u01(r[6]), -1.0f);                        // utils2/utils.py:15-16
  const float ey = fmaf(2.0f, u01(r[7]), -1.0f);
  const float ez = fmaf(2.5f, u01(r[9]), 0.5f);
  const uint32_t axis = r[10] % 3u;                                     // utils2/utils.py:32
  const float fyaw = fmaf(2.0f * PIF, u01(r[11]), -PIF);                // :72
  const int kind = sel0 < 0.3f ? 0 : (sel1 < 0.6f ? 1 : 2);
#pragma unroll
  for (int k = 1; k <= KW; k++) {
    if (k > K) break;
    float wx, wy, wz;
    if (kind == 2) {                                                    // helical, utils2/utils.py:61-95
      wx = fmaf(0.8f, P.traj_cos[k - 1], sx);
      wy = fmaf(0.8f, P.traj_sin[k - 1], sy);
      wz = fmaxf(fmaf(float(k), 0.4f, sz), 0.2f);
    } else {                                                            // linear / curved, :12-57
      const float t = float(k) / float(K);
      wx = fmaf(t, ex - sx, sx); wy = fmaf(t, ey - sy, sy); wz = fmaf(t, ez - sz, sz);
      if (kind == 1) {
        const float s = P.traj_sin[k - 1];
        if (axis == 0) wz = wz + s; else if (axis == 1) wy = wy + s; else wx = wx + s;
        wz = fmaxf(wz, 0.2f);
      }
    }
    e.wp[k - 1][0] = T(wx); e.wp[k - 1][1] = T(wy); e.wp[k - 1][2] = T(wz);
  }
  e.px = T(sx); e.py = T(sy); e.pz = T(sz);
  e.vx = e.vy = e.vz = T(0);
  e.qw = T(1); e.qx = e.qy = e.qz = T(0);                               // quadcopter.py:28-38, attitude (0,0,0)
  e.wx = e.wy = e.wz = T(0);
  e.final_yaw = T(fyaw);
#pragma unroll
  for (int k = 0; k < AMENV_MAX_JOINTS; k++) { e.th[k] = T(0); e.thd[k] = T(0); }     // arm at home (the caller sets the tool offset)
  e.last_distance = T(-1);                                              // :76 None
  e.ep_return = T(0);
  e.step = 0; e.counter = 0; e.flags = 0;                               // :55-59,74-77
  e.episode += 1;
}

template <typename T, int KW>
__device__ __forceinline__ void reset_env(const ColdParams& P, int K, bool v1, Env<T, KW>& e, int64_t gid) {
  uint32_t r[12];
  reset_words_serial(P, gid, e.episode, r);
  if (v1) reset_from_words_v1<T, KW>(K, e, r);
  else reset_from_words<T, KW>(P, K, e, r);
}

}  // namespace amenv_dev
