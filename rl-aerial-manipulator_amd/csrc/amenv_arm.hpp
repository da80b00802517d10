// amenv_arm.hpp -- device dynamics of the hexacopter carrying the 3-joint arm (BASELINE config 3), DESIGN.md "arm".
//
// No reference code simulates this vehicle (the reference uses Gazebo for it): parity unpinned.  Parameters come from
// the repo's SDF files (tools/arm_params.py).  Formulation (same as the fp64 oracle, written independently):
//   state   [p(3) world, v(3) world, q(4), w(3) body, th(3), thd(3)]
//   joints  acceleration-limited position servos: thdd = clamp(kp (cmd - th) - kd thd, +-amax)
//   base    exact rigid multibody reaction.  With r_k, u_k, a_k the CoM position / velocity / acceleration of body k
//           relative to the body frame, J_k its inertia in body axes, w_k / al_k its relative angular velocity /
//           acceleration, Newton-Euler summed about the body origin O gives, for A = acceleration of O (body
//           components) and wd = angular acceleration of the base,
//             mtot A - S x wd = f      S = sum m_k r_k                     f = F_ext - sum m_k b_k
//             S x A + I_O wd  = n      I_O = sum J_k + m_k(|r_k|^2 1 - r_k r_k^T)
//                                       n = M_ext,O - sum [ m_k r_k x b_k + J_k(al_k + w x w_k) + W_k x (J_k W_k) ]
//           b_k = w x (w x r_k) + 2 w x u_k + a_k,  W_k = w + w_k;  solved as  I_c wd = n - S x f / mtot  with the
//           composite inertia about the system CoM  I_c = I_O - (|S|^2 1 - S S^T)/mtot,  then A = (f + S x wd)/mtot.
#pragma once
#include <type_traits>

#include "amenv_model.hpp"

namespace amenv_dev {

template <typename T>
struct ArmParams {
  T jo[3][3];      // joint origin in the parent frame
  T ja[3][3];      // joint axis (unit)
  T lm[3];         // link mass
  T lc[3][3];      // link CoM in the link frame
  T li[3][6];      // link inertia about its CoM, link frame: xx xy xz yy yz zz
  T kp, kd, amax;
  T mtot, inv_mtot;
  float mid[3], half[3];  // joint command = fmaf(action, half, mid), formed in fp32
  int32_t generic_axes;   // 0: joint axes are (z, x, x) -> AxesZXX fast path; 1: general axes
  T tool[3];              // tool point in the last link's frame (manipulator.sdf:371,450)
  T ee_home[3];           // tool point relative to the body origin with the arm at home (all joints 0), formed in fp64 on the host
};

// Joint-axis pattern of an instantiation: 0/1/2 = the joint axis is +x/+y/+z (compile time, so R's columns are picked by constant
// index), -1 = general axis (Rodrigues).  The repo's arm is <2, 0, 0> (manipulator.sdf:103,163,237).
template <int A0, int A1, int A2> struct Axes { static constexpr int code[3] = {A0, A1, A2}; };
using AxesZXX = Axes<2, 0, 0>;
using AxesAny = Axes<-1, -1, -1>;

template <typename T> struct V3 { T x, y, z; };
template <typename T> __device__ __forceinline__ V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> __device__ __forceinline__ V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> __device__ __forceinline__ V3<T> cross(V3<T> a, V3<T> b) {
  return {fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x))};
}
template <typename T> __device__ __forceinline__ T dot(V3<T> a, V3<T> b) { return dot3_(a.x, a.y, a.z, b.x, b.y, b.z); }

template <typename T> struct M3 { T m[9]; };   // row-major
template <typename T> __device__ __forceinline__ V3<T> mul(const M3<T>& A, V3<T> v) {
  return {dot3_(A.m[0], A.m[1], A.m[2], v.x, v.y, v.z), dot3_(A.m[3], A.m[4], A.m[5], v.x, v.y, v.z), dot3_(A.m[6], A.m[7], A.m[8], v.x, v.y, v.z)};
}
template <typename T> __device__ __forceinline__ V3<T> mulT(const M3<T>& A, V3<T> v) {  // A^T v
  return {dot3_(A.m[0], A.m[3], A.m[6], v.x, v.y, v.z), dot3_(A.m[1], A.m[4], A.m[7], v.x, v.y, v.z), dot3_(A.m[2], A.m[5], A.m[8], v.x, v.y, v.z)};
}
template <typename T> __device__ __forceinline__ M3<T> mul(const M3<T>& A, const M3<T>& B) {
  M3<T> C;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) C.m[3 * i + j] = dot3_(A.m[3 * i], A.m[3 * i + 1], A.m[3 * i + 2], B.m[j], B.m[3 + j], B.m[6 + j]);
  return C;
}
__device__ __forceinline__ void sincos_(float x, float& s, float& c) { s = __sinf(x); c = __cosf(x); }   // v_sin/v_cos, |x| <= pi
__device__ __forceinline__ void sincos_(double x, double& s, double& c) { sincos(x, &s, &c); }

template <typename T>
__device__ __forceinline__ M3<T> rodrigues(V3<T> a, T th) {   // rotation by th about the unit axis a
  T s, c;
  sincos_(th, s, c);
  const T t = T(1) - c;
  M3<T> R;
  R.m[0] = fma_(a.x * a.x, t, c);        R.m[1] = fma_(a.x * a.y, t, -(a.z * s)); R.m[2] = fma_(a.x * a.z, t, a.y * s);
  R.m[3] = fma_(a.y * a.x, t, a.z * s);   R.m[4] = fma_(a.y * a.y, t, c);         R.m[5] = fma_(a.y * a.z, t, -(a.x * s));
  R.m[6] = fma_(a.z * a.x, t, -(a.y * s)); R.m[7] = fma_(a.z * a.y, t, a.x * s);   R.m[8] = fma_(a.z * a.z, t, c);
  return R;
}

// R <- R * Rot(axis c, th) for a coordinate axis: only the two other columns mix (12 FMAs instead of Rodrigues + a 3x3 product)
template <int c, typename T>
__device__ __forceinline__ void rotate_about_column(M3<T>& R, T th) {
  T s, co;
  sincos_(th, s, co);
  constexpr int a = c == 0 ? 1 : (c == 1 ? 2 : 0), b = c == 0 ? 2 : (c == 1 ? 0 : 1);   // (a, b, c) is a cyclic permutation
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const T ra = R.m[3 * i + a], rb = R.m[3 * i + b];
    R.m[3 * i + a] = fma_(co, ra, s * rb);
    R.m[3 * i + b] = fma_(co, rb, -(s * ra));
  }
}

// Forward kinematics (north_star "arm forward kinematics"): tool point relative to the body origin in WORLD axes, for the unit
// quaternion (qw, qx, qy, qz) and joint angles th.  Chain: joint origins / axes (manipulator.sdf:99,159,233 / :103,163,237), then
// the tool offset in the last link's frame (:371,450); world = Rq^T . body, Rq as in the dynamics below.
template <typename AX, typename T>
__device__ __forceinline__ V3<T> ee_offset_world(const ArmParams<T>& A, T qw, T qx, T qy, T qz, const T* th) {
  V3<T> b;
  if constexpr (AX::code[0] == 2 && AX::code[1] == 0 && AX::code[2] == 0) {
    // z, x, x arm: joints 2 and 3 turn about the same axis, so  ee_b = o1 + Rz(th1) (o2 + Rx(th2) o3 + Rx(th2 + th3) tool)
    T s1, c1, s2, c2, s3, c3;
    sincos_(th[0], s1, c1); sincos_(th[1], s2, c2); sincos_(th[1] + th[2], s3, c3);
    const T ux = A.jo[1][0] + A.jo[2][0] + A.tool[0];
    const T uy = A.jo[1][1] + fma_(c2, A.jo[2][1], -(s2 * A.jo[2][2])) + fma_(c3, A.tool[1], -(s3 * A.tool[2]));
    const T uz = A.jo[1][2] + fma_(s2, A.jo[2][1], c2 * A.jo[2][2]) + fma_(s3, A.tool[1], c3 * A.tool[2]);
    b = V3<T>{A.jo[0][0] + fma_(c1, ux, -(s1 * uy)), A.jo[0][1] + fma_(s1, ux, c1 * uy), A.jo[0][2] + uz};
  } else {
    M3<T> R{{T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}};
    b = V3<T>{T(0), T(0), T(0)};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      b = b + mul(R, V3<T>{A.jo[k][0], A.jo[k][1], A.jo[k][2]});
      R = mul(R, rodrigues(V3<T>{A.ja[k][0], A.ja[k][1], A.ja[k][2]}, th[k]));
    }
    b = b + mul(R, V3<T>{A.tool[0], A.tool[1], A.tool[2]});
  }
  const T two = T(2);
  M3<T> Rq;
  Rq.m[0] = fma_(-two, fma_(qy, qy, qz * qz), T(1)); Rq.m[1] = two * fma_(qx, qy, -(qw * qz)); Rq.m[2] = two * fma_(qx, qz, qw * qy);
  Rq.m[3] = two * fma_(qx, qy, qw * qz); Rq.m[4] = fma_(-two, fma_(qx, qx, qz * qz), T(1)); Rq.m[5] = two * fma_(qy, qz, -(qw * qx));
  Rq.m[6] = two * fma_(qx, qz, -(qw * qy)); Rq.m[7] = two * fma_(qy, qz, qw * qx); Rq.m[8] = fma_(-two, fma_(qx, qx, qy * qy), T(1));
  return mulT(Rq, b);
}
// runtime choice of the axis pattern (cold kernels: reset, observe, amenv_ee_position)
template <typename T>
__device__ __forceinline__ V3<T> ee_offset_world_any(const ArmParams<T>& A, T qw, T qx, T qy, T qz, const T* th) {
  if (A.generic_axes) return ee_offset_world<AxesAny>(A, qw, qx, qy, qz, th);
  return ee_offset_world<AxesZXX>(A, qw, qx, qy, qz, th);
}

// Two-wave variant (small batches; step_kernel_arm2w): the 64 environments of a tile are integrated by TWO wavefronts of one
// workgroup.  Both carry the full 19-state RK4; per RHS the helper wave computes link 3 (chain across joint 3 + its Newton-Euler
// terms), the main wave the base, links 1-2, the 3x3 solve.  They meet twice per RHS through LDS: helper -> main 21 partial sums,
// main -> helper the 6 solved accelerations.  The partials are added in the order the one-wave code adds them, so both variants
// (and the rollout kernel) give bit-identical trajectories.
// WORDS: rigid kernels whose reset words come from a helper wave; FLAGS: rigid kernel whose helper waves do ALL episode-end work (the main
// wave only reports which lanes ended)
enum { ARM_ROLE_ALL = 0, ARM_ROLE_MAIN = 1, ARM_ROLE_HELPER = 2, ARM_ROLE_WORDS = 3, ARM_ROLE_FLAGS = 4 };
constexpr int kArmPreSlot = 27;     // link 2's body-frame kinematics of the NEXT stage, precomputed by the helper: r u a_ (9) J (6) w al (6)
constexpr int kArmXchgSlots = 48;   // 21 partial sums + wd(3) + vd(3) + 21 precomputed, [slot][64 lanes] floats
// Chain quantities behind joint 2 (rotation, position / velocity / acceleration of the joint-3 origin, angular velocity / acceleration
// of link 2), carried by the helper wave from its look-ahead to its own share of the next RHS.
template <typename T> struct ChainState { T R[9], v[15]; };
struct NoXchg {};
struct NoIdle { __device__ __forceinline__ void operator()(int) const {} };   // helper-wave filler work: nothing
struct LdsXchg {
  float* base; int lane;
  const uint32_t* words;   // [12][64] reset words of the tile's lanes, written by the helper wave
#ifdef AMENV_STAMPS
  unsigned long long* st;   // diagnostic build: accumulated durations of the main wave's RHS phases (own | bar1 | serial | bar2)
  __device__ __forceinline__ unsigned long long now() const {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
  }
#endif
  __device__ __forceinline__ void put(int slot, float v) const { base[slot * 64 + lane] = v; }
  __device__ __forceinline__ float get(int slot) const { return base[slot * 64 + lane]; }
  // barrier fenced for the instruction scheduler on both sides: ALU work must neither sink below nor rise above it, or the two
  // waves stop overlapping (observed: the main wave's share of RHS n+1 was scheduled in front of the barrier that releases the helper)
  __device__ __forceinline__ void sync() const {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
};

// 19 derivatives of the arm vehicle.  y: state, F / M: rotor wrench after the mixer, cmd: joint position commands.
// Two-wave build only: STAGE = index of this RHS in the RK4 step.  The joint servos do not feel the base, so the joint state of
// stage n+1 (y0 + cnext * (thd, thdd) of stage n) is known before stage n's solve: while the main wave is in its serial section the
// helper computes the next stage's chain up to joint 2 and link 2's body-frame kinematics (r, u, a, J) and leaves them in LDS; from
// stage 1 on the main wave only adds the terms that depend on the base's angular velocity.  Same expressions, same order: bit-identical.
template <typename T, typename AX, typename PT, int ROLE = ARM_ROLE_ALL, typename X = NoXchg, int STAGE = 0, typename IW = NoIdle>
__device__ __forceinline__ void arm_rhs_body(const PT& P, const ArmParams<T>& A, const T* y, T F, V3<T> M, const T* cmd, T* d, const X& x = X{},
                                             const T* y0 = nullptr, T cnext = T(0), ChainState<T>* cs = nullptr, const IW& idle = IW{}) {
  constexpr bool kPreIn = ROLE != ARM_ROLE_ALL && STAGE > 0;     // link 2 / the chain were prepared during the previous stage
  constexpr bool kPreOut = ROLE == ARM_ROLE_HELPER && STAGE < 3;  // prepare them for the next stage
  static_assert(ROLE == ARM_ROLE_ALL || (AX::code[0] == 2 && AX::code[1] == 0 && sizeof(T) == 4), "two-wave roles: z,x,x arm, fp32");
#ifdef AMENV_STAMPS
  unsigned long long t0_ = 0, t1_ = 0, t2_ = 0, t3_ = 0;
  if constexpr (ROLE == ARM_ROLE_MAIN) t0_ = x.now();
#endif
  const V3<T> om{y[10], y[11], y[12]};
  // rotation of the normalised quaternion (body -> world is its transpose, as in the rigid model)
  const T n2 = fma_(y[6], y[6], fma_(y[7], y[7], fma_(y[8], y[8], y[9] * y[9])));
  const T in2 = rcp_(n2);
  const T qw = y[6], qx = y[7], qy = y[8], qz = y[9];
  const T two = T(2) * in2;
  M3<T> Rq;
  Rq.m[0] = fma_(-two, fma_(qy, qy, qz * qz), T(1)); Rq.m[1] = two * fma_(qx, qy, -(qw * qz)); Rq.m[2] = two * fma_(qx, qz, qw * qy);
  Rq.m[3] = two * fma_(qx, qy, qw * qz); Rq.m[4] = fma_(-two, fma_(qx, qx, qz * qz), T(1)); Rq.m[5] = two * fma_(qy, qz, -(qw * qx));
  Rq.m[6] = two * fma_(qx, qz, -(qw * qy)); Rq.m[7] = two * fma_(qy, qz, qw * qx); Rq.m[8] = fma_(-two, fma_(qx, qx, qy * qy), T(1));
  const V3<T> gb{-P.g * Rq.m[2], -P.g * Rq.m[5], -P.g * Rq.m[8]};   // gravity in body components
  T thdd[3];
#pragma unroll
  for (int k = 0; k < 3; k++) thdd[k] = clamp_(fma_(A.kp, cmd[k] - y[13 + k], -(A.kd * y[16 + k])), -A.amax, A.amax);
  // base body: r = 0, J = I0
  const M3<T> I0{{P.Ixx, P.Ixy, P.Ixz, P.Ixy, P.Iyy, P.Iyz, P.Ixz, P.Iyz, P.Izz}};
  V3<T> S{T(0), T(0), T(0)}, fb{T(0), T(0), T(0)};
  V3<T> nb = cross(om, mul(I0, om));
  T IO[6] = {P.Ixx, P.Ixy, P.Ixz, P.Iyy, P.Iyz, P.Izz};               // xx xy xz yy yz zz
  // chain kinematics relative to the body frame
  M3<T> R{{T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}};
  V3<T> p{T(0), T(0), T(0)}, pd = p, pdd = p, w = p, al = p;
  // (compile-time recursion over the joints so each joint's axis code is a constant)
  // advance: carry the chain (R, p, pd, pdd, w, al) across joint k.  leaf: link k's CoM motion, inertia and its sums.
  auto advance = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int ac = AX::code[k];
    const V3<T> o{A.jo[k][0], A.jo[k][1], A.jo[k][2]}, ax{A.ja[k][0], A.ja[k][1], A.ja[k][2]};
    const V3<T> Ro = mul(R, o);
    const V3<T> wRo = cross(w, Ro);
    pd = pd + wRo;
    pdd = pdd + cross(al, Ro) + cross(w, wRo);
    p = p + Ro;
    V3<T> z;
    if constexpr (ac >= 0) z = V3<T>{R.m[ac], R.m[3 + ac], R.m[6 + ac]};      // R e_c = column c of R
    else z = mul(R, ax);
    al = al + thdd[k] * z + y[16 + k] * cross(w, z);
    w = w + y[16 + k] * z;
    if constexpr (ac >= 0) rotate_about_column<ac>(R, y[13 + k]);
    else R = mul(R, rodrigues(ax, y[13 + k]));
  };
  // sums of one link given its CoM position / velocity / acceleration (relative to the body frame) and J (body axes)
  auto accumulate = [&](T m, V3<T> r, V3<T> b, const M3<T>& J, V3<T> aa, V3<T> Om) {
    const V3<T> Jaa = mul(J, aa), JOm = mul(J, Om);
    if constexpr (ROLE == ARM_ROLE_HELPER) {   // hand every addend to the main wave separately (same association there)
      const V3<T> mr = m * r, mb = m * b, mrb = m * cross(r, b), gy = cross(Om, JOm);
      const T r2 = dot(r, r);
      const T v[21] = {mr.x, mr.y, mr.z, mb.x, mb.y, mb.z, mrb.x, mrb.y, mrb.z, Jaa.x, Jaa.y, Jaa.z, gy.x, gy.y, gy.z,
                       J.m[0] + m * (r2 - r.x * r.x), J.m[1] - m * (r.x * r.y), J.m[2] - m * (r.x * r.z),
                       J.m[4] + m * (r2 - r.y * r.y), J.m[5] - m * (r.y * r.z), J.m[8] + m * (r2 - r.z * r.z)};
#pragma unroll
      for (int q = 0; q < 21; q++) x.put(q, float(v[q]));
      return;
    }
    S = S + m * r; fb = fb + m * b;
    nb = nb + m * cross(r, b) + Jaa + cross(Om, JOm);
    const T r2 = dot(r, r);                                            // I_O += J + m (|r|^2 1 - r r^T)
    IO[0] += J.m[0] + m * (r2 - r.x * r.x);
    IO[1] += J.m[1] - m * (r.x * r.y);
    IO[2] += J.m[2] - m * (r.x * r.z);
    IO[3] += J.m[4] + m * (r2 - r.y * r.y);
    IO[4] += J.m[5] - m * (r.y * r.z);
    IO[5] += J.m[8] + m * (r2 - r.z * r.z);
  };
  // body-frame kinematics of link k (independent of the base's motion): CoM position / velocity / acceleration, inertia J = R I R^T
  auto leaf_indep = [&](auto kc, V3<T>& r, V3<T>& u, V3<T>& a_, T* J6) {
    constexpr int k = decltype(kc)::value;
    const V3<T> Rc = mul(R, V3<T>{A.lc[k][0], A.lc[k][1], A.lc[k][2]});
    const V3<T> wRc = cross(w, Rc);
    r = p + Rc; u = pd + wRc; a_ = pdd + cross(al, Rc) + cross(w, wRc);
    const M3<T> Ik{{A.li[k][0], A.li[k][1], A.li[k][2], A.li[k][1], A.li[k][3], A.li[k][4], A.li[k][2], A.li[k][4], A.li[k][5]}};
    const M3<T> RI = mul(R, Ik);
    J6[0] = dot3_(RI.m[0], RI.m[1], RI.m[2], R.m[0], R.m[1], R.m[2]); J6[1] = dot3_(RI.m[0], RI.m[1], RI.m[2], R.m[3], R.m[4], R.m[5]);
    J6[2] = dot3_(RI.m[0], RI.m[1], RI.m[2], R.m[6], R.m[7], R.m[8]); J6[3] = dot3_(RI.m[3], RI.m[4], RI.m[5], R.m[3], R.m[4], R.m[5]);
    J6[4] = dot3_(RI.m[3], RI.m[4], RI.m[5], R.m[6], R.m[7], R.m[8]); J6[5] = dot3_(RI.m[6], RI.m[7], RI.m[8], R.m[6], R.m[7], R.m[8]);
  };
  // the terms that involve the base's angular velocity, and the sums
  auto leaf_dep = [&](T m, V3<T> r, V3<T> u, V3<T> a_, const T* J6, V3<T> wl, V3<T> all) {
    const V3<T> b = cross(om, cross(om, r)) + T(2) * cross(om, u) + a_;
    accumulate(m, r, b, M3<T>{{J6[0], J6[1], J6[2], J6[1], J6[3], J6[4], J6[2], J6[4], J6[5]}}, all + cross(om, wl), om + wl);
  };
  auto leaf = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    V3<T> r, u, a_; T J6[6];
    leaf_indep(kc, r, u, a_, J6);
    leaf_dep(A.lm[k], r, u, a_, J6, w, al);
  };
  using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
  if constexpr (AX::code[0] == 2 && AX::code[1] == 0) {
    // The repo's arm: joint 1 about z at the start of the chain (R = 1, p = pd = pdd = w = al = 0), joint 2 about x.  Written
    // out with the zeros / ones of Rz(th1), w = (0,0,wz), al = (0,0,az) removed: ~180 of the ~960 operations of one RHS.
    T s0, c0;
    sincos_(y[13], s0, c0);
    const T wz = y[16], az = thdd[0];
    const V3<T> o0{A.jo[0][0], A.jo[0][1], A.jo[0][2]};
    if constexpr (ROLE != ARM_ROLE_HELPER) {  // link 1
      const T lx = A.lc[0][0], ly = A.lc[0][1], lz = A.lc[0][2];
      const V3<T> Rc{fma_(c0, lx, -(s0 * ly)), fma_(s0, lx, c0 * ly), lz};
      const V3<T> u{-(wz * Rc.y), wz * Rc.x, T(0)};                                   // w x Rc
      const V3<T> r = o0 + Rc;
      const T ax_ = fma_(-az, Rc.y, -(wz * u.y)), ay_ = fma_(az, Rc.x, wz * u.x);      // al x Rc + w x (w x Rc); z part 0
      const V3<T> oor = cross(om, cross(om, r));
      const V3<T> b{fma_(T(-2) * om.z, u.y, oor.x) + ax_, fma_(T(2) * om.z, u.x, oor.y) + ay_,
                    fma_(T(2), fma_(om.x, u.y, -(om.y * u.x)), oor.z)};
      // J = Rz I Rz^T
      const T Ixx = A.li[0][0], Ixy = A.li[0][1], Ixz = A.li[0][2], Iyy = A.li[0][3], Iyz = A.li[0][4], Izz = A.li[0][5];
      const T q00 = fma_(c0, Ixx, -(s0 * Ixy)), q01 = fma_(c0, Ixy, -(s0 * Iyy)), q02 = fma_(c0, Ixz, -(s0 * Iyz));
      const T q10 = fma_(s0, Ixx, c0 * Ixy), q11 = fma_(s0, Ixy, c0 * Iyy), q12 = fma_(s0, Ixz, c0 * Iyz);
      const T Jxx = fma_(q00, c0, -(q01 * s0)), Jxy = fma_(q00, s0, q01 * c0), Jyy = fma_(q10, s0, q11 * c0);
      accumulate(A.lm[0], r, b, M3<T>{{Jxx, Jxy, q02, Jxy, Jyy, q12, q02, q12, Izz}},
                 V3<T>{om.y * wz, -(om.x * wz), az}, V3<T>{om.x, om.y, om.z + wz});
    }
    // across joint 2 (axis x of the frame Rz(th1)): chain quantities at joint 2, then R = Rz(th1) Rx(th2)
    auto chain_to_joint2 = [&](T s0_, T c0_, T wz_, T az_, T th1, T td, T tdd) {
      const T ox = A.jo[1][0], oy = A.jo[1][1], oz = A.jo[1][2];
      const V3<T> Ro{fma_(c0_, ox, -(s0_ * oy)), fma_(s0_, ox, c0_ * oy), oz};
      pd = V3<T>{-(wz_ * Ro.y), wz_ * Ro.x, T(0)};                                        // w x Ro
      pdd = V3<T>{fma_(-az_, Ro.y, -(wz_ * pd.y)), fma_(az_, Ro.x, wz_ * pd.x), T(0)};
      p = o0 + Ro;
      al = V3<T>{fma_(tdd, c0_, -(td * (wz_ * s0_))), fma_(tdd, s0_, td * (wz_ * c0_)), az_};   // + thdd z + thd (w x z), z = (c0, s0, 0)
      w = V3<T>{td * c0_, td * s0_, wz_};
      T s1, c1;
      sincos_(th1, s1, c1);
      R = M3<T>{{c0_, -(c1 * s0_), s1 * s0_, s0_, c1 * c0_, -(s1 * c0_), T(0), s1, c1}};
    };
    if constexpr (!kPreIn) {
      chain_to_joint2(s0, c0, wz, az, y[14], y[17], thdd[1]);
    } else if constexpr (ROLE == ARM_ROLE_HELPER) {   // carried over from this wave's look-ahead during the previous stage
#pragma unroll
      for (int q = 0; q < 9; q++) R.m[q] = cs->R[q];
      p = V3<T>{cs->v[0], cs->v[1], cs->v[2]}; pd = V3<T>{cs->v[3], cs->v[4], cs->v[5]}; pdd = V3<T>{cs->v[6], cs->v[7], cs->v[8]};
      w = V3<T>{cs->v[9], cs->v[10], cs->v[11]}; al = V3<T>{cs->v[12], cs->v[13], cs->v[14]};
    }
    if constexpr (ROLE != ARM_ROLE_HELPER) {
      if constexpr (kPreIn) {                         // link 2's body-frame kinematics came from the helper
        T g[21];
#pragma unroll
        for (int q = 0; q < 21; q++) g[q] = T(x.get(kArmPreSlot + q));
        leaf_dep(A.lm[1], V3<T>{g[0], g[1], g[2]}, V3<T>{g[3], g[4], g[5]}, V3<T>{g[6], g[7], g[8]}, &g[9], V3<T>{g[15], g[16], g[17]},
                 V3<T>{g[18], g[19], g[20]});
      } else {
        leaf(K1{});
      }
    }
    if constexpr (ROLE != ARM_ROLE_MAIN) { advance(K2{}); leaf(K2{}); }
  } else {
    advance(K0{}); leaf(K0{});
    advance(K1{}); leaf(K1{});
    advance(K2{}); leaf(K2{});
  }
  V3<T> wd, vd;
  if constexpr (ROLE == ARM_ROLE_HELPER) {
    // filler work for the helper's idle windows (the kernel passes the Philox blocks of the reset words): part 0 where it arrives early
    // at the first barrier of the step (its share of stage 0 is shorter than the main wave's), part 1 while the main wave does the last
    // solve of the step (no look-ahead there)
    if constexpr (STAGE == 0) idle(0);
    x.sync();                                     // partial sums are in LDS
    if constexpr (STAGE == 3) idle(1);
    if constexpr (kPreOut) {                      // look-ahead while the main wave solves: joint state, chain and link 2 of the next stage
      if constexpr (AX::code[0] == 2 && AX::code[1] == 0) {
        T thn[3], tdn[3], tddn[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          thn[k] = fma_(cnext, y[16 + k], y0[13 + k]);          // = the next stage's s[13+k], s[16+k] as dynamics_arm forms them
          tdn[k] = fma_(cnext, thdd[k], y0[16 + k]);
          tddn[k] = clamp_(fma_(A.kp, cmd[k] - thn[k], -(A.kd * tdn[k])), -A.amax, A.amax);
        }
        T s0n, c0n;
        sincos_(thn[0], s0n, c0n);
        const V3<T> o0{A.jo[0][0], A.jo[0][1], A.jo[0][2]};
        {
          const T ox = A.jo[1][0], oy = A.jo[1][1], oz = A.jo[1][2];
          const V3<T> Ro{fma_(c0n, ox, -(s0n * oy)), fma_(s0n, ox, c0n * oy), oz};
          const T wz_ = tdn[0], az_ = tddn[0], td = tdn[1], tdd = tddn[1];
          pd = V3<T>{-(wz_ * Ro.y), wz_ * Ro.x, T(0)};
          pdd = V3<T>{fma_(-az_, Ro.y, -(wz_ * pd.y)), fma_(az_, Ro.x, wz_ * pd.x), T(0)};
          p = o0 + Ro;
          al = V3<T>{fma_(tdd, c0n, -(td * (wz_ * s0n))), fma_(tdd, s0n, td * (wz_ * c0n)), az_};
          w = V3<T>{td * c0n, td * s0n, wz_};
          T s1, c1;
          sincos_(thn[1], s1, c1);
          R = M3<T>{{c0n, -(c1 * s0n), s1 * s0n, s0n, c1 * c0n, -(s1 * c0n), T(0), s1, c1}};
        }
        V3<T> r2, u2, a2; T J6[6];
        leaf_indep(K1{}, r2, u2, a2, J6);
        const T g[21] = {r2.x, r2.y, r2.z, u2.x, u2.y, u2.z, a2.x, a2.y, a2.z, J6[0], J6[1], J6[2], J6[3], J6[4], J6[5], w.x, w.y, w.z, al.x, al.y, al.z};
#pragma unroll
        for (int q = 0; q < 21; q++) x.put(kArmPreSlot + q, float(g[q]));
#pragma unroll
        for (int q = 0; q < 9; q++) cs->R[q] = R.m[q];
        const T cv[15] = {p.x, p.y, p.z, pd.x, pd.y, pd.z, pdd.x, pdd.y, pdd.z, w.x, w.y, w.z, al.x, al.y, al.z};
#pragma unroll
        for (int q = 0; q < 15; q++) cs->v[q] = cv[q];
      }
    }
    x.sync();                                     // the main wave has solved
    wd = V3<T>{T(x.get(21)), T(x.get(22)), T(x.get(23))};
    vd = V3<T>{T(x.get(24)), T(x.get(25)), T(x.get(26))};
  } else {
    if constexpr (ROLE == ARM_ROLE_MAIN) {        // link 3 from the helper wave, added exactly as `accumulate` would
      // The compiler otherwise sinks this wave's own share of the RHS below the barrier (next to its uses), which serialises the
      // two waves: pin every value it produced so far in front of the barrier.
      asm volatile("" : "+v"(S.x), "+v"(S.y), "+v"(S.z), "+v"(fb.x), "+v"(fb.y), "+v"(fb.z), "+v"(nb.x), "+v"(nb.y), "+v"(nb.z),
                        "+v"(IO[0]), "+v"(IO[1]), "+v"(IO[2]), "+v"(IO[3]), "+v"(IO[4]), "+v"(IO[5]));
      asm volatile("" : "+v"(Rq.m[0]), "+v"(Rq.m[1]), "+v"(Rq.m[2]), "+v"(Rq.m[3]), "+v"(Rq.m[4]), "+v"(Rq.m[5]), "+v"(Rq.m[6]),
                        "+v"(Rq.m[7]), "+v"(Rq.m[8]));
#ifdef AMENV_STAMPS
      t1_ = x.now();
#endif
      x.sync();
#ifdef AMENV_STAMPS
      t2_ = x.now();
#endif
      S = S + V3<T>{T(x.get(0)), T(x.get(1)), T(x.get(2))};
      fb = fb + V3<T>{T(x.get(3)), T(x.get(4)), T(x.get(5))};
      nb = nb + V3<T>{T(x.get(6)), T(x.get(7)), T(x.get(8))} + V3<T>{T(x.get(9)), T(x.get(10)), T(x.get(11))} +
           V3<T>{T(x.get(12)), T(x.get(13)), T(x.get(14))};
#pragma unroll
      for (int q = 0; q < 6; q++) IO[q] += T(x.get(15 + q));
    }
    // external wrench about O: rotor thrust / moments, gravity at every CoM
    V3<T> f = A.mtot * gb - fb;
    f.z += F;
    const V3<T> n = M + cross(S, gb) - nb;
    // I_c = I_O - (|S|^2 1 - S S^T)/mtot ; solve I_c wd = n - S x f / mtot by the adjugate
    const T S2 = dot(S, S), im = A.inv_mtot;
    const T a = IO[0] - (S2 - S.x * S.x) * im, bq = IO[1] + (S.x * S.y) * im, c = IO[2] + (S.x * S.z) * im;
    const T dd = IO[3] - (S2 - S.y * S.y) * im, e = IO[4] + (S.y * S.z) * im, ff = IO[5] - (S2 - S.z * S.z) * im;
    const V3<T> rhs = n - im * cross(S, f);
    const T c00 = fma_(dd, ff, -(e * e)), c01 = fma_(c, e, -(bq * ff)), c02 = fma_(bq, e, -(c * dd));
    const T c11 = fma_(a, ff, -(c * c)), c12 = fma_(bq, c, -(a * e)), c22 = fma_(a, dd, -(bq * bq));
    const T idet = rcp_(fma_(a, c00, fma_(bq, c01, c * c02)));   // v_rcp_f32 (1 ulp) for fp32: the IEEE division sequence is ~10 dependent instructions on the serial path
    wd = V3<T>{idet * dot3_(c00, c01, c02, rhs.x, rhs.y, rhs.z), idet * dot3_(c01, c11, c12, rhs.x, rhs.y, rhs.z),
               idet * dot3_(c02, c12, c22, rhs.x, rhs.y, rhs.z)};
    const V3<T> Aacc = im * (f + cross(S, wd));
    vd = mulT(Rq, Aacc);                                             // world acceleration of O
    if constexpr (ROLE == ARM_ROLE_MAIN) {
      x.put(21, float(wd.x)); x.put(22, float(wd.y)); x.put(23, float(wd.z));
      x.put(24, float(vd.x)); x.put(25, float(vd.y)); x.put(26, float(vd.z));
#ifdef AMENV_STAMPS
      t3_ = x.now();
#endif
      x.sync();
#ifdef AMENV_STAMPS
      { const unsigned long long t4_ = x.now(); x.st[3] += t1_ - t0_; x.st[4] += t2_ - t1_; x.st[5] += t3_ - t2_; x.st[6] += t4_ - t3_; }
#endif
    }
  }
  d[0] = y[3]; d[1] = y[4]; d[2] = y[5];
  d[3] = vd.x; d[4] = vd.y; d[5] = vd.z;
  const T kq = fma_(T(-2), n2, T(2));
  d[6] = fma_(T(0.5), fma_(om.x, qx, fma_(om.y, qy, om.z * qz)), kq * qw);
  d[7] = fma_(T(-0.5), fma_(om.x, qw, fma_(om.y, qz, -(om.z * qy))), kq * qx);
  d[8] = fma_(T(-0.5), fma_(om.y, qw, fma_(om.z, qx, -(om.x * qz))), kq * qy);
  d[9] = fma_(T(-0.5), fma_(om.z, qw, fma_(om.x, qy, -(om.y * qx))), kq * qz);
  d[10] = wd.x; d[11] = wd.y; d[12] = wd.z;
#pragma unroll
  for (int k = 0; k < 3; k++) { d[13 + k] = y[16 + k]; d[16 + k] = thdd[k]; }
}

template <typename AX, int ROLE = ARM_ROLE_ALL, typename X = NoXchg, int STAGE = 0, typename IW = NoIdle, typename T, typename PT>
__device__ __forceinline__ void arm_rhs(const PT& P, const ArmParams<T>& A, const T* y, T F, V3<T> M, const T* cmd, T* d, const X& x = X{},
                                        const T* y0 = nullptr, T cnext = T(0), ChainState<T>* cs = nullptr, const IW& idle = IW{}) {
  arm_rhs_body<T, AX, PT, ROLE, X, STAGE, IW>(P, A, y, F, M, cmd, d, x, y0, cnext, cs, idle);
}

// One control step of the arm vehicle: mixer as for the rigid body, joint commands from actions 4..6, RK4 on 19 states.
template <typename T, int NROT, int KW, typename AX, int ROLE = ARM_ROLE_ALL, typename X = NoXchg, typename IW = NoIdle>
__device__ __forceinline__ void dynamics_arm(const HotParams<T, NROT>& P, const ArmParams<T>& A, Env<T, KW>& e, const float* act, const X& x = X{},
                                             const IW& idle = IW{}) {
  const float Ff = (act[0] * P.mass_f) * P.g_f;
  const T u0 = T(Ff), u1 = T(act[1] * P.mscale_f), u2 = T(act[2] * P.mscale_f), u3 = T(act[3] * P.mscale_f);
  T F = T(0), Mx = T(0), My = T(0), Mz = T(0);
#pragma unroll
  for (int r = 0; r < NROT; r++) {
    if (NROT == AMENV_MAX_ROTORS && r >= P.n_rotors) break;
    T t = fma_(P.alloc[r][0], u0, fma_(P.alloc[r][1], u1, fma_(P.alloc[r][2], u2, P.alloc[r][3] * u3)));
    t = clamp_(t, P.tmin[r], P.tmax[r]);
    F = F + t; Mx = fma_(P.mixm[0][r], t, Mx); My = fma_(P.mixm[1][r], t, My); Mz = fma_(P.mixm[2][r], t, Mz);
  }
  T cmd[3];
#pragma unroll
  for (int k = 0; k < 3; k++) cmd[k] = T(__builtin_fmaf(act[4 + k], A.half[k], A.mid[k]));
  const V3<T> M{Mx, My, Mz};
  T y[19] = {e.px, e.py, e.pz, e.vx, e.vy, e.vz, e.qw, e.qx, e.qy, e.qz, e.wx, e.wy, e.wz, e.th[0], e.th[1], e.th[2], e.thd[0], e.thd[1], e.thd[2]};
  const T h = P.h, hh = T(0.5) * h, h6 = h * T(1.0 / 6.0);
  int it = 0;
#ifndef AMENV_F64_ARM_UNROLLED
  if constexpr (sizeof(T) == 8 && ROLE == ARM_ROLE_ALL) {
    // fp64 logic-check build: the four RK4 stages as a LOOP around ONE copy of the RHS (stage weights from the stage index).  Four
    // inlined fp64 RHS bodies need > 512 registers; that build spills 110 VGPRs to scratch (332 B per lane, next to 324 spilled SGPRs)
    // and returned garbage on gfx950 / ROCm 7.2 (joint rates of 1e15 after one step: tools/micro/f64_arm_unrolled_repro.py rebuilds it with
    // -DAMENV_F64_ARM_UNROLLED).  The loop needs 256 VGPRs + 202 AGPRs and no scratch.  Same expressions per stage as the unrolled form.
    do {
      T k[19], acc[19], s[19];
#pragma unroll
      for (int i = 0; i < 19; i++) { acc[i] = T(0); s[i] = y[i]; }
#pragma unroll 1
      for (int st = 0; st < 4; st++) {
        arm_rhs<AX, ROLE, X, 0>(P, A, s, F, M, cmd, k, x, y, hh, static_cast<ChainState<T>*>(nullptr));
        const T wgt = (st == 0 || st == 3) ? T(1) : T(2);
        const T cn = st == 2 ? h : hh;
#pragma unroll
        for (int i = 0; i < 19; i++) { acc[i] = fma_(wgt, k[i], acc[i]); s[i] = fma_(cn, k[i], y[i]); }
      }
#pragma unroll
      for (int i = 0; i < 19; i++) y[i] = fma_(h6, acc[i], y[i]);
    } while (++it < P.substeps);
  } else
#endif
  do {
    // RK4 with a running weighted sum (acc = k1 + 2 k2 + 2 k3 + k4): four 19-vectors live instead of six
    T k[19], acc[19], s[19];
    [[maybe_unused]] ChainState<T> cs;   // helper wave: chain behind joint 2, from its look-ahead to the next RHS
    arm_rhs<AX, ROLE, X, 0, IW>(P, A, y, F, M, cmd, k, x, y, hh, &cs, idle);
#pragma unroll
    for (int i = 0; i < 19; i++) { acc[i] = k[i]; s[i] = fma_(hh, k[i], y[i]); }
    arm_rhs<AX, ROLE, X, 1>(P, A, s, F, M, cmd, k, x, y, hh, &cs);
#pragma unroll
    for (int i = 0; i < 19; i++) { acc[i] = fma_(T(2), k[i], acc[i]); s[i] = fma_(hh, k[i], y[i]); }
    arm_rhs<AX, ROLE, X, 2>(P, A, s, F, M, cmd, k, x, y, h, &cs);
#pragma unroll
    for (int i = 0; i < 19; i++) { acc[i] = fma_(T(2), k[i], acc[i]); s[i] = fma_(h, k[i], y[i]); }
    arm_rhs<AX, ROLE, X, 3, IW>(P, A, s, F, M, cmd, k, x, y, T(0), &cs, idle);
#pragma unroll
    for (int i = 0; i < 19; i++) y[i] = fma_(h6, acc[i] + k[i], y[i]);
  } while (++it < P.substeps);
  const T rn = rsqrt_(fma_(y[6], y[6], fma_(y[7], y[7], fma_(y[8], y[8], y[9] * y[9]))));
  e.px = y[0]; e.py = y[1]; e.pz = y[2]; e.vx = y[3]; e.vy = y[4]; e.vz = y[5];
  e.qw = y[6] * rn; e.qx = y[7] * rn; e.qy = y[8] * rn; e.qz = y[9] * rn;
  e.wx = y[10]; e.wy = y[11]; e.wz = y[12];
#pragma unroll
  for (int k = 0; k < 3; k++) { e.th[k] = y[13 + k]; e.thd[k] = y[16 + k]; }
}

// ---- Staged form (step_kernel_armk, amenv_kernels.hpp): joint-configuration aggregates, then the base dynamics on them ---------------
// The joint servos do not feel the base, so the joint state of all four RK4 stages is known from (th, thd, cmd) alone, and every sum
// over the links in the equations at the top of this file can be taken BEFORE the base's angular velocity w is known:
//   fb = sum m_k b_k                                    = w x (w x S) + 2 w x U + Aa            U = sum m_k u_k,  Aa = sum m_k a_k
//   nb = sum [ m_k r_k x b_k + J_k (al_k + w x w_k) + W_k x (J_k W_k) ]  (base body included)
//      = w x (I_O w) + G w + w x H + Tn
//        G  = sum 2 m_k ((u_k . r_k) 1 - u_k r_k^T) - (D_k + D_k^T),  D_k = J_k [w_k]x      (r x (2 w x u);  J (w x w_k) and w_k x (J w))
//        H  = sum J_k w_k                                                                      (w x (J w_k))
//        Tn = sum m_k r_k x a_k + J_k al_k + w_k x (J_k w_k)
//   (sum m_k r_k x (w x (w x r_k)) + w x (J_k w) = w x (I_O w): the point-mass and the link inertias are the ones I_O sums anyway.)
// 36 numbers per stage (S, U, Aa, I_O, G, H, Tn and the scaled adjugate of the composite inertia I_c) carry a stage's joint
// configuration to the base dynamics, which is then ~160 instructions per stage instead of ~960.  The identity is exact (checked in
// fp64 against the oracle's RHS to 2e-15); in fp32 the sums associate differently from arm_rhs_body, so this form is not bit-identical
// to the lane kernel -- same gate (<= 2e-6 rel per step against the fp64 oracle).  z,x,x arm, fp32 or fp64.
constexpr int kAggSlots = 36;
enum { AG_S = 0, AG_U = 3, AG_A = 6, AG_IO = 9, AG_G = 15, AG_H = 24, AG_T = 27, AG_C = 30 };

template <typename T>
__device__ __forceinline__ T servo_(const ArmParams<T>& A, T cmd, T th, T td) { return clamp_(fma_(A.kp, cmd - th, -(A.kd * td)), -A.amax, A.amax); }

template <typename T, typename PT>
__device__ __forceinline__ void arm_kin_aggregates(const PT& P, const ArmParams<T>& A, const T* th, const T* td, const T* tdd, T* g) {
  V3<T> S{T(0), T(0), T(0)}, U = S, Aa = S, H = S, Tn = S;
  T IO[6] = {P.Ixx, P.Ixy, P.Ixz, P.Iyy, P.Iyz, P.Izz};
  T G[9] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
  // one link: CoM position / velocity / acceleration relative to the body frame, inertia (body axes, xx xy xz yy yz zz), relative
  // angular velocity / acceleration.  ZONLY: w = (0, 0, w.z), al = (0, 0, al.z) (link 1 of the z,x,x arm).
  auto add_link = [&](auto zonly, T m, V3<T> r, V3<T> u, V3<T> a_, const T* J, V3<T> w, V3<T> al) {
    constexpr bool ZONLY = decltype(zonly)::value;
    S = S + m * r; U = U + m * u; Aa = Aa + m * a_;
    const T r2 = dot(r, r);
    IO[0] += J[0] + m * (r2 - r.x * r.x); IO[1] += J[1] - m * (r.x * r.y); IO[2] += J[2] - m * (r.x * r.z);
    IO[3] += J[3] + m * (r2 - r.y * r.y); IO[4] += J[4] - m * (r.y * r.z); IO[5] += J[5] + m * (r2 - r.z * r.z);
    const T m2 = m + m, d = m2 * dot(u, r);
    const V3<T> mu = m2 * u;
    T E00, E11, E22, E01, E02, E12;   // D + D^T
    V3<T> Jw, Jal;
    if constexpr (ZONLY) {
      E00 = T(2) * (w.z * J[1]); E11 = -E00; E22 = T(0);
      E01 = w.z * (J[3] - J[0]); E02 = w.z * J[4]; E12 = -(w.z * J[2]);
      Jw = V3<T>{w.z * J[2], w.z * J[4], w.z * J[5]};
      Jal = V3<T>{al.z * J[2], al.z * J[4], al.z * J[5]};
    } else {
      E00 = T(2) * fma_(w.z, J[1], -(w.y * J[2])); E11 = T(2) * fma_(w.x, J[4], -(w.z * J[1])); E22 = T(2) * fma_(w.y, J[2], -(w.x * J[4]));
      E01 = fma_(w.x, J[2], -(w.z * J[0])) + fma_(w.z, J[3], -(w.y * J[4]));
      E02 = fma_(w.y, J[0], -(w.x * J[1])) + fma_(w.z, J[4], -(w.y * J[5]));
      E12 = fma_(w.y, J[1], -(w.x * J[3])) + fma_(w.x, J[5], -(w.z * J[2]));
      Jw = V3<T>{dot3_(J[0], J[1], J[2], w.x, w.y, w.z), dot3_(J[1], J[3], J[4], w.x, w.y, w.z), dot3_(J[2], J[4], J[5], w.x, w.y, w.z)};
      Jal = V3<T>{dot3_(J[0], J[1], J[2], al.x, al.y, al.z), dot3_(J[1], J[3], J[4], al.x, al.y, al.z), dot3_(J[2], J[4], J[5], al.x, al.y, al.z)};
    }
    G[0] += (d - mu.x * r.x) - E00; G[1] += -(mu.x * r.y) - E01; G[2] += -(mu.x * r.z) - E02;
    G[3] += -(mu.y * r.x) - E01; G[4] += (d - mu.y * r.y) - E11; G[5] += -(mu.y * r.z) - E12;
    G[6] += -(mu.z * r.x) - E02; G[7] += -(mu.z * r.y) - E12; G[8] += (d - mu.z * r.z) - E22;
    H = H + Jw;
    V3<T> wJw;
    if constexpr (ZONLY) wJw = V3<T>{-(w.z * Jw.y), w.z * Jw.x, T(0)};
    else wJw = cross(w, Jw);
    Tn = Tn + m * cross(r, a_) + Jal + wJw;
  };
  // link 1: joint 1 about z at the start of the chain
  T s0, c0;
  sincos_(th[0], s0, c0);
  const T wz = td[0], az = tdd[0];
  const V3<T> o0{A.jo[0][0], A.jo[0][1], A.jo[0][2]};
  {
    const T lx = A.lc[0][0], ly = A.lc[0][1], lz = A.lc[0][2];
    const V3<T> Rc{fma_(c0, lx, -(s0 * ly)), fma_(s0, lx, c0 * ly), lz};
    const V3<T> u{-(wz * Rc.y), wz * Rc.x, T(0)};
    const V3<T> a_{fma_(-az, Rc.y, -(wz * u.y)), fma_(az, Rc.x, wz * u.x), T(0)};
    const T Ixx = A.li[0][0], Ixy = A.li[0][1], Ixz = A.li[0][2], Iyy = A.li[0][3], Iyz = A.li[0][4], Izz = A.li[0][5];
    const T q00 = fma_(c0, Ixx, -(s0 * Ixy)), q01 = fma_(c0, Ixy, -(s0 * Iyy)), q02 = fma_(c0, Ixz, -(s0 * Iyz));
    const T q10 = fma_(s0, Ixx, c0 * Ixy), q11 = fma_(s0, Ixy, c0 * Iyy), q12 = fma_(s0, Ixz, c0 * Iyz);
    const T J[6] = {fma_(q00, c0, -(q01 * s0)), fma_(q00, s0, q01 * c0), q02, fma_(q10, s0, q11 * c0), q12, Izz};
    add_link(std::true_type{}, A.lm[0], o0 + Rc, u, a_, J, V3<T>{T(0), T(0), wz}, V3<T>{T(0), T(0), az});
  }
  // chain at joint 2 (axis x of Rz(th1)), then R = Rz(th1) Rx(th2)
  M3<T> R;
  V3<T> p, pd, pdd, w, al;
  {
    const T ox = A.jo[1][0], oy = A.jo[1][1], oz = A.jo[1][2];
    const V3<T> Ro{fma_(c0, ox, -(s0 * oy)), fma_(s0, ox, c0 * oy), oz};
    pd = V3<T>{-(wz * Ro.y), wz * Ro.x, T(0)};
    pdd = V3<T>{fma_(-az, Ro.y, -(wz * pd.y)), fma_(az, Ro.x, wz * pd.x), T(0)};
    p = o0 + Ro;
    al = V3<T>{fma_(tdd[1], c0, -(td[1] * (wz * s0))), fma_(tdd[1], s0, td[1] * (wz * c0)), az};
    w = V3<T>{td[1] * c0, td[1] * s0, wz};
    T s1, c1;
    sincos_(th[1], s1, c1);
    R = M3<T>{{c0, -(c1 * s0), s1 * s0, s0, c1 * c0, -(s1 * c0), T(0), s1, c1}};
  }
  auto link = [&](int k) {
    const V3<T> Rc = mul(R, V3<T>{A.lc[k][0], A.lc[k][1], A.lc[k][2]});
    const V3<T> wRc = cross(w, Rc);
    const V3<T> r = p + Rc, u = pd + wRc, a_ = pdd + cross(al, Rc) + cross(w, wRc);
    const M3<T> Ik{{A.li[k][0], A.li[k][1], A.li[k][2], A.li[k][1], A.li[k][3], A.li[k][4], A.li[k][2], A.li[k][4], A.li[k][5]}};
    const M3<T> RI = mul(R, Ik);
    const T J[6] = {dot3_(RI.m[0], RI.m[1], RI.m[2], R.m[0], R.m[1], R.m[2]), dot3_(RI.m[0], RI.m[1], RI.m[2], R.m[3], R.m[4], R.m[5]),
                    dot3_(RI.m[0], RI.m[1], RI.m[2], R.m[6], R.m[7], R.m[8]), dot3_(RI.m[3], RI.m[4], RI.m[5], R.m[3], R.m[4], R.m[5]),
                    dot3_(RI.m[3], RI.m[4], RI.m[5], R.m[6], R.m[7], R.m[8]), dot3_(RI.m[6], RI.m[7], RI.m[8], R.m[6], R.m[7], R.m[8])};
    add_link(std::false_type{}, A.lm[k], r, u, a_, J, w, al);
  };
  link(1);
  {  // across joint 3 (axis x of the current frame = column 0 of R)
    const V3<T> Ro = mul(R, V3<T>{A.jo[2][0], A.jo[2][1], A.jo[2][2]});
    const V3<T> wRo = cross(w, Ro);
    pd = pd + wRo;
    pdd = pdd + cross(al, Ro) + cross(w, wRo);
    p = p + Ro;
    const V3<T> z{R.m[0], R.m[3], R.m[6]};
    al = al + tdd[2] * z + td[2] * cross(w, z);
    w = w + td[2] * z;
    rotate_about_column<0>(R, th[2]);
  }
  link(2);
  // composite inertia about the system CoM and its adjugate / determinant: wd = C (n - S x f / mtot)
  const T S2 = dot(S, S), im = A.inv_mtot;
  const T a = IO[0] - (S2 - S.x * S.x) * im, bq = IO[1] + (S.x * S.y) * im, c = IO[2] + (S.x * S.z) * im;
  const T dd = IO[3] - (S2 - S.y * S.y) * im, e = IO[4] + (S.y * S.z) * im, ff = IO[5] - (S2 - S.z * S.z) * im;
  const T c00 = fma_(dd, ff, -(e * e)), c01 = fma_(c, e, -(bq * ff)), c02 = fma_(bq, e, -(c * dd));
  const T c11 = fma_(a, ff, -(c * c)), c12 = fma_(bq, c, -(a * e)), c22 = fma_(a, dd, -(bq * bq));
  const T idet = rcp_(fma_(a, c00, fma_(bq, c01, c * c02)));
  g[AG_S] = S.x; g[AG_S + 1] = S.y; g[AG_S + 2] = S.z; g[AG_U] = U.x; g[AG_U + 1] = U.y; g[AG_U + 2] = U.z;
  g[AG_A] = Aa.x; g[AG_A + 1] = Aa.y; g[AG_A + 2] = Aa.z;
#pragma unroll
  for (int q = 0; q < 6; q++) g[AG_IO + q] = IO[q];
#pragma unroll
  for (int q = 0; q < 9; q++) g[AG_G + q] = G[q];
  g[AG_H] = H.x; g[AG_H + 1] = H.y; g[AG_H + 2] = H.z; g[AG_T] = Tn.x; g[AG_T + 1] = Tn.y; g[AG_T + 2] = Tn.z;
  g[AG_C] = idet * c00; g[AG_C + 1] = idet * c01; g[AG_C + 2] = idet * c02; g[AG_C + 3] = idet * c11; g[AG_C + 4] = idet * c12; g[AG_C + 5] = idet * c22;
}

// What one stage wave of step_kernel_armk does: joint state of RK4 stage `stage` (the servo chain up to it, formed exactly as dynamics_arm
// forms its stage states) -> aggregates of that configuration -> agg[stage][slot][lane]; the wave of stage 3 also has all four joint
// derivatives and writes the integrated joint state jn[6][lane] (same expressions as dynamics_arm: the joint trajectories of the staged and
// the lane kernels are bit-identical).
template <typename T, typename PT>
__device__ __forceinline__ void arm_kin_stage(const PT& P, const ArmParams<T>& A, int stage, const T* th0, const T* td0, const T* cmd, T* agg, T* jn,
                                              int lane) {
  const T h = P.h, hh = T(0.5) * h, h6 = h * T(1.0 / 6.0);
  T th[3], td[3], a[3], accp[3], accv[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { th[k] = th0[k]; td[k] = td0[k]; a[k] = servo_(A, cmd[k], th[k], td[k]); accp[k] = td[k]; accv[k] = a[k]; }
  for (int st = 0; st < stage; st++) {     // wave-uniform trip count
    const T cn = st == 2 ? h : hh;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const T thn = fma_(cn, td[k], th0[k]), tdn = fma_(cn, a[k], td0[k]);
      th[k] = thn; td[k] = tdn; a[k] = servo_(A, cmd[k], thn, tdn);
      if (st < 2) { accp[k] = fma_(T(2), td[k], accp[k]); accv[k] = fma_(T(2), a[k], accv[k]); }
    }
  }
  if (stage == 3) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      jn[k * 64 + lane] = fma_(h6, accp[k] + td[k], th0[k]);
      jn[(3 + k) * 64 + lane] = fma_(h6, accv[k] + a[k], td0[k]);
    }
  }
  T g[kAggSlots];
  arm_kin_aggregates<T, PT>(P, A, th, td, a, g);
  T* out = agg + size_t(stage) * kAggSlots * 64 + lane;
#pragma unroll
  for (int q = 0; q < kAggSlots; q++) out[q * 64] = g[q];
}

// 13 base derivatives from a stage's aggregates g.  y: position, velocity, quaternion, body rates.
template <typename T, typename PT>
__device__ __forceinline__ void arm_dyn_agg(const PT& P, const ArmParams<T>& A, const T* y, T F, V3<T> M, const T* g, T* d) {
  const V3<T> om{y[10], y[11], y[12]};
  const T n2 = fma_(y[6], y[6], fma_(y[7], y[7], fma_(y[8], y[8], y[9] * y[9])));
  const T in2 = rcp_(n2);
  const T qw = y[6], qx = y[7], qy = y[8], qz = y[9];
  const T two = T(2) * in2;
  M3<T> Rq;
  Rq.m[0] = fma_(-two, fma_(qy, qy, qz * qz), T(1)); Rq.m[1] = two * fma_(qx, qy, -(qw * qz)); Rq.m[2] = two * fma_(qx, qz, qw * qy);
  Rq.m[3] = two * fma_(qx, qy, qw * qz); Rq.m[4] = fma_(-two, fma_(qx, qx, qz * qz), T(1)); Rq.m[5] = two * fma_(qy, qz, -(qw * qx));
  Rq.m[6] = two * fma_(qx, qz, -(qw * qy)); Rq.m[7] = two * fma_(qy, qz, qw * qx); Rq.m[8] = fma_(-two, fma_(qx, qx, qy * qy), T(1));
  const V3<T> gb{-P.g * Rq.m[2], -P.g * Rq.m[5], -P.g * Rq.m[8]};
  const V3<T> S{g[AG_S], g[AG_S + 1], g[AG_S + 2]}, U{g[AG_U], g[AG_U + 1], g[AG_U + 2]}, Aa{g[AG_A], g[AG_A + 1], g[AG_A + 2]};
  const V3<T> H{g[AG_H], g[AG_H + 1], g[AG_H + 2]}, Tn{g[AG_T], g[AG_T + 1], g[AG_T + 2]};
  const V3<T> fb = cross(om, cross(om, S)) + T(2) * cross(om, U) + Aa;
  const V3<T> IOw{dot3_(g[AG_IO], g[AG_IO + 1], g[AG_IO + 2], om.x, om.y, om.z), dot3_(g[AG_IO + 1], g[AG_IO + 3], g[AG_IO + 4], om.x, om.y, om.z),
                  dot3_(g[AG_IO + 2], g[AG_IO + 4], g[AG_IO + 5], om.x, om.y, om.z)};
  const V3<T> Gw{dot3_(g[AG_G], g[AG_G + 1], g[AG_G + 2], om.x, om.y, om.z), dot3_(g[AG_G + 3], g[AG_G + 4], g[AG_G + 5], om.x, om.y, om.z),
                 dot3_(g[AG_G + 6], g[AG_G + 7], g[AG_G + 8], om.x, om.y, om.z)};
  const V3<T> nb = cross(om, IOw) + Gw + cross(om, H) + Tn;
  V3<T> f = A.mtot * gb - fb;
  f.z += F;
  const V3<T> n = M + cross(S, gb) - nb;
  const V3<T> rhs = n - A.inv_mtot * cross(S, f);
  const V3<T> wd{dot3_(g[AG_C], g[AG_C + 1], g[AG_C + 2], rhs.x, rhs.y, rhs.z), dot3_(g[AG_C + 1], g[AG_C + 3], g[AG_C + 4], rhs.x, rhs.y, rhs.z),
                 dot3_(g[AG_C + 2], g[AG_C + 4], g[AG_C + 5], rhs.x, rhs.y, rhs.z)};
  const V3<T> vd = mulT(Rq, A.inv_mtot * (f + cross(S, wd)));
  d[0] = y[3]; d[1] = y[4]; d[2] = y[5];
  d[3] = vd.x; d[4] = vd.y; d[5] = vd.z;
  const T kq = fma_(T(-2), n2, T(2));
  d[6] = fma_(T(0.5), fma_(om.x, qx, fma_(om.y, qy, om.z * qz)), kq * qw);
  d[7] = fma_(T(-0.5), fma_(om.x, qw, fma_(om.y, qz, -(om.z * qy))), kq * qx);
  d[8] = fma_(T(-0.5), fma_(om.y, qw, fma_(om.z, qx, -(om.x * qz))), kq * qy);
  d[9] = fma_(T(-0.5), fma_(om.z, qw, fma_(om.x, qy, -(om.y * qx))), kq * qz);
  d[10] = wd.x; d[11] = wd.y; d[12] = wd.z;
}

// LDS hand-over of step_kernel_armk: agg [4 stages][kAggSlots][64], jn [6][64] integrated joint state, words [12][64] reset words
template <typename T>
struct StagedXchg {   // (exchanged in the kernel's arithmetic type: the fp64 instantiation is the logic gate of this hand-over)
  const T* agg; const T* jn; const uint32_t* words; int lane;
  __device__ __forceinline__ void sync() const {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
};
constexpr int ARM_ROLE_STAGED = 5;

// The main wave's control step of the staged kernel: mixer, ONE barrier (the stage waves have left their aggregates), RK4 on the 13 base
// states with arm_dyn_agg, joints from the stage-3 wave.  One RK4 sub-step per control step.
template <typename T, int NROT, int KW>
__device__ __forceinline__ void dynamics_arm_staged(const HotParams<T, NROT>& P, const ArmParams<T>& A, Env<T, KW>& e, const float* act, const StagedXchg<T>& x) {
  const float Ff = (act[0] * P.mass_f) * P.g_f;
  const T u0 = T(Ff), u1 = T(act[1] * P.mscale_f), u2 = T(act[2] * P.mscale_f), u3 = T(act[3] * P.mscale_f);
  T F = T(0), Mx = T(0), My = T(0), Mz = T(0);
#pragma unroll
  for (int r = 0; r < NROT; r++) {
    T t = fma_(P.alloc[r][0], u0, fma_(P.alloc[r][1], u1, fma_(P.alloc[r][2], u2, P.alloc[r][3] * u3)));
    t = clamp_(t, P.tmin[r], P.tmax[r]);
    F = F + t; Mx = fma_(P.mixm[0][r], t, Mx); My = fma_(P.mixm[1][r], t, My); Mz = fma_(P.mixm[2][r], t, Mz);
  }
  const V3<T> M{Mx, My, Mz};
  T y[13] = {e.px, e.py, e.pz, e.vx, e.vy, e.vz, e.qw, e.qx, e.qy, e.qz, e.wx, e.wy, e.wz};
  const T h = P.h, hh = T(0.5) * h, h6 = h * T(1.0 / 6.0);
  asm volatile("" : "+v"(F), "+v"(Mx), "+v"(My), "+v"(Mz));   // the mixer stays in front of the barrier
  x.sync();
  T k[13], acc[13], s[13], g[kAggSlots];
  auto stage_agg = [&](int st) {
#pragma unroll
    for (int q = 0; q < kAggSlots; q++) g[q] = x.agg[(st * kAggSlots + q) * 64 + x.lane];
  };
  stage_agg(0);
  arm_dyn_agg<T>(P, A, y, F, M, g, k);
#pragma unroll
  for (int i = 0; i < 13; i++) { acc[i] = k[i]; s[i] = fma_(hh, k[i], y[i]); }
  stage_agg(1);
  arm_dyn_agg<T>(P, A, s, F, M, g, k);
#pragma unroll
  for (int i = 0; i < 13; i++) { acc[i] = fma_(T(2), k[i], acc[i]); s[i] = fma_(hh, k[i], y[i]); }
  stage_agg(2);
  arm_dyn_agg<T>(P, A, s, F, M, g, k);
#pragma unroll
  for (int i = 0; i < 13; i++) { acc[i] = fma_(T(2), k[i], acc[i]); s[i] = fma_(h, k[i], y[i]); }
  stage_agg(3);
  arm_dyn_agg<T>(P, A, s, F, M, g, k);
#pragma unroll
  for (int i = 0; i < 13; i++) y[i] = fma_(h6, acc[i] + k[i], y[i]);
  const T rn = rsqrt_(fma_(y[6], y[6], fma_(y[7], y[7], fma_(y[8], y[8], y[9] * y[9]))));
  e.px = y[0]; e.py = y[1]; e.pz = y[2]; e.vx = y[3]; e.vy = y[4]; e.vz = y[5];
  e.qw = y[6] * rn; e.qx = y[7] * rn; e.qy = y[8] * rn; e.qz = y[9] * rn;
  e.wx = y[10]; e.wy = y[11]; e.wz = y[12];
#pragma unroll
  for (int j = 0; j < 3; j++) { e.th[j] = x.jn[j * 64 + x.lane]; e.thd[j] = x.jn[(3 + j) * 64 + x.lane]; }
}

// Parity gate of the two formulations of the right-hand side (amenv_arm_rhs): the 19 derivatives of n states [n][19] under wrench [n][4]
// (F, Mx, My, Mz after the mixer) and joint commands [n][3], per-link form (arm_rhs_body) or staged form (arm_kin_aggregates + arm_dyn_agg),
// in the arithmetic type T -- the fp64 instantiation is compared with the fp64 oracle's orc_arm_rhs.  z,x,x arm.
template <typename T, int FORM, typename PT>
__global__ void arm_rhs_kernel(const PT P, const ArmParams<T> A, const T* __restrict__ s, const T* __restrict__ wrench, const T* __restrict__ cmd,
                               T* __restrict__ d, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T y[19], c[3], out[19];
#pragma unroll
  for (int k = 0; k < 19; k++) y[k] = s[i * 19 + k];
#pragma unroll
  for (int k = 0; k < 3; k++) c[k] = cmd[i * 3 + k];
  const T F = wrench[i * 4];
  const V3<T> M{wrench[i * 4 + 1], wrench[i * 4 + 2], wrench[i * 4 + 3]};
  if constexpr (FORM == 0) {
    arm_rhs<AxesZXX>(P, A, y, F, M, c, out);
  } else {
    T thdd[3], g[kAggSlots];
#pragma unroll
    for (int k = 0; k < 3; k++) thdd[k] = servo_(A, c[k], y[13 + k], y[16 + k]);
    arm_kin_aggregates<T>(P, A, &y[13], &y[16], thdd, g);
    arm_dyn_agg<T>(P, A, y, F, M, g, out);
#pragma unroll
    for (int k = 0; k < 3; k++) { out[13 + k] = y[16 + k]; out[16 + k] = thdd[k]; }
  }
#pragma unroll
  for (int k = 0; k < 19; k++) d[i * 19 + k] = out[k];
}

}  // namespace amenv_dev
