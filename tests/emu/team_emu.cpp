// team_emu.cpp -- TEST INFRASTRUCTURE: a host emulation of the lane-team kernels' arithmetic (csrc/amenv_team_math.hpp compiled with g++).
//
// One env = one DPP row of 16 lanes; here the 16 lanes are explicit (HV = 16 doubles) and the cross-lane primitives (quad_perm, row_ror,
// row_shr) are permutations of that array with the hardware's semantics.  The SAME source as the HIP kernels, so a wrong selector, a
// mis-associated row sum or a wrong hand-over direction in the formulation shows up against the fp64 oracle in the CPU suite
// (tests/test_team_emu_cpu.py).  Nothing in the product loads this library; the HIP kernels themselves are checked on the GPU
// (tests/test_gpu_arm.py: fp64 build <= 1e-12, fp32 build to rounding).
#include <cmath>
#include <cstdint>
#include <cstring>

#define AMENV_FN inline

namespace amenv_dev {

struct HV {
  double x[16];
  HV() { for (int i = 0; i < 16; i++) x[i] = 0.0; }
  HV(double v) { for (int i = 0; i < 16; i++) x[i] = v; }   // implicit: wave-uniform scalars broadcast
};
struct HM { bool m[16]; };

#define HV_BINOP(op) inline HV operator op(const HV& a, const HV& b) { HV r; for (int i = 0; i < 16; i++) r.x[i] = a.x[i] op b.x[i]; return r; }
HV_BINOP(+) HV_BINOP(-) HV_BINOP(*)
#undef HV_BINOP
inline HV operator-(const HV& a) { HV r; for (int i = 0; i < 16; i++) r.x[i] = -a.x[i]; return r; }

inline double fma_(double a, double b, double c) { return std::fma(a, b, c); }
inline HV fma_(const HV& a, const HV& b, const HV& c) { HV r; for (int i = 0; i < 16; i++) r.x[i] = std::fma(a.x[i], b.x[i], c.x[i]); return r; }
inline HV rcp_(const HV& a) { HV r; for (int i = 0; i < 16; i++) r.x[i] = 1.0 / a.x[i]; return r; }
inline HV rsqrt_(const HV& a) { HV r; for (int i = 0; i < 16; i++) r.x[i] = 1.0 / std::sqrt(a.x[i]); return r; }
inline HV clamp_(const HV& a, const HV& lo, const HV& hi) { HV r; for (int i = 0; i < 16; i++) r.x[i] = std::fmax(std::fmin(a.x[i], hi.x[i]), lo.x[i]); return r; }
inline void sincos_t(const HV& a, HV& s, HV& c) { for (int i = 0; i < 16; i++) { s.x[i] = std::sin(a.x[i]); c.x[i] = std::cos(a.x[i]); } }
inline HV sel(const HM& m, const HV& a, const HV& b) { HV r; for (int i = 0; i < 16; i++) r.x[i] = m.m[i] ? a.x[i] : b.x[i]; return r; }
// the action scalings are fp32 arithmetic, left to right, in every build (v2/rl_env_scaledObs.py:125-126)
inline HV scale_action_f32(const HV& a, const HV& s1, const HV& s2) {
  HV r;
  for (int i = 0; i < 16; i++) { volatile float t = float(a.x[i]) * float(s1.x[i]); volatile float u = t * float(s2.x[i]); r.x[i] = double(u); }
  return r;
}
inline HV joint_cmd_f32(const HV& a, const HV& half, const HV& mid) {
  HV r;
  for (int i = 0; i < 16; i++) { volatile float t = fmaf(float(a.x[i]), float(half.x[i]), float(mid.x[i])); r.x[i] = double(t); }
  return r;
}
// DPP: lane c of every quad reads lane P_c of its quad
template <int P0, int P1, int P2, int P3> inline HV qp(const HV& v) {
  const int p[4] = {P0, P1, P2, P3};
  HV r;
  for (int i = 0; i < 16; i++) r.x[i] = v.x[(i & ~3) + p[i & 3]];
  return r;
}
template <int N> inline HV row_ror(const HV& v) { HV r; for (int i = 0; i < 16; i++) r.x[i] = v.x[(i + 16 - N) & 15]; return r; }   // rotate right within the row
template <int N> inline HV row_shr(const HV& v) { HV r; for (int i = 0; i < 16; i++) r.x[i] = i >= N ? v.x[i - N] : 0.0; return r; }   // lane i reads lane i - N; 0 shifted in

}  // namespace amenv_dev

#include "../../rl-aerial-manipulator_amd/csrc/amenv_team_math.hpp"
namespace amenv_dev { template <> struct LaneTraits<HV> { using T = double; using M = HM; }; }
#include "../../rl-aerial-manipulator_amd/csrc/amenv_team_host.hpp"

using namespace amenv_dev;

// One control step of the dynamics (mixer, joint commands, RK4, renormalisation) of n envs.
//   s19    [n][19] in/out: p(3) v(3) q(4) w(3) th(3) thd(3)  (the oracle's orc_arm_dynamics_step layout)
//   act7   [n][7]
//   eo     [n][3] or NULL: tool offset of the new state, world axes
//   probe  [n][4][13] or NULL: per RK4 stage (last sub-step) dV(3) dQ(4) dW(3) and the stage's w(3)
//   quad_check [n] or NULL: max |difference| of the new base state between the four quads of the row (must be exactly 0)
extern "C" int team_emu_step(const amenv_config* cfg, double* s19, const float* act7, int n, double* eo, double* probe, double* quad_check) {
  if (!cfg || cfg->vehicle.n_joints != 3 || cfg->vehicle.n_rotors != 6) return -1;
  const std::vector<double> tab = team_const_table<double>(*cfg, false);
  const TeamParamsT<double> P = make_team_params<double>(*cfg, nullptr);
  HV c[kTeamConsts];
  for (int k = 0; k < kTeamConsts; k++)
    for (int l = 0; l < 16; l++) c[k].x[l] = tab[size_t(k) * 16 + l];
  HM q0, q1, q2;
  for (int l = 0; l < 16; l++) { q0.m[l] = (l >> 2) == 0; q1.m[l] = (l >> 2) == 1; q2.m[l] = (l >> 2) == 2; }
  for (int i = 0; i < n; i++) {
    double* s = s19 + size_t(i) * 19;
    const float* a = act7 + size_t(i) * 7;
    TeamStateT<HV> y;
    HV act, actj;
    for (int l = 0; l < 16; l++) {
      const int cc = l & 3;
      y.P.x[l] = cc < 3 ? s[cc] : 0.0; y.V.x[l] = cc < 3 ? s[3 + cc] : 0.0; y.Q.x[l] = s[6 + cc]; y.W.x[l] = cc < 3 ? s[10 + cc] : 0.0;
      y.TH.x[l] = cc < 3 ? s[13 + cc] : 0.0; y.THD.x[l] = cc < 3 ? s[16 + cc] : 0.0;
      act.x[l] = a[cc]; actj.x[l] = a[4 + (cc < 3 ? cc : 2)];
    }
    TeamStageDeriv<HV> pr;
    const HV EO = team_dynamics(P, c, q0, q1, q2, y, act, actj, &pr);
    if (quad_check) {
      double d = 0.0;
      for (int l = 4; l < 16; l++) {
        const int cc = l & 3;
        if (cc < 3) { d = std::fmax(d, std::fabs(y.P.x[l] - y.P.x[cc])); d = std::fmax(d, std::fabs(y.V.x[l] - y.V.x[cc])); d = std::fmax(d, std::fabs(y.W.x[l] - y.W.x[cc])); }
        d = std::fmax(d, std::fabs(y.Q.x[l] - y.Q.x[cc]));
        if (cc < 3) d = std::fmax(d, std::fabs(EO.x[l] - EO.x[cc]));
      }
      quad_check[i] = d;
    }
    for (int cc = 0; cc < 3; cc++) { s[cc] = y.P.x[cc]; s[3 + cc] = y.V.x[cc]; s[10 + cc] = y.W.x[cc]; s[13 + cc] = y.TH.x[cc]; s[16 + cc] = y.THD.x[cc]; }
    for (int cc = 0; cc < 4; cc++) s[6 + cc] = y.Q.x[cc];
    if (eo) for (int cc = 0; cc < 3; cc++) eo[size_t(i) * 3 + cc] = EO.x[cc];
    if (probe)
      for (int st = 0; st < 4; st++) {
        double* o = probe + (size_t(i) * 4 + st) * 13;
        for (int cc = 0; cc < 3; cc++) { o[cc] = pr.dV.x[4 * st + cc]; o[7 + cc] = pr.dW.x[4 * st + cc]; o[10 + cc] = pr.Ws.x[4 * st + cc]; }
        for (int cc = 0; cc < 4; cc++) o[3 + cc] = pr.dQ.x[4 * st + cc];
      }
  }
  return 0;
}
