// amenv_team_host.hpp -- host side of the lane-team kernels: the per-lane constant table and the wave-uniform parameters, from an
// amenv_config.  Plain C++ (no HIP): included by amenv_capi.hip (product) and by tests/emu/team_emu.cpp (host emulation of the arithmetic).
#pragma once
#include <cstring>
#include <vector>

#include "../../include/amenv.h"
#include "amenv_team_math.hpp"

namespace amenv_dev {

// Table entry k of lane column l = 4 * quad + component.  packed4: float4 pieces [k / 4][lane][k % 4] (one 16-byte load per piece and
// lane: the fp32 kernels); otherwise [k][lane].
template <typename T>
std::vector<T> team_const_table(const amenv_config& c, bool packed4) {
  const amenv_vehicle& v = c.vehicle;
  constexpr int NC4 = (kTeamConsts + 3) / 4;
  std::vector<T> t(size_t(NC4) * 4 * 16, T(0));   // [k][lane]
  const float inv_pi = float(0.31830988618379067154);
  const int ns = c.task.rk4_substeps > 0 ? c.task.rk4_substeps : 1;
  const T h = T(c.task.dt / ns), hh = T(0.5) * h, h6 = h * T(1.0 / 6.0);   // the kernel's own h, h/2, h/6 (same roundings)
  for (int b = 0; b < 4; b++)
    for (int cc = 0; cc < 4; cc++) {
      const int l = 4 * b + cc;
      auto set = [&](int k, double x) { t[size_t(k) * 16 + l] = T(x); };
      for (int k = 0; k < 3; k++) set(TC_E0 + k, cc == k ? 1.0 : 0.0);
      if (cc < 3)
        for (int j = 0; j < 3; j++) set(TC_I0C0 + j, v.inertia[3 * cc + j]);
      for (int k = 0; k < 3; k++) {
        const double* I = &v.link_inertia[9 * k];
        const T tr = T(I[0]) + T(I[4]) + T(I[8]);
        for (int j = 0; j < 3; j++) t[size_t(TC_NTR0 + 3 * k + j) * 16 + l] = cc == j ? -tr : T(0);
      }
      for (int r = 0; r < v.n_rotors && r < 6; r++) { set(TC_ALLOC0 + r, v.alloc[r * 4 + cc]); set(TC_MIX0 + r, v.mix[cc * v.n_rotors + r]); }
      {   // lane-team mixer: this lane's rotor = 4 b + cc
        const int rot = 4 * b + cc;
        const bool has = rot < v.n_rotors && rot < 6;
        for (int j = 0; j < 4; j++) set(TC_ALC0 + j, has ? v.alloc[rot * 4 + j] : 0.0);
        set(TC_TMIN, has ? v.t_min[rot] : 0.0); set(TC_TMAX, has ? v.t_max[rot] : 0.0);
        for (int j = 0; j < 4; j++) { const int rj = 4 * b + j; set(TC_MXQ0 + j, (rj < v.n_rotors && rj < 6) ? v.mix[cc * v.n_rotors + rj] : 0.0); }
      }
      set(TC_E01, cc < 2 ? 1.0 : 0.0); set(TC_ZX, cc == 0 ? -1.0 : (cc == 1 ? 1.0 : 0.0));
      if (cc < 3 && v.n_joints > 0) {
        for (int j = 0; j < 3; j++) set(TC_I1C0 + j, v.link_inertia[3 * cc + j]);
        set(TC_LC0, v.link_com[cc]); set(TC_O1, v.joint_origin[3 + cc]);
      }
      const double sp[4] = {.5, -.5, .5, -.5}, sq[4] = {.5, -.5, -.5, .5}, sr[4] = {.5, .5, -.5, -.5};
      set(TC_SP, sp[cc]); set(TC_SQ, sq[cc]); set(TC_SR, sr[cc]);
      t[size_t(TC_ACT1) * 16 + l] = T(cc == 0 ? float(v.mass) : float(v.moment_scale));
      t[size_t(TC_ACT2) * 16 + l] = T(cc == 0 ? float(v.g) : 1.0f);
      if (cc < 3 && v.n_joints > 0) {
        const float lo = float(v.joint_limit[2 * cc]), hi = float(v.joint_limit[2 * cc + 1]);
        t[size_t(TC_JHALF) * 16 + l] = T(0.5f * (hi - lo)); t[size_t(TC_JMID) * 16 + l] = T(0.5f * (hi + lo));
        set(TC_O0, v.joint_origin[cc]);
      }
      set(TC_GV, cc == 2 ? -v.g : 0.0); set(TC_GV1, cc == 1 ? -v.g : 0.0); set(TC_GV2, cc == 0 ? -v.g : 0.0);
      const float oa[4] = {0.1f, 0.2f, 1.0f, 0.2f}, ob[4] = {0.5f, 0.0f, inv_pi, inv_pi}, oc[4] = {0.2f, 2.0f, 0.0f, 0.0f};
      t[size_t(TC_OBS_A) * 16 + l] = T(oa[b]); t[size_t(TC_OBS_B) * 16 + l] = T(ob[b]); t[size_t(TC_OBS_C) * 16 + l] = T(oc[b]);
      const T hs[4] = {T(0), hh, hh, h}, wg[4] = {h6, T(2) * h6, T(2) * h6, h6}, pw[4] = {h * h6, h * h6, h * h6, T(0)};
      t[size_t(TC_HSTEP) * 16 + l] = hs[b]; t[size_t(TC_WGT) * 16 + l] = wg[b]; t[size_t(TC_PW) * 16 + l] = pw[b];
    }
  if (!packed4) return t;
  std::vector<T> packed(t.size(), T(0));
  for (int k = 0; k < NC4 * 4; k++)
    for (int l = 0; l < 16; l++) packed[(size_t(k / 4) * 16 + l) * 4 + k % 4] = t[size_t(k) * 16 + l];
  return packed;
}

template <typename T>
TeamParamsT<T> make_team_params(const amenv_config& c, const void* consts) {
  const amenv_vehicle& v = c.vehicle;
  TeamParamsT<T> P;
  std::memset(&P, 0, sizeof(P));
  for (int j = 0; j < 3; j++) {
    P.o1[j] = T(v.joint_origin[3 + j]); P.o2[j] = T(v.joint_origin[6 + j]); P.tool[j] = T(v.tool_offset[j]);
    P.ee_home[j] = T(v.joint_origin[j] + v.joint_origin[3 + j] + v.joint_origin[6 + j] + v.tool_offset[j]);
  }
  for (int k = 0; k < 3; k++) {
    P.lm[k] = T(v.link_mass[k]);
    for (int j = 0; j < 3; j++) P.lcm[k][j] = T(v.link_com[3 * k + j]);
    const double* I = &v.link_inertia[9 * k];
    const double six[6] = {I[0], I[1], I[2], I[4], I[5], I[8]};
    for (int j = 0; j < 6; j++) P.li[k][j] = T(six[j]);
  }
  P.kp = T(v.joint_kp); P.kd = T(v.joint_kd); P.amax = T(v.joint_acc_max);
  P.mtot = T(v.mass); P.inv_mtot = T(1.0 / v.mass); P.g = T(v.g);
  const int ns = c.task.rk4_substeps > 0 ? c.task.rk4_substeps : 1;
  P.h = T(c.task.dt / ns); P.substeps = ns;
  P.max_steps = c.task.max_episode_steps; P.counter_limit = c.task.counter_limit; P.flags = c.flags;
  P.ee_task = c.task.ee_task == AMENV_EE_TASK_TOOL ? 1 : 0;
  P.K = 1;
  P.consts = consts;
  return P;
}

}  // namespace amenv_dev
