// amenv_team.hpp -- lane-TEAM step kernel for the hexacopter + z,x,x arm (BASELINE config 3) in the latency regime.
//
// Why.  At 4096 envs a one-lane-per-env launch is 64 wavefronts on a chip with 1024 SIMDs, and its length is ONE lane's instruction
// stream: ~4000 dependent-ish VALU instructions at ~4 clocks each (a lone wave issues one VALU instruction per 4 clocks).  The
// arithmetic itself would take the whole chip 0.2 us.  So the step is spread over more lanes instead of more envs:
//
//   one env = one DPP ROW of 16 lanes = 4 quads x 4 components (x, y, z, spare / quaternion w-z)
//   one wavefront = 4 envs, 4096 envs = 1024 wavefronts = one per SIMD
//
//   * 3-vectors live with ONE component per lane (lanes c = 0..2 of a quad), 3x3 matrices as three column registers with one ROW per
//     lane, quaternions in the 4 lanes of a quad.  Cross products, matrix-vector and matrix-matrix products, R I R^T, the adjugate
//     all become 3-5 instructions instead of 6-45: the operands of the other components come through DPP quad_perm modifiers
//     (folded into v_mul / v_add by the compiler, v_mov_b32_dpp otherwise) -- no LDS, no barrier, no readlane.
//   * inside the RK4 quad s of the row works on STAGE s: the arithmetic, its derivation and the round-3 form of the base dynamics
//     (gravity-free w chain, systolic stage hand-over through row_shr:4, the four stages' translational accelerations side by side) are
//     in amenv_team_math.hpp, which this file instantiates for float (product) and double (fp64 logic-gate build of the same kernel);
//   * per-env scalars (reward logic, state machine, reset) are computed redundantly by all 16 lanes from broadcast copies of the state,
//     with the SAME task_step / reset code as the one-lane kernels (amenv_model.hpp): every branch is uniform within a row, which is
//     what keeps DPP legal inside it (DPP reads of EXEC-disabled lanes return 0).
//
// Same model as amenv_arm.hpp / the oracle; the terms are sorted differently (amenv_team_math.hpp) and sums are trees over lanes, so results
// agree with the one-lane kernel to rounding (tests: <= 3e-6 rel per step against the fp64 oracle for the fp32 build, <= 1e-12 for the fp64
// build), not bit for bit.
#pragma once
#include "amenv_kernels.hpp"

#define AMENV_FN __device__ __forceinline__

namespace amenv_dev {

// ---- DPP primitives for the two device value types ----------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_(double v) {   // register pair: the same lane permutation on both halves
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int P0, int P1, int P2, int P3> __device__ __forceinline__ float qp(float v) { return dpp_<P0 | (P1 << 2) | (P2 << 4) | (P3 << 6)>(v); }   // lane c of every quad reads lane P_c of its quad
template <int P0, int P1, int P2, int P3> __device__ __forceinline__ double qp(double v) { return dpp_<P0 | (P1 << 2) | (P2 << 4) | (P3 << 6)>(v); }
template <int P0, int P1, int P2, int P3> __device__ __forceinline__ int qpi(int v) {
  return __builtin_amdgcn_update_dpp(0, v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true);
}
template <int N> __device__ __forceinline__ float row_ror(float v) { return dpp_<0x120 + N>(v); }    // rotate within the 16-lane row
template <int N> __device__ __forceinline__ double row_ror(double v) { return dpp_<0x120 + N>(v); }
template <int N> __device__ __forceinline__ float row_shr(float v) { return dpp_<0x110 + N>(v); }    // lane i reads lane i - N of its row (0 where there is none)
template <int N> __device__ __forceinline__ double row_shr(double v) { return dpp_<0x110 + N>(v); }
__device__ __forceinline__ float sel(bool m, float a, float b) { return m ? a : b; }
__device__ __forceinline__ double sel(bool m, double a, double b) { return m ? a : b; }
__device__ __forceinline__ int sel(bool m, int a, int b) { return m ? a : b; }   // (by value: a ?: over struct fields would select ADDRESSES and can pin the struct in scratch)
__device__ __forceinline__ void sincos_t(float x, float& s, float& c) { s = __sinf(x); c = __cosf(x); }
__device__ __forceinline__ void sincos_t(double x, double& s, double& c) { s = sin(x); c = cos(x); }
// the action scalings are fp32 arithmetic, left to right, in every build (v2/rl_env_scaledObs.py:125-126)
__device__ __forceinline__ float scale_action_f32(float a, float s1, float s2) { return (a * s1) * s2; }
__device__ __forceinline__ double scale_action_f32(double a, double s1, double s2) { return double((float(a) * float(s1)) * float(s2)); }
__device__ __forceinline__ float joint_cmd_f32(float a, float half, float mid) { return __builtin_fmaf(a, half, mid); }
__device__ __forceinline__ double joint_cmd_f32(double a, double half, double mid) { return double(__builtin_fmaf(float(a), float(half), float(mid))); }

}  // namespace amenv_dev

#include "amenv_team_math.hpp"

namespace amenv_dev {

using TeamParams = TeamParamsT<float>;
using TeamState = TeamStateT<float>;

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one, each XCD has its own L2).  A team wavefront touches
// 64 B of every state group; with the identity mapping the two halves of each 128-B line would be fetched by two different XCDs.
// Blocks that share an XCD therefore take CONSECUTIVE 4-env groups (speed only: any mapping is correct).  gridDim.x is a multiple of 64.
__device__ __forceinline__ int team_group_of_block(int b, int nblocks) { return (b & 7) * (nblocks >> 3) + (b >> 3); }

// ---- one env row's registers, and the per-lane role of a lane in its team -------------------------------------------------------------
template <typename X>
struct TeamEnvT {
  TeamStateT<X> y;                               // dynamic state (component per lane, replicated in the four quads)
  X WP;                                          // waypoint (component per lane)
  X final_yaw, last_distance, ep_return;         // per-env scalars, replicated in all 16 lanes
  int32_t step, counter, flags, episode;
};
using TeamEnv = TeamEnvT<float>;

template <typename X>
struct TeamLaneT {
  int lane, cc, bb;
  bool q0, q1, q2, lead;
  uint32_t offA, offB, offC;                     // observation columns this lane writes in the three row segments
  bool okA, okB, okC;
  X c[((kTeamConsts + 3) / 4) * 4];              // this lane's constants
  __device__ __forceinline__ void init(const TeamParamsT<X>& P) { init(P.consts); }
  __device__ __forceinline__ void init(const void* table) {
    lane = int(threadIdx.x) & 63; cc = lane & 3; bb = (lane >> 2) & 3;
    q0 = bb == 0; q1 = bb == 1; q2 = bb == 2; lead = (lane & 15) == 0;
    // A = [p/10 | v/5 | q | w/5] by quad, B = [(wp - task point)/2 | 0 | yaw/pi | th/pi], C = [thd/5 | tool offset*2]
    // (arithmetic, not nested selects: the compiler turned those into EXEC-masked branches in front of the first load)
    offA = uint32_t(cc + 3 * bb + (bb == 3 ? 1 : 0));                       // 0 + c | 3 + c | 6 + c | 10 + c
    offB = uint32_t(13 + 3 * bb - (bb == 3 ? 2 : 0) + (q2 ? 0 : cc));       // 13 + c | 16 + c | 19 | 20 + c
    offC = uint32_t(23 + cc + (bb > 0 ? 3 : 0));                            // 23 + c | 26 + c
    okA = cc < 3 || q2; okB = q2 ? cc == 0 : cc < 3; okC = cc < 3 && bb < 2;
    constexpr int NC4 = (kTeamConsts + 3) / 4;
    if constexpr (sizeof(X) == 4) {              // float4 pieces [k / 4][lane][k % 4]
      const float4* tab = static_cast<const float4*>(table);
#pragma unroll
      for (int k = 0; k < NC4; k++) {
        const float4 v = tab[k * 16 + (lane & 15)];
        c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
      }
    } else {                                     // [k][lane]
      const X* tab = static_cast<const X*>(table);
#pragma unroll
      for (int k = 0; k < NC4 * 4; k++) c[k] = tab[k * 16 + (lane & 15)];
    }
  }
};
using TeamLane = TeamLaneT<float>;

// state of env i from its tile: lane c of every quad reads component c of each group; the per-env scalars ride in slot 3 of the
// p / v / w groups
template <typename X>
__device__ __forceinline__ void team_load_issue(const char* tile, int i, const TeamLaneT<X>& L, TeamEnvT<X>& E) {   // the loads only
  constexpr uint32_t GB = 64u * 4u * sizeof(X);
  const uint32_t eoff = (uint32_t(i & 63) * 4u + uint32_t(L.cc)) * uint32_t(sizeof(X));
  auto gload = [&](int g) { return *reinterpret_cast<const X*>(tile + kIntBytes + uint32_t(g) * GB + eoff); };
  E.y = TeamStateT<X>{gload(0), gload(1), gload(2), gload(3), gload(5), gload(6)};
  E.WP = gload(4);
  const int4 iv = *(reinterpret_cast<const int4*>(tile) + (i & 63));
  E.step = iv.x; E.counter = iv.y; E.flags = iv.z; E.episode = iv.w;
}
template <typename X>
__device__ __forceinline__ void team_load_unpack(TeamEnvT<X>& E) {   // the per-env scalars out of slot 3 (first use of the loaded data)
  E.final_yaw = bc<3>(E.y.P); E.last_distance = bc<3>(E.y.V); E.ep_return = bc<3>(E.y.W);
}
template <typename X>
__device__ __forceinline__ void team_load(const char* tile, int i, const TeamLaneT<X>& L, TeamEnvT<X>& E) {
  team_load_issue(tile, i, L, E);
  team_load_unpack(E);
}

// Two stores: (1) quad b writes group b (p|yaw, v|last_distance, q, w|return); (2) quad 0 the joint angles, quad 1 the joint rates, quad 2
// the int plane, quad 3 the waypoint group -- per-lane offsets; the step kernel masks quad 3 off between resets (16 B per env-step of write
// traffic; the rollout kernels store once per launch).
template <typename X>
__device__ __forceinline__ void team_store(char* tile, int i, const TeamLaneT<X>& L, const TeamEnvT<X>& E, bool store_wp = true) {
  constexpr uint32_t GB = 64u * 4u * sizeof(X);
  const uint32_t eoff = (uint32_t(i & 63) * 4u + uint32_t(L.cc)) * uint32_t(sizeof(X));
  const bool l3 = L.cc == 3;
  const X g0 = sel(l3, E.final_yaw, E.y.P), g1 = sel(l3, E.last_distance, E.y.V), g3 = sel(l3, E.ep_return, E.y.W);
  const X sv = sel(L.q0, g0, sel(L.q1, g1, sel(L.q2, E.y.Q, g3)));
  *reinterpret_cast<X*>(tile + kIntBytes + uint32_t(L.bb) * GB + eoff) = sv;
  const int ival = sel(L.cc == 0, E.step, sel(L.cc == 1, E.counter, sel(L.cc == 2, E.flags, E.episode)));
  if constexpr (sizeof(X) == 4) {                // one store, per-lane offset (the int plane has the same 16 B per env)
    const float s2 = sel(L.q0, E.y.TH, sel(L.q1, E.y.THD, sel(L.q2, __int_as_float(ival), E.WP)));
    const uint32_t off2 = L.q2 ? eoff : kIntBytes + (L.q0 ? 5u : (L.q1 ? 6u : 4u)) * GB + eoff;
    if (L.q0 || L.q1 || L.q2 || store_wp) *reinterpret_cast<float*>(tile + off2) = s2;   // the waypoint group (quad 3) changes only at a reset (uniform within the row)
  } else {
    if (L.q2) reinterpret_cast<int32_t*>(tile)[(i & 63) * 4 + L.cc] = ival;
    else if (L.q0 || L.q1 || store_wp) *reinterpret_cast<X*>(tile + kIntBytes + (L.q0 ? 5u : (L.q1 ? 6u : 4u)) * GB + eoff) = sel(L.q0, E.y.TH, sel(L.q1, E.y.THD, E.WP));
  }
}

template <typename X> struct TeamOutT { X reward; uint32_t bits; X vA, vB, vC; bool ended; int ep_len; float ep_ret; };
using TeamOut = TeamOutT<float>;

// What a reset leaves in this lane's registers: position / waypoint component, final yaw, and the three observation values of the reset
// state (at rest, level, arm at home).  Lane c < 3 of every quad computes Philox block c; the 12 words are then broadcast inside the quad.
// Pure function of (seed, global env id, episode): the step kernel's helper wave evaluates it while the main wave integrates.
template <typename X> struct TeamResetT { X P, WP, final_yaw, vA, vB, vC; };
template <typename X>
__device__ __forceinline__ TeamResetT<X> team_reset(const TeamParamsT<X>& P, const ColdParams& C, const TeamLaneT<X>& L, int32_t episode, int i) {
  const X* c = L.c;
  uint32_t wds[4];
  const int64_t gid = C.gid0 + i;
  philox4x32_10(C.seed_lo, C.seed_hi, uint32_t(uint64_t(gid)), uint32_t(uint64_t(gid) >> 32), uint32_t(episode), uint32_t(L.cc), wds);
  uint32_t r[12];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    r[k] = uint32_t(qpi<0, 0, 0, 0>(int(wds[k]))); r[4 + k] = uint32_t(qpi<1, 1, 1, 1>(int(wds[k]))); r[8 + k] = uint32_t(qpi<2, 2, 2, 2>(int(wds[k])));
  }
  Env<X, 1> e;
  e.episode = episode;
  reset_from_words<X, 1>(C, 1, e, r);
  const X e0 = c[TC_E0], e1 = c[TC_E1], e2 = c[TC_E2];
  TeamResetT<X> R;
  R.P = fma_(e0, e.px, fma_(e1, e.py, e2 * e.pz));
  R.WP = fma_(e0, e.wp[0][0], fma_(e1, e.wp[0][1], e2 * e.wp[0][2]));
  R.final_yaw = e.final_yaw;
  const X eo = fma_(e0, P.ee_home[0], fma_(e1, P.ee_home[1], e2 * P.ee_home[2]));
  const X zero = X(0), Q = L.cc == 0 ? X(1) : X(0);
  R.vA = (L.q0 ? R.P : (L.q1 ? zero : (L.q2 ? Q : zero))) * c[TC_OBS_A];
  const X tp = P.ee_task != 0 ? R.P + eo : R.P;
  R.vB = (L.q0 ? R.WP - tp : (L.q1 ? zero : (L.q2 ? R.final_yaw : zero))) * c[TC_OBS_B];
  R.vC = (L.q0 ? zero : eo) * c[TC_OBS_C];
  return R;
}

// A reset row's registers from team_reset's values; the rest of WaypointQuadEnv.reset's result is constant (at rest, level, arm at home,
// counters cleared, episode + 1).
template <typename X>
__device__ __forceinline__ void team_apply_reset(const TeamLaneT<X>& L, const TeamResetT<X>& R, TeamEnvT<X>& E, TeamOutT<X>& o) {
  TeamStateT<X>& y = E.y;
  y.P = R.P; y.V = X(0); y.W = X(0); y.TH = X(0); y.THD = X(0);
  y.Q = L.cc == 0 ? X(1) : X(0);
  E.WP = R.WP; E.final_yaw = R.final_yaw;
  E.last_distance = X(-1); E.ep_return = X(0);
  E.step = 0; E.counter = 0; E.flags = 0; E.episode += 1;
  o.vA = R.vA; o.vB = R.vB; o.vC = R.vC;
  o.bits |= AMENV_INFO_WAS_RESET;
}

// observation values of the lane's three row segments for the state in registers (eo = tool offset, world axes)
template <typename X>
__device__ __forceinline__ void team_obs_vals(const TeamParamsT<X>& P, const TeamLaneT<X>& L, const TeamEnvT<X>& E, X eo, X& vA, X& vB, X& vC) {
  const X* c = L.c;
  const TeamStateT<X>& z = E.y;
  vA = sel(L.q0, z.P, sel(L.q1, z.V, sel(L.q2, z.Q, z.W))) * c[TC_OBS_A];
  const X tp = P.ee_task != 0 ? z.P + eo : z.P;
  vB = sel(L.q0, E.WP - tp, sel(L.q1, X(0), sel(L.q2, E.final_yaw, z.TH))) * c[TC_OBS_B];
  vC = sel(L.q0, z.THD, eo) * c[TC_OBS_C];
}

// One control step of one env row, state in registers: mixer -> RK4 -> forward kinematics -> task step -> (episode end: Monitor outputs,
// reset) -> observation values.  act: wrench action a0..a3 (one per lane), actj: joint commands (joint per lane).
// HELPED: the caller's helper wave does the episode-end work (step kernel); this function then stops after the task step and the
// observation values, with o.ended / o.ep_len / o.ep_ret set.
template <int NROT, bool HELPED = false, typename X>
__device__ __forceinline__ TeamOutT<X> team_advance(const TeamParamsT<X>& P, const ColdParams& C, const TeamLaneT<X>& L, TeamEnvT<X>& E, X act, X actj, int i, bool active,
                                                    float* terminal_obs, float* ep_return_out, int32_t* ep_len_out) {
  constexpr int OD = 29;
  TeamStateT<X>& y = E.y;
#ifdef AMENV_TEAM_DIAG_NORK4   // diagnostic build: what the launch costs without the arithmetic (tools/build_variant.py)
  const X EO = y.TH + act * actj;
#else
  const X EO = team_dynamics(P, L.c, L.q0, L.q1, L.q2, y, act, actj);
#endif
  // ---- task step: the one-lane kernels' code on broadcast copies of the state (identical in the 16 lanes of a row)
  Env<X, 1> e;
  e.px = bc<0>(y.P); e.py = bc<1>(y.P); e.pz = bc<2>(y.P);
  e.vx = bc<0>(y.V); e.vy = bc<1>(y.V); e.vz = bc<2>(y.V);
  e.qw = bc<0>(y.Q); e.qx = bc<1>(y.Q); e.qy = bc<2>(y.Q); e.qz = bc<3>(y.Q);
  e.wx = bc<0>(y.W); e.wy = bc<1>(y.W); e.wz = bc<2>(y.W);
  e.wp[0][0] = bc<0>(E.WP); e.wp[0][1] = bc<1>(E.WP); e.wp[0][2] = bc<2>(E.WP);
  e.eox = bc<0>(EO); e.eoy = bc<1>(EO); e.eoz = bc<2>(EO);
#pragma unroll
  for (int k = 0; k < 3; k++) { e.th[k] = X(0); e.thd[k] = X(0); }    // (joints stay in team registers; the task code does not read them)
  e.final_yaw = E.final_yaw; e.last_distance = E.last_distance; e.ep_return = E.ep_return;
  e.step = E.step; e.counter = E.counter; e.flags = E.flags; e.episode = E.episode;
  TeamOutT<X> o;
  o.bits = task_step<X, 1, true>(P, e, o.reward);
  e.ep_return += o.reward;
  E.last_distance = e.last_distance; E.ep_return = e.ep_return;
  E.step = e.step; E.counter = e.counter; E.flags = e.flags;
  team_obs_vals(P, L, E, EO, o.vA, o.vB, o.vC);
  o.ended = (o.bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
  o.ep_len = 0; o.ep_ret = 0.0f;
  if (o.ended) {   // uniform within the row; SB3 DummyVecEnv + Monitor contract
    o.ep_len = e.step; o.ep_ret = float(e.ep_return);
    if constexpr (!HELPED) {
      if (active) {
        if (terminal_obs) {
          const uint32_t row = uint32_t(i) * OD;
          if (L.okA) terminal_obs[row + L.offA] = float(o.vA);
          if (L.okB) terminal_obs[row + L.offB] = float(o.vB);
          if (L.okC) terminal_obs[row + L.offC] = float(o.vC);
        }
        if (L.lead) {
          if (ep_return_out) ep_return_out[i] = o.ep_ret;
          if (ep_len_out) ep_len_out[i] = o.ep_len;
        }
      }
      if (P.flags & AMENV_FLAG_AUTO_RESET) team_apply_reset(L, team_reset(P, C, L, E.episode, i), E, o);
    }
  }
  return o;
}

// per-step outputs; OPT = the caller may pass null pointers (rollout kernels), else all are present (step kernel: no pointer tests)
template <bool OPT, typename X>
__device__ __forceinline__ void team_store_outputs(const TeamLaneT<X>& L, const TeamOutT<X>& o, uint32_t i, bool active, float* __restrict__ obs, X* __restrict__ reward_out,
                                                   uint8_t* __restrict__ done, uint32_t* __restrict__ info) {
  const uint32_t row = i * 29u;
  const bool ob = active && (!OPT || obs != nullptr);
  if (ob && L.okA) obs[row + L.offA] = float(o.vA);
  if (ob && L.okB) obs[row + L.offB] = float(o.vB);
  if (ob && L.okC) obs[row + L.offC] = float(o.vC);
  if (active && L.lead) {
    if (!OPT || reward_out) reward_out[i] = o.reward;
    if (!OPT || done) done[i] = o.ended ? 1 : 0;
    if (!OPT || info) info[i] = o.bits;
  }
}

// "This wave reads these registers here": an empty asm per value.  The helper wave of the step kernel starts with it, so that the loads both
// waves issue in front of the role branch stay there -- the compiler otherwise sinks every load only the main wave consumes into the main
// wave's branch, behind that branch's own (second) batch of kernel-argument loads.
template <typename X> __device__ __forceinline__ void team_touch(X v) {
  if constexpr (sizeof(X) == 4) asm volatile("" ::"v"(v));
  else asm volatile("" ::"v"(__double2loint(v)), "v"(__double2hiint(v)));
}

// "These kernel arguments are read here": an empty asm per wave-uniform value.  The step kernel starts with the arguments its first vector
// loads need, so that they arrive in ONE batch of scalar loads issued at wave start; left to itself the compiler batches arguments by
// basic block, and every batch in front of the first vector load is a full (dependent) trip to the kernel-argument segment.
template <typename A> __device__ __forceinline__ void team_arg(A a) { asm volatile("" ::"s"(a)); }
template <typename A, typename... R> __device__ __forceinline__ void team_arg(A a, R... r) { team_arg(a); team_arg(r...); }

// raw buffer descriptor over [p, p + bytes): a store whose byte offset is >= bytes is dropped by the hardware -- lanes (and rows past the
// last env) are masked by their OFFSET instead of by EXEC, so the store tail of the step kernel has no branches
__device__ __forceinline__ __amdgpu_buffer_rsrc_t team_rsrc(void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, int(bytes), 0x00020000); }
constexpr uint32_t kOob = 0xFFFFFFFFu;

// One control step of 4 envs per main wavefront.  grid = n_tiles * 16 workgroups of 128 threads: wave 0 integrates, wave 1 helps with
// episode ends.  At 4096 envs ~10 of the 1024 main waves see an episode end in every launch, the launch is as slow as its slowest wave,
// and episode-end code (rarely run on any one CU) costs ~7 clocks per instruction: with everything inline those waves ran 2,100 clocks
// (0.9 us of 5.8) longer than the rest (tools/stamp_team.py).  So the helper wave evaluates team_reset while the main wave integrates and
// leaves the values in LDS; after the kernel's ONE barrier the main wave's episode-end path is six LDS reads (+ three terminal-observation
// stores after its regular stores), and the helper writes Monitor's return / length and adds the ended episodes to its replica of the
// totals (owned for the launch when the grid has at most kStatsReplicas workgroups: plain read-modify-write; otherwise atomics).
// Round 3: the helper shares every SIMD with some main wave, so it only spends the reset's ~200 instructions on rows that CAN end their
// episode in this step -- a conservative test on the loaded state (time limit: exact; hold counter at its limit; height / range within one
// step's travel of the crash / bounds thresholds).  A row that ends without having been announced (no such case is known) is reset by the
// main wave itself, so the test only decides who does the work, never the result.  The totals replica is read only when an episode ended.
// The first 8 dwords of the arguments are what the first loads need; the tile size is a constant of this kernel (one waypoint group, two joint groups).
template <typename X, int NROT>
__global__ __launch_bounds__(128) void step_kernel_team(void* __restrict__ blob, int32_t n_envs, int32_t n_blocks, const float* __restrict__ actions,
                                                        const void* __restrict__ lane_consts, float* __restrict__ obs, X* __restrict__ reward_out,
                                                        uint8_t* __restrict__ done, uint32_t* __restrict__ info, const StepTail tl, const ColdParams C) {
  constexpr uint32_t tile_bytes = kIntBytes + 7u * 64u * 4u * uint32_t(sizeof(X));
  // wave-uniform parameters: scalar loads from the device block behind the per-lane table (see team_block_bytes)
  TeamParamsT<X> P;
  {
    typedef const __attribute__((address_space(4))) uint32_t ConstWord;   // constant address space: uniform loads from it are scalar loads
    ConstWord* src = (ConstWord*)(static_cast<const char*>(lane_consts) + team_table_bytes<X>());
    uint32_t* dst = reinterpret_cast<uint32_t*>(&P);
    static_assert(sizeof(TeamParamsT<X>) % 4 == 0, "copied word by word");
#pragma unroll
    for (int k = 0; k < int(sizeof(TeamParamsT<X>) / 4); k++) dst[k] = src[k];
  }
  static_assert(NROT == 6, "team kernel: 6-rotor airframe");
  constexpr int AD = 7, OD = 29;
  __shared__ X rst[6][64];                         // team_reset's values, lane for lane
  __shared__ uint32_t fl[4][4];                    // per row: bit 0 ended on a real env, bit 1 reset | info bits | length | return
  __shared__ uint32_t have[4];                     // per row: the helper has left reset values
  __shared__ unsigned long long acc[S_COUNT];      // this launch's additions to the Monitor totals
#ifdef AMENV_STAMPS
  unsigned long long stamps_[kStampSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const int role = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  team_arg(blob, n_envs, n_blocks, actions, lane_consts);   // what the first loads need: ONE small batch of argument loads, nothing else in front of it
  AMENV_STAMP(0);
  // Both waves of the workgroup start the same way -- constants, state, actions: ONE batch of kernel-argument loads and the vector loads
  // right behind it (with the role branch first, the main wave's arguments arrived in a second, dependent batch: +370 clocks before its
  // first load was issued).  The helper needs the state for its episode-end test anyway; its copies of the lines come from the L1 / L2.
  TeamLaneT<X> L;
  L.init(lane_consts);
  const int row = L.lane >> 4;
  const int i0 = team_group_of_block(int(blockIdx.x), n_blocks) * 4;   // first env of this workgroup (wave-uniform; its 4 envs share a tile)
  const int i = i0 + row;                                                    // env of this row
  const bool active = i < n_envs;
  char* tile = static_cast<char*>(blob) + size_t(i0 >> 6) * tile_bytes;
  const int ia = active ? i : n_envs - 1;                   // rows past the end redo the last env's arithmetic (their outputs are masked)
  TeamEnvT<X> E;
  team_load_issue(tile, i, L, E);
  const float act_f = actions[size_t(ia) * AD + L.cc];                                    // a0..a3: one per lane
  const float actj_f = actions[size_t(ia) * AD + 4 + (L.cc < 3 ? L.cc : 2)];               // joint commands a4..a6
#ifdef AMENV_TEAM_DIAG_NOHELPER   // diagnostic build: the helper wave leaves at once (no episode-end service: timing only)
  if (role == 1) { __syncthreads(); return; }
#endif
  if (role == 1) {
    // (the helper "reads" everything the main wave loaded -- see team_touch -- and drops it at once: its own work needs few registers)
#pragma unroll
    for (int k = 0; k < kTeamConsts; k++)
      if (k < TC_ALLOC0 || k >= TC_GV1) team_touch(L.c[k]);
    team_touch(E.y.Q); team_touch(E.y.W); team_touch(E.y.TH); team_touch(E.y.THD); team_touch(E.WP); team_touch(act_f); team_touch(actj_f);
    const bool owned = n_blocks <= kStatsReplicas;   // wave-uniform
    if (L.lane < S_COUNT) acc[L.lane] = 0ull;
    // Can this row's episode end in this step?  Lane-local tests + one ballot (the helper shares its SIMD with an integrating wave: no DPP,
    // no square roots here): time limit (exact); hold counter at its limit; height within one step's travel of the crash threshold (lane z);
    // some coordinate beyond 10 / sqrt(3) after one step's travel (|p| > 10 needs one).  NaNs compare "may".
    const X dtc = P.h * X(P.substeps), slack = X(0.01) + X(50) * dtc * dtc;   // one step's travel beyond |v| dt: accelerations up to 100 m/s^2
    const X reach = fma_(dtc, abs_(E.y.V), abs_(E.y.P)) + slack;              // lanes 0..2 (lane 3 carries the per-env scalars: excluded below)
    const bool lane_may = (L.cc < 3 && !(reach < X(5.7735))) || (L.cc == 2 && !(E.y.P - dtc * abs_(E.y.V) - slack > X(0.1)));
    const unsigned long long mm = __ballot(lane_may);
#ifdef AMENV_TEAM_DIAG_NOPREDICT   // diagnostic build: the helper never prepares a reset (the main wave computes it inline): timing only
    constexpr bool kPredict = false;
#else
    constexpr bool kPredict = true;
#endif
    const bool may = kPredict && (((mm >> (L.lane & 48)) & 0xFFFFull) != 0ull || E.step >= P.max_steps || ((E.flags & AMENV_FLAGBIT_FWR) && E.counter >= P.counter_limit) ||
                                  (P.flags & AMENV_FLAG_NAN_GUARD));
    if (L.lead) have[row] = may ? 1u : 0u;
    unsigned long long* totals = tl.stats + size_t(blockIdx.x & (kStatsReplicas - 1)) * kStatsStride;
    unsigned long long mine = 0ull;
    bool fetched = false;                           // wave-uniform
    if (__ballot(may) != 0ull) {   // wave-uniform; rows are uniform, DPP stays inside quads
      if (owned && L.lane < S_COUNT) mine = totals[L.lane];   // an episode end is likely: the replica's line, in flight during the barrier
      fetched = true;
      if (may) {
        const TeamResetT<X> R = team_reset(P, C, L, E.episode, i);
        rst[0][L.lane] = R.P; rst[1][L.lane] = R.WP; rst[2][L.lane] = R.final_yaw;
        rst[3][L.lane] = R.vA; rst[4][L.lane] = R.vB; rst[5][L.lane] = R.vC;
      }
    }
    __syncthreads();
#ifdef AMENV_TEAM_DIAG_NOPOST   // diagnostic build: the helper leaves after the barrier (no Monitor service: timing only)
    return;
#endif
    const bool ended = (fl[row][0] & 1u) != 0 && L.lead;   // one lane per ended row
    if (__ballot(ended) != 0ull) {   // wave-uniform
      if (!fetched && owned && L.lane < S_COUNT) mine = totals[L.lane];
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
#ifndef AMENV_TEAM_DIAG_NOPRIO
  __builtin_amdgcn_s_setprio(3);   // every SIMD also holds some workgroup's helper wave: the integrating wave goes first whenever both can issue
#endif
  team_load_unpack(E);
  const X act = X(act_f), actj = X(actj_f);
  // store plumbing, formed while the loads are in flight (pinned there: the scheduler would otherwise sink it into the tail, which the
  // launch waits for): byte offsets into the observation rows, out of range where this lane -- or a row past the last env -- does not store
  const uint32_t obs_bytes = uint32_t(n_envs) * uint32_t(OD * 4);
  const uint32_t rowb = uint32_t(i) * uint32_t(OD * 4);
  uint32_t voA = L.okA ? rowb + L.offA * 4u : kOob, voB = L.okB ? rowb + L.offB * 4u : kOob, voC = L.okC ? rowb + L.offC * 4u : kOob;
  asm volatile("" : "+v"(voA), "+v"(voB), "+v"(voC));
  const __amdgpu_buffer_rsrc_t r_obs = team_rsrc(obs, obs_bytes), r_term = team_rsrc(tl.terminal_obs, tl.terminal_obs ? obs_bytes : 0u);
  AMENV_STAMP(1);          // loads issued
  AMENV_STAMP_DRAIN();
  AMENV_STAMP(2);          // loads landed
  TeamOutT<X> o = team_advance<NROT, true>(P, C, L, E, act, actj, i, active, nullptr, nullptr, nullptr);
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
  const bool any_end = __ballot(o.ended) != 0ull;           // wave-uniform
  const X tA = o.vA, tB = o.vB, tC = o.vC;                  // the terminal observation of a row that ended
  if (any_end && resets) {
    TeamResetT<X> R;
    if (have[row]) R = TeamResetT<X>{rst[0][L.lane], rst[1][L.lane], rst[2][L.lane], rst[3][L.lane], rst[4][L.lane], rst[5][L.lane]};
    else R = team_reset(P, C, L, E.episode, i);             // not announced by the helper's test (never observed): same values, computed here
    team_apply_reset(L, R, E, o);
  }
  AMENV_STAMP(5);          // barrier + reset values
  team_store(tile, i, L, E, resets);
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(float(o.vA)), r_obs, int(voA), 0, 0);
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(float(o.vB)), r_obs, int(voB), 0, 0);
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(float(o.vC)), r_obs, int(voC), 0, 0);
  if (active && L.lead) { reward_out[uint32_t(i)] = o.reward; done[uint32_t(i)] = o.ended ? 1 : 0; info[uint32_t(i)] = o.bits; }
  if (any_end) {           // wave-uniform; rows that did not end store nowhere
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(float(tA)), r_term, int(o.ended ? voA : kOob), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(float(tB)), r_term, int(o.ended ? voB : kOob), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(float(tC)), r_term, int(o.ended ? voC : kOob), 0, 0);
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
