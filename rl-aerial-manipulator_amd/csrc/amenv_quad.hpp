// amenv_quad.hpp -- lane-QUAD step kernel for the rigid vehicles (reference quadrotor, hexacopter: BASELINE configs 0-1) in the latency regime.
//
// The lane-team idea of amenv_team.hpp without the arm: at 4096 envs a one-lane-per-env launch is 64 (+ helper) wavefronts on 1024 SIMDs
// and its length is one lane's instruction stream (~700 VALU instructions on the integrating wave of step_kernel_pw).  Here
//   one env = one DPP QUAD of 4 lanes (x, y, z, spare / quaternion w-z), one wavefront = 16 envs, 4096 envs = 256 wavefronts,
// 3-vectors live with one component per lane, the inertia and its inverse as three column registers with one row per lane, the quaternion
// in the four lanes; cross products and matrix-vector products take the other components through DPP quad_perm operands (helpers of
// amenv_team.hpp).  The RK4 right-hand side (quadcopter.py:66-103) becomes ~45 instructions per stage instead of ~90; reward, state machine
// and reset run redundantly in the 4 lanes of a quad from broadcast copies of the state with the SAME task_step / reset_from_words code as
// the one-lane kernels (every branch is uniform within a quad, which keeps DPP legal inside it).  Episode ends: a helper wavefront per
// workgroup evaluates the reset (Philox + reset_from_words) for all 16 rows while the main wave integrates, owns the workgroup's replica of
// the Monitor totals and writes Monitor's outputs -- the step_kernel_team scheme.
// MEASURED RESULT (MI355X, tools/gpu_quad.sh): no faster than step_kernel_pw -- 2.85 vs 2.86 us at 1024 envs, 2.98 vs 3.00 at 4096, slower from
// 8192 envs on (four times the wavefronts).  The rigid step at these sizes is not bound by an instruction stream: it sits on the dependent-
// launch floor (1.7 us) plus one load -> compute -> store round trip of memory latency.  Kept as an opt-in kernel (AMENV_KERNEL_TEAM on a
// rigid vehicle; AUTO never selects it) because it is the evidence for that statement and shares its tests with the arm's team kernel.
// Same expressions as amenv_model.hpp; sums over components associate differently (trees over lanes), so the result agrees with the
// one-lane kernels to rounding (tests: teacher-forced golden episodes <= 1e-5, tracks the lane kernel <= 2e-6 per step), not bit for bit.
#pragma once
#include "amenv_team.hpp"

namespace amenv_dev {

struct QuadParams {          // wave-uniform (SGPRs)
  float I[6], Iinv[6];       // inertia and its inverse: xx xy xz yy yz zz
  float inv_mass, h;
  float tmin[6], tmax[6];
  int32_t substeps, max_steps, counter_limit, ee_task, K;   // ee_task = 0, K = 1 (the task code reads them)
  uint32_t flags;
  const float4* consts;      // the team constant table (amenv_capi.hip team_const_table): this kernel reads the columns of body 0
};

struct QuadLane {
  int lane, cc;
  bool lead, c3;
  float c[((kTeamConsts + 3) / 4) * 4];
  __device__ __forceinline__ void init(const QuadParams& P) {
    lane = int(threadIdx.x) & 63; cc = lane & 3; lead = cc == 0; c3 = cc == 3;
    constexpr int NC4 = (kTeamConsts + 3) / 4;
#pragma unroll
    for (int k = 0; k < NC4; k++) {
      const float4 v = P.consts[k * 16 + cc];
      c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
    }
  }
};

struct QuadEnv {
  float P, V, Q, W, WP;                          // component per lane (lane 3 of P / V / W carries the group's scalar slot on load only)
  float final_yaw, last_distance, ep_return;     // per-env scalars, replicated in the 4 lanes
  int32_t step, counter, flags, episode;
};

__device__ __forceinline__ void quad_load(const char* tile, int i, const QuadLane& L, QuadEnv& E) {
  const uint32_t eoff = uint32_t(i & 63) * 16u + uint32_t(L.cc) * 4u;
  auto gload = [&](int g) { return *reinterpret_cast<const float*>(tile + kIntBytes + uint32_t(g) * 1024u + eoff); };
  E.P = gload(0); E.V = gload(1); E.Q = gload(2); E.W = gload(3); E.WP = gload(4);
  const int4 iv = *(reinterpret_cast<const int4*>(tile) + (i & 63));
  E.final_yaw = bc<3>(E.P); E.last_distance = bc<3>(E.V); E.ep_return = bc<3>(E.W);
  E.step = iv.x; E.counter = iv.y; E.flags = iv.z; E.episode = iv.w;
}

// every lane writes component c of the four state groups and of the int plane; the waypoint group only after a reset
__device__ __forceinline__ void quad_store(char* tile, int i, const QuadLane& L, const QuadEnv& E, bool was_reset) {
  const uint32_t eoff = uint32_t(i & 63) * 16u + uint32_t(L.cc) * 4u;
  auto gp = [&](int g) { return reinterpret_cast<float*>(tile + kIntBytes + uint32_t(g) * 1024u + eoff); };
  *gp(0) = L.c3 ? E.final_yaw : E.P;
  *gp(1) = L.c3 ? E.last_distance : E.V;
  *gp(2) = E.Q;
  *gp(3) = L.c3 ? E.ep_return : E.W;
  *reinterpret_cast<int32_t*>(tile + eoff) = L.cc == 0 ? E.step : (L.cc == 1 ? E.counter : (L.cc == 2 ? E.flags : E.episode));
  if (was_reset) *gp(4) = L.c3 ? 0.0f : E.WP;
}

struct QuadDeriv { float V, Q, W; };

// quadcopter.py:66-103 in component layout.  Fz = (F / m) e_z (thrust acceleration along body z), Mv = moments, I / J = inertia / inverse.
__device__ __forceinline__ QuadDeriv quad_rhs(const float* c, const TM<float>& I, const TM<float>& J, float Q, float W, float Fz, float Mv) {
  QuadDeriv d;
  const float n2 = sum4(Q * Q);
  const float two_in2 = 2.0f * rcp_(n2);
  const X3<float> qv{qp<1, 2, 3, 3>(Q), qp<2, 3, 1, 3>(Q), qp<3, 1, 2, 3>(Q)};
  const float qw = bc<0>(Q);
  // world acceleration: Rq^T (F/m e_z) + (0, 0, -g) = A + (2/|q|^2) qv x (qv x A - qw A) + g      (:73-75)
  d.V = fma_(two_in2, cross_c(qv, fma_(-qw, Fz, cross_c(qv, Fz))), Fz) + c[TC_GV];
  // quaternion kinematics (4 lanes): -1/2 Omega(w) q + 2 (1 - |q|^2) q                              (:77-82)
  float dq = fma_(-2.0f, n2, 2.0f) * Q;
  dq = fma_(c[TC_SP] * bc<0>(W), qp<1, 0, 3, 2>(Q), dq);
  dq = fma_(c[TC_SQ] * bc<1>(W), qp<2, 3, 0, 1>(Q), dq);
  dq = fma_(c[TC_SR] * bc<2>(W), qp<3, 2, 1, 0>(Q), dq);
  d.Q = dq;
  // body rates: J (M - w x (I w))                                                                    (:86-87)
  const float IW = matvec(I, W);
  d.W = matvec(J, Mv - cross_c(x3(W), IW));
  return d;
}

struct QuadOut { float reward; uint32_t bits; bool ended; int ep_len; float ep_ret; };

// One control step of one env quad, state in registers: mixer -> RK4 -> task step.  The caller handles the episode end.
template <int NROT>
__device__ __forceinline__ QuadOut quad_advance(const QuadParams& P, const QuadLane& L, QuadEnv& E, float act) {
  const float* c = L.c;
  // mixer -> per-rotor clamp -> re-mix (quadcopter.py:109-112); wrench entry per lane; action scaling in fp32, left to right
  const float uu = (act * c[TC_ACT1]) * c[TC_ACT2];
  float wr = 0.0f;
#pragma unroll
  for (int r = 0; r < NROT; r++) {
    float t = sum4(c[TC_ALLOC0 + r] * uu);
    t = clamp_(t, P.tmin[r], P.tmax[r]);
    wr = fma_(c[TC_MIX0 + r], t, wr);
  }
  const float Fz = (bc<0>(wr) * P.inv_mass) * c[TC_E2], Mv = qp<1, 2, 3, 3>(wr);
  const float e0 = c[TC_E0], e1 = c[TC_E1], e2 = c[TC_E2];
  const TM<float> I{fma_(e0, P.I[0], fma_(e1, P.I[1], e2 * P.I[2])), fma_(e0, P.I[1], fma_(e1, P.I[3], e2 * P.I[4])), fma_(e0, P.I[2], fma_(e1, P.I[4], e2 * P.I[5]))};
  const TM<float> J{fma_(e0, P.Iinv[0], fma_(e1, P.Iinv[1], e2 * P.Iinv[2])), fma_(e0, P.Iinv[1], fma_(e1, P.Iinv[3], e2 * P.Iinv[4])),
             fma_(e0, P.Iinv[2], fma_(e1, P.Iinv[4], e2 * P.Iinv[5]))};
  const float h = P.h, hh = 0.5f * h, h6 = h * (1.0f / 6.0f);
  int it = 0;
  do {   // RK4 with a running weighted sum
    QuadDeriv d = quad_rhs(c, I, J, E.Q, E.W, Fz, Mv);
    float aP = E.V, aV = d.V, aQ = d.Q, aW = d.W;
    float sV = fma_(hh, d.V, E.V), sQ = fma_(hh, d.Q, E.Q), sW = fma_(hh, d.W, E.W);
    d = quad_rhs(c, I, J, sQ, sW, Fz, Mv);
    aP = fma_(2.0f, sV, aP); aV = fma_(2.0f, d.V, aV); aQ = fma_(2.0f, d.Q, aQ); aW = fma_(2.0f, d.W, aW);
    sV = fma_(hh, d.V, E.V); sQ = fma_(hh, d.Q, E.Q); sW = fma_(hh, d.W, E.W);
    d = quad_rhs(c, I, J, sQ, sW, Fz, Mv);
    aP = fma_(2.0f, sV, aP); aV = fma_(2.0f, d.V, aV); aQ = fma_(2.0f, d.Q, aQ); aW = fma_(2.0f, d.W, aW);
    sV = fma_(h, d.V, E.V); sQ = fma_(h, d.Q, E.Q); sW = fma_(h, d.W, E.W);
    d = quad_rhs(c, I, J, sQ, sW, Fz, Mv);
    E.P = fma_(h6, aP + sV, E.P); E.V = fma_(h6, aV + d.V, E.V); E.Q = fma_(h6, aQ + d.Q, E.Q); E.W = fma_(h6, aW + d.W, E.W);
  } while (++it < P.substeps);
  E.Q = E.Q * rsqrt_(sum4(E.Q * E.Q));
  // ---- task step: the one-lane kernels' code on broadcast copies of the state (identical in the 4 lanes of a quad)
  Env<float, 1> e;
  e.px = bc<0>(E.P); e.py = bc<1>(E.P); e.pz = bc<2>(E.P);
  e.vx = bc<0>(E.V); e.vy = bc<1>(E.V); e.vz = bc<2>(E.V);
  e.qw = bc<0>(E.Q); e.qx = bc<1>(E.Q); e.qy = bc<2>(E.Q); e.qz = bc<3>(E.Q);
  e.wx = bc<0>(E.W); e.wy = bc<1>(E.W); e.wz = bc<2>(E.W);
  e.wp[0][0] = bc<0>(E.WP); e.wp[0][1] = bc<1>(E.WP); e.wp[0][2] = bc<2>(E.WP);
  e.eox = e.eoy = e.eoz = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; k++) { e.th[k] = 0.0f; e.thd[k] = 0.0f; }
  e.final_yaw = E.final_yaw; e.last_distance = E.last_distance; e.ep_return = E.ep_return;
  e.step = E.step; e.counter = E.counter; e.flags = E.flags; e.episode = E.episode;
  QuadOut o;
  o.bits = task_step<float, 1, false>(P, e, o.reward);
  e.ep_return += o.reward;
  o.ended = (o.bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
  o.ep_len = e.step; o.ep_ret = e.ep_return;
  E.last_distance = e.last_distance; E.ep_return = e.ep_return;
  E.step = e.step; E.counter = e.counter; E.flags = e.flags; E.episode = e.episode;
  return o;
}

// What a reset leaves in this lane's registers: position / waypoint component and the final yaw (the rest of WaypointQuadEnv.reset's result
// is constant: at rest, level, counters cleared, episode + 1).  Lane c < 3 computes Philox block c; the 12 words are broadcast in the quad.
struct QuadReset { float P, WP, final_yaw; };
__device__ __forceinline__ QuadReset quad_reset(const ColdParams& C, const QuadLane& L, int32_t episode, int i) {
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
  return QuadReset{fma_(e0, e.px, fma_(e1, e.py, e2 * e.pz)), fma_(e0, e.wp[0][0], fma_(e1, e.wp[0][1], e2 * e.wp[0][2])), e.final_yaw};
}
__device__ __forceinline__ void quad_apply_reset(const QuadLane& L, const QuadReset& R, QuadEnv& E) {
  E.P = R.P; E.V = 0.0f; E.W = 0.0f; E.Q = L.lead ? 1.0f : 0.0f;
  E.WP = R.WP; E.final_yaw = R.final_yaw;
  E.last_distance = -1.0f; E.ep_return = 0.0f;
  E.step = 0; E.counter = 0; E.flags = 0; E.episode += 1;
}

// observation row of env i (rl_env_scaledObs.py:98-121) from the quad's registers: six dword stores per lane (lane 3: q_z and the yaw entry)
__device__ __forceinline__ void quad_store_obs(const QuadLane& L, const QuadEnv& E, float* __restrict__ row) {
  const bool v3 = !L.c3;
  if (v3) row[L.cc] = E.P * 0.1f;
  if (v3) row[3 + L.cc] = E.V * 0.2f;
  row[6 + L.cc] = E.Q;
  if (v3) row[10 + L.cc] = E.W * 0.2f;
  if (v3) row[13 + L.cc] = (E.WP - E.P) * 0.5f;
  row[16 + L.cc] = L.c3 ? E.final_yaw * 0.31830988618379067154f : 0.0f;
}

// One control step of 16 envs per main wavefront.  Workgroup = 128 threads: wave 0 integrates, wave 1 helps with episode ends (reset values
// of all 16 rows in LDS before the ONE barrier; owner of the workgroup's replica of the Monitor totals; writes Monitor's outputs).
template <int NROT>
__global__ __launch_bounds__(128) void step_kernel_quad(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, const float4* __restrict__ actions,
                                                        float* __restrict__ obs, float* __restrict__ reward_out, uint8_t* __restrict__ done,
                                                        uint32_t* __restrict__ info, const StepTail tl, const ColdParams C, const QuadParams P) {
  constexpr int OD = 20;
  __shared__ float rst[3][64];                     // quad_reset's values, lane for lane
  __shared__ uint32_t fl[16][4];                   // per row: bit 0 ended on a real env, bit 1 reset | info bits | length | return
  __shared__ unsigned long long acc[S_COUNT];      // this launch's additions to the Monitor totals
  const int role = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  QuadLane L;
  L.init(P);
  const int row = L.lane >> 2;
  const int i = int(blockIdx.x) * 16 + row;        // env of this quad (a workgroup's 16 envs are contiguous: one 256-B piece of every group)
  const bool active = i < n_envs;
  char* tile = static_cast<char*>(blob) + size_t(i >> 6) * tile_bytes;
  if (role == 1) {
    const bool owned = gridDim.x <= kStatsReplicas;   // wave-uniform
    unsigned long long* totals = tl.stats + size_t(blockIdx.x & (kStatsReplicas - 1)) * kStatsStride;
    unsigned long long mine = 0ull;
    if (L.lane < S_COUNT) { if (owned) mine = totals[L.lane]; acc[L.lane] = 0ull; }
    const int32_t episode = (reinterpret_cast<const int4*>(tile) + (i & 63))->w;
    const QuadReset R = quad_reset(C, L, episode, i);
    rst[0][L.lane] = R.P; rst[1][L.lane] = R.WP; rst[2][L.lane] = R.final_yaw;
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
  const int ia = active ? i : n_envs - 1;          // rows past the end redo the last env's arithmetic (their outputs are masked)
  QuadEnv E;
  quad_load(tile, i, L, E);
  const float act = reinterpret_cast<const float*>(actions)[size_t(ia) * 4 + L.cc];
  const QuadOut o = quad_advance<NROT>(P, L, E, act);
  const bool resets = o.ended && (P.flags & AMENV_FLAG_AUTO_RESET);
  if (L.lead) {
    fl[row][0] = (o.ended && active ? 1u : 0u) | (resets ? 2u : 0u);
    if (o.ended) { fl[row][1] = o.bits; fl[row][2] = uint32_t(o.ep_len); fl[row][3] = __float_as_uint(o.ep_ret); }
  }
  __syncthreads();
  if (o.ended && active && tl.terminal_obs) quad_store_obs(L, E, tl.terminal_obs + size_t(i) * OD);   // uniform within the quad
  uint32_t bits = o.bits;
  if (resets) {
    quad_apply_reset(L, QuadReset{rst[0][L.lane], rst[1][L.lane], rst[2][L.lane]}, E);
    bits |= AMENV_INFO_WAS_RESET;
  }
  quad_store(tile, i, L, E, resets);
  if (active) {
    quad_store_obs(L, E, obs + size_t(i) * OD);
    if (L.lead) {
      reward_out[i] = o.reward;
      done[i] = o.ended ? 1 : 0;
      info[i] = bits;
    }
  }
}

// n_steps control steps per launch with open-loop actions [T][N][4]; per-step outputs [T][N]... (any of them may be null).  State and
// constants stay in registers across steps.  Same quad_advance as the step kernel: bit-identical trajectories.
template <int NROT>
__global__ __launch_bounds__(64) void rollout_kernel_quad(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, const float4* __restrict__ actions,
                                                          float* __restrict__ obs, float* __restrict__ reward_out, uint8_t* __restrict__ done,
                                                          uint32_t* __restrict__ info, int n_steps, const StepTail tl, const ColdParams C, const QuadParams P) {
  constexpr int OD = 20;
  QuadLane L;
  L.init(P);
  const int i = int(blockIdx.x) * 16 + (L.lane >> 2);
  const bool active = i < n_envs;
  const int ia = active ? i : n_envs - 1;
  char* tile = static_cast<char*>(blob) + size_t(i >> 6) * tile_bytes;
  QuadEnv E;
  quad_load(tile, i, L, E);
  const size_t n = size_t(n_envs);
  const float* ap = reinterpret_cast<const float*>(actions) + size_t(ia) * 4 + L.cc;
  float act = *ap;
  bool any_reset = false;
  for (int t = 0; t < n_steps; t++) {
    const float a_now = act;
    if (t + 1 < n_steps) { ap += n * 4; act = *ap; }        // next step's action: in flight during this step
    const QuadOut o = quad_advance<NROT>(P, L, E, a_now);
    accumulate_stats(tl.stats, int(blockIdx.x), o.bits, active && o.ended && L.lead, o.ep_len, o.ep_ret);
    uint32_t bits = o.bits;
    if (o.ended && (P.flags & AMENV_FLAG_AUTO_RESET)) {     // uniform within the quad
      quad_apply_reset(L, quad_reset(C, L, E.episode, i), E);
      bits |= AMENV_INFO_WAS_RESET;
      any_reset = true;
    }
    const size_t tn = size_t(t) * n;
    if (active) {
      if (obs) quad_store_obs(L, E, obs + (tn + i) * OD);
      if (L.lead) {
        if (reward_out) reward_out[tn + i] = o.reward;
        if (done) done[tn + i] = o.ended ? 1 : 0;
        if (info) info[tn + i] = bits;
      }
    }
  }
  quad_store(tile, i, L, E, any_reset);
}

}  // namespace amenv_dev
