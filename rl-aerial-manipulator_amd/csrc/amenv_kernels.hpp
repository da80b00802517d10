// amenv_kernels.hpp -- gfx950 kernels of the batched waypoint environment.
//
// Data layout in HBM (DESIGN.md "Layout")
//   state blob : tiles of 64 environments, tile t at byte t * tile_bytes:
//                  [int4 plane ][64 envs] {step, counter, flags, episode}             offset 0, 16 B/env
//                  [G float4 groups][64 envs], group g at offset 1024 + g*64*4*sizeof(T); each group is ONE vector quantity
//                  (+ one per-env scalar in its 4th slot), so that both kernel families read it with one instruction:
//                    g0 {px,py,pz, final_yaw}  g1 {vx,vy,vz, last_distance}  g2 {qw,qx,qy,qz}  g3 {wx,wy,wz, ep_return}
//                    g4.. {waypoint k: x,y,z,-} (K groups)   then with an arm {th0,th1,th2,-} {thd0,thd1,thd2,-}
//                * one-lane-per-env kernels: a lane moves its env's group as 16 B -> a wave moves 1 KiB per instruction, coalesced;
//                * lane-team kernels (4 or 16 lanes per env): lane c of a team moves component c as a dword: the team's lanes cover
//                  the 16 B, consecutive envs are contiguous, the per-lane offset is the same for every group.
//                The public amenv_get/set_state view stays [field][N] (enum amenv_float_field); field_slot() is the map.
//   actions    : f32 [N][A]   one float4 per lane (A = 4), coalesced
//   obs        : f32 [N][OD]  row-major for the policy MLP: rows are staged through LDS and written as contiguous 1-KiB wave stores
//                             (lane-team kernels write their components directly)
#pragma once
#include "amenv_model.hpp"
#include "amenv_arm.hpp"

namespace amenv_dev {

// Monitor totals are kept in kStatsReplicas copies (one 128-byte line each, chosen by wave index) and summed by amenv_stats_read: with one
// copy, the ~10 atomics of every wave that finished an episode hit the same addresses and serialise in L2 -- negligible for 64 waves,
// 45 % of the launch at 1 M envs (81.8 -> 45.4 us).
constexpr int kStatsReplicas = 1024, kStatsStride = 16;
enum StatSlot { S_EPISODES = 0, S_TERMINATED, S_TRUNCATED, S_SUCCESS, S_CRASHED, S_OOB, S_NONFINITE, S_LENGTH, S_RETURN_Q10, S_COUNT };

constexpr uint32_t kIntBytes = 64 * 4 * sizeof(int32_t);  // 1024: one int4 per lane

template <typename T> struct alignas(4 * sizeof(T)) Vec4 { T a, b, c, d; };

// groups of a configuration: 4 (rigid-body state + the three per-env scalars) + one per waypoint + two with an arm
__host__ __device__ inline int n_groups_for(int K, int nj) { return 4 + K + (nj > 0 ? 2 : 0); }
__host__ __device__ inline uint32_t tile_bytes_for(int K, int nj, int tsize) {
  return kIntBytes + uint32_t(n_groups_for(K, nj)) * 64u * 4u * uint32_t(tsize);
}
// public field index (enum amenv_float_field; joints after the waypoints) -> 4 * group + slot in the tile
__host__ __device__ inline int field_slot(int f, int K, int nj) {
  if (f < 3) return f;                                  // position           g0.0-2
  if (f < 6) return 4 + (f - 3);                        // velocity           g1.0-2
  if (f < 10) return 8 + (f - 6);                       // quaternion         g2
  if (f < 13) return 12 + (f - 10);                     // body rates         g3.0-2
  if (f == AMENV_F_FINAL_YAW) return 3;                 //                    g0.3
  if (f == AMENV_F_LAST_DISTANCE) return 7;             //                    g1.3
  if (f == AMENV_F_EP_RETURN) return 15;                //                    g3.3
  const int w = f - AMENV_F_WP0;
  if (w < 3 * K) return 4 * (4 + w / 3) + w % 3;        // waypoint k         g(4+k).0-2
  const int j = w - 3 * K, d = nj > 0 ? nj : 1;         // joints: angles then rates
  return 4 * (4 + K + j / d) + j % d;
}

// byte address of the tile that holds env i
__device__ __forceinline__ const char* tile_base(const void* blob, uint32_t tile_bytes, int i) {
  return static_cast<const char*>(blob) + size_t(i >> 6) * tile_bytes;
}
template <typename T> __device__ __forceinline__ Vec4<T>* gptr(char* tile, int lane, int group) {
  return reinterpret_cast<Vec4<T>*>(tile + kIntBytes + size_t(group) * 64 * sizeof(Vec4<T>)) + lane;
}
__device__ __forceinline__ int4* iptr4(char* tile, int lane) { return reinterpret_cast<int4*>(tile) + lane; }
// scalar views (transposes, lazy episode load)
template <typename T> __device__ __forceinline__ T* fptr(char* tile, int lane, int field, int K, int nj) {
  const int gs = field_slot(field, K, nj);
  return reinterpret_cast<T*>(gptr<T>(tile, lane, gs >> 2)) + (gs & 3);
}
__device__ __forceinline__ int32_t* iptr(char* tile, int lane, int field) { return reinterpret_cast<int32_t*>(iptr4(tile, lane)) + field; }

static_assert(AMENV_F_WP0 == 16 && AMENV_F_FINAL_YAW == 13 && AMENV_F_LAST_DISTANCE == 14 && AMENV_F_EP_RETURN == 15, "field_slot() assumes this field order");

template <typename T, int KW, int NJ = 0>
__device__ __forceinline__ void load_env(int K, const char* tile_c, int lane, Env<T, KW>& e) {
  char* tile = const_cast<char*>(tile_c);
  const Vec4<T> g0 = *gptr<T>(tile, lane, 0), g1 = *gptr<T>(tile, lane, 1), g2 = *gptr<T>(tile, lane, 2), g3 = *gptr<T>(tile, lane, 3);
  const int4 iv = *iptr4(tile, lane);
  e.px = g0.a; e.py = g0.b; e.pz = g0.c; e.final_yaw = g0.d;
  e.vx = g1.a; e.vy = g1.b; e.vz = g1.c; e.last_distance = g1.d;
  e.qw = g2.a; e.qx = g2.b; e.qy = g2.c; e.qz = g2.d;
  e.wx = g3.a; e.wy = g3.b; e.wz = g3.c; e.ep_return = g3.d;
#pragma unroll
  for (int k = 0; k < KW; k++) {   // one group per waypoint
    Vec4<T> v{T(0), T(0), T(0), T(0)};
    if (k < K) v = *gptr<T>(tile, lane, 4 + k);
    e.wp[k][0] = v.a; e.wp[k][1] = v.b; e.wp[k][2] = v.c;
  }
#pragma unroll
  for (int k = 0; k < AMENV_MAX_JOINTS; k++) { e.th[k] = T(0); e.thd[k] = T(0); }
  if constexpr (NJ > 0) {          // joint angles, joint rates: the two groups behind the K waypoint groups
    const Vec4<T> a = *gptr<T>(tile, lane, 4 + K), r = *gptr<T>(tile, lane, 5 + K);
    e.th[0] = a.a; e.th[1] = a.b; e.th[2] = a.c;
    e.thd[0] = r.a; e.thd[1] = r.b; e.thd[2] = r.c;
  }
  e.eox = e.eoy = e.eoz = T(0);
  e.step = iv.x; e.counter = iv.y; e.flags = iv.z; e.episode = iv.w;
}

// per-step store: groups 0..3 + the int4 (final_yaw and episode are rewritten with their unchanged values); with an arm the two
// joint groups as well (the waypoint group is not rewritten)
template <typename T, int KW, int NJ = 0>
__device__ __forceinline__ void store_env_step(char* tile, int lane, const Env<T, KW>& e, int K = KW) {   // K: waypoint groups of the configuration (arm: the joint groups sit behind them)
  *gptr<T>(tile, lane, 0) = Vec4<T>{e.px, e.py, e.pz, e.final_yaw};
  *gptr<T>(tile, lane, 1) = Vec4<T>{e.vx, e.vy, e.vz, e.last_distance};
  *gptr<T>(tile, lane, 2) = Vec4<T>{e.qw, e.qx, e.qy, e.qz};
  *gptr<T>(tile, lane, 3) = Vec4<T>{e.wx, e.wy, e.wz, e.ep_return};
  if constexpr (NJ > 0) {
    *gptr<T>(tile, lane, 4 + K) = Vec4<T>{e.th[0], e.th[1], e.th[2], T(0)};
    *gptr<T>(tile, lane, 5 + K) = Vec4<T>{e.thd[0], e.thd[1], e.thd[2], T(0)};
  }
  *iptr4(tile, lane) = make_int4(e.step, e.counter, e.flags, e.episode);
}

// per-episode constants (waypoints), written only by lanes that were reset
template <typename T, int KW>
__device__ __forceinline__ void store_env_episode(int K, char* tile, int lane, const Env<T, KW>& e) {
#pragma unroll
  for (int k = 0; k < KW; k++)
    if (k < K) *gptr<T>(tile, lane, 4 + k) = Vec4<T>{e.wp[k][0], e.wp[k][1], e.wp[k][2], T(0)};
}

// Stage this lane's observation row in LDS.  OD = 20 (80-B rows): five ds_write_b128, the 8 lanes of a group land on banks
// {0,20,8,28,16,4,24,12}+0..3 -> conflict-free.  OD = 17 (68-B rows, not 16-B aligned): dword writes, stride 17 is conflict-free.
template <int OD>
__device__ __forceinline__ void stage_obs(float* lds_row, const float* o) {
  if constexpr (OD % 4 == 0) {
    float4* d = reinterpret_cast<float4*>(lds_row);
#pragma unroll
    for (int j = 0; j < OD / 4; j++) d[j] = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
  } else {
#pragma unroll
    for (int j = 0; j < OD; j++) lds_row[j] = o[j];
  }
}

// Copy the block's staged rows to global memory: consecutive lanes write consecutive float4 (1 KiB per wave instruction).
// The block's region starts 16-B aligned (blockDim.x * OD * 4 bytes per block, blockDim.x a multiple of 64).
template <int OD>
__device__ __forceinline__ void flush_obs(const float* lds, float* __restrict__ obs_block, int rows_valid, int bs, int tid) {
  if (rows_valid <= 0) return;   // a workgroup made only of padding lanes publishes nothing
  const float4* s = reinterpret_cast<const float4*>(lds);
  float4* d = reinterpret_cast<float4*>(obs_block);
  const int nflt = rows_valid * OD, nvec = nflt >> 2;
  constexpr int PASSES = (OD + 3) / 4;
  // all LDS reads first (unconditional: the staging area always holds blockDim.x rows), then the predicated stores
  float4 v[PASSES];
#pragma unroll
  for (int j = 0; j < PASSES; j++) {
    const int f = j * bs + tid;
    v[j] = (4 * f + 3 < bs * OD) ? s[f] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < PASSES; j++) {
    const int f = j * bs + tid;
    if (f < nvec) d[f] = v[j];
  }
  if constexpr (OD % 4 != 0) {  // a partial last block can leave up to 3 floats past the last full float4
    const int t = 4 * nvec + tid;
    if (t < nflt) obs_block[t] = lds[t];
  }
}

// Monitor-style running totals.  Episode ends are rare (usually zero or one lane of a wave per step), so:
// ballot + popcount for the counters, a SCALAR loop over the finished lanes (v_readlane) for the length /
// return sums -- no cross-lane shuffles -- and one no-return atomic per wave and non-zero counter.
// Called right after the state machine so the atomics drain while the wave stores its outputs.
__device__ __forceinline__ void accumulate_stats(unsigned long long* __restrict__ stats_base, int wave_index, uint32_t bits, bool is_done, int ep_len,
                                                 float ep_ret) {
#ifdef AMENV_DIAG_NO_STATS
  return;
#endif
  unsigned long long m_done = __ballot(is_done);
  if (m_done == 0ull) return;  // wave-uniform
  const unsigned long long m_term = __ballot(is_done && (bits & AMENV_INFO_TERMINATED));
  const unsigned long long m_trunc = __ballot(is_done && (bits & AMENV_INFO_TRUNCATED) && !(bits & AMENV_INFO_TERMINATED));
  const unsigned long long m_succ = __ballot(is_done && (bits & AMENV_INFO_SUCCESS));
  const unsigned long long m_crash = __ballot(is_done && (bits & AMENV_INFO_CRASHED));
  const unsigned long long m_oob = __ballot(is_done && (bits & AMENV_INFO_OOB));
  const unsigned long long m_nf = __ballot(is_done && (bits & AMENV_INFO_NONFINITE));
  const int n_done = __popcll(m_done);
  long long len_sum = 0, ret_sum = 0;
  const int ret_bits = __float_as_int(ep_ret);
  while (m_done) {  // scalar loop: one iteration per finished lane, in lane order (deterministic)
    const int l = __builtin_ctzll(m_done);
    m_done &= m_done - 1;
    len_sum += __builtin_amdgcn_readlane(ep_len, l);
    const float r = __int_as_float(__builtin_amdgcn_readlane(ret_bits, l));
    ret_sum += __builtin_isfinite(r) ? (long long)__builtin_rintf(r * 1024.0f) : 0ll;
  }
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* stats = stats_base + size_t(wave_index & (kStatsReplicas - 1)) * kStatsStride;
    atomicAdd(&stats[S_EPISODES], (unsigned long long)n_done);
    if (m_term) atomicAdd(&stats[S_TERMINATED], (unsigned long long)__popcll(m_term));
    if (m_trunc) atomicAdd(&stats[S_TRUNCATED], (unsigned long long)__popcll(m_trunc));
    if (m_succ) atomicAdd(&stats[S_SUCCESS], (unsigned long long)__popcll(m_succ));
    if (m_crash) atomicAdd(&stats[S_CRASHED], (unsigned long long)__popcll(m_crash));
    if (m_oob) atomicAdd(&stats[S_OOB], (unsigned long long)__popcll(m_oob));
    if (m_nf) atomicAdd(&stats[S_NONFINITE], (unsigned long long)__popcll(m_nf));
    atomicAdd(&stats[S_LENGTH], (unsigned long long)len_sum);
    atomicAdd(&stats[S_RETURN_Q10], (unsigned long long)ret_sum);
  }
}

// The same totals from inside a divergent "this lane's episode ended" branch: every ended lane adds its own episode with five no-return
// atomics (integer: the sums do not depend on the order).  No ballots, no scalar loop -- ~30 instructions: the episode-end code of a
// wavefront is rarely executed on any one CU, so its cost is instruction-cache misses and what counts is its size.
__device__ __forceinline__ void accumulate_stats_lane(unsigned long long* __restrict__ stats_base, int wave_index, uint32_t bits, int ep_len, float ep_ret) {
#ifdef AMENV_DIAG_NO_STATS
  return;
#endif
  unsigned long long* stats = stats_base + size_t(wave_index & (kStatsReplicas - 1)) * kStatsStride;
  atomicAdd(&stats[S_EPISODES], 1ull);
  atomicAdd(&stats[(bits & AMENV_INFO_TERMINATED) ? S_TERMINATED : S_TRUNCATED], 1ull);
  // at most one of the four causes is set (task_step / task_step_v1 return or chain them exclusively)
  const int cause = (bits & AMENV_INFO_SUCCESS) ? S_SUCCESS : (bits & AMENV_INFO_CRASHED) ? S_CRASHED : (bits & AMENV_INFO_OOB) ? S_OOB : S_NONFINITE;
  if (bits & (AMENV_INFO_SUCCESS | AMENV_INFO_CRASHED | AMENV_INFO_OOB | AMENV_INFO_NONFINITE)) atomicAdd(&stats[cause], 1ull);
  atomicAdd(&stats[S_LENGTH], (unsigned long long)(long long)ep_len);
  atomicAdd(&stats[S_RETURN_Q10], (unsigned long long)(__builtin_isfinite(ep_ret) ? (long long)__builtin_rintf(ep_ret * 1024.0f) : 0ll));
}

// The step/rollout kernels take what a wave needs before it can issue its first load as LEADING SCALAR arguments (blob, tile_bytes, n,
// actions, obs, reward, done, info = 14 dwords).  (Kernel-argument preloading into SGPRs, -mllvm -amdgpu-kernarg-preload-count=16, was
// measured in round 2 with the state loads moved in front of every s_load wait: no change at 4096 envs, so the library is built without.)
struct StepIO {
  const float4* actions;  // [N] (or [T][N] for rollout)
  float* obs;             // [N][20]
  void* reward;           // [N] T
  uint8_t* done;          // [N]
  uint32_t* info;         // [N]
  float* terminal_obs;    // [N][20] or null
  float* ep_return;       // [N] or null
  int32_t* ep_len;        // [N] or null
  unsigned long long* stats;
};

// Everything one lane does for one control step, state in registers.  Returns info bits;
// `o` holds the observation to publish (post-reset when the env was auto-reset).
template <typename T, int NJ> struct ArmArg { ArmParams<T> p; };
template <typename T> struct ArmArg<T, 0> { int unused; };

template <typename T, int KW>
__device__ __forceinline__ void observe_joints(const Env<T, KW>& e, float* o) {   // arm extension of the v2 observation
#pragma unroll
  for (int k = 0; k < AMENV_MAX_JOINTS; k++) {
    o[20 + k] = float(e.th[k] * T(0.31830988618379067154));
    o[20 + AMENV_MAX_JOINTS + k] = float(e.thd[k] * T(0.2));
  }
  o[26] = float(e.eox * T(2)); o[27] = float(e.eoy * T(2)); o[28] = float(e.eoz * T(2));   // (tool point - base position) / 0.5
}
// forward kinematics of the lane's current state into e.eo* (the joint-axis pattern is wave-uniform)
template <typename T, int KW, bool ZXX_ONLY>
__device__ __forceinline__ void update_tool_offset(const ArmParams<T>& A, Env<T, KW>& e) {
  V3<T> eo;
  if constexpr (ZXX_ONLY) eo = ee_offset_world<AxesZXX>(A, e.qw, e.qx, e.qy, e.qz, e.th);
  else eo = ee_offset_world_any(A, e.qw, e.qx, e.qy, e.qz, e.th);
  e.eox = eo.x; e.eoy = eo.y; e.eoz = eo.z;
}

// Observation of a freshly reset env of the single-waypoint v2 task, written out: at rest, level, arm at home, one waypoint -- the same
// values the general `observe` produces (its products with the zero velocities / rates are +0), ~12 instructions instead of ~130.
template <typename T, int KW, bool EE, int NJ>
__device__ __forceinline__ void observe_reset(const Env<T, KW>& e, bool ee_task, float* o) {
  static_assert(KW == 1, "single-waypoint task");
  const T c10 = T(0.1), c2 = T(0.5);
  o[0] = float(e.px * c10); o[1] = float(e.py * c10); o[2] = float(e.pz * c10);
  o[3] = o[4] = o[5] = 0.0f;
  o[6] = 1.0f; o[7] = o[8] = o[9] = 0.0f;
  o[10] = o[11] = o[12] = 0.0f;
  T tx, ty, tz;
  task_point<EE>(e, ee_task, tx, ty, tz);
  o[13] = float((e.wp[0][0] - tx) * c2); o[14] = float((e.wp[0][1] - ty) * c2); o[15] = float((e.wp[0][2] - tz) * c2);
  o[16] = o[17] = o[18] = 0.0f;
  o[19] = float(e.final_yaw * T(0.31830988618379067154));
  if constexpr (NJ > 0) {
#pragma unroll
    for (int k = 0; k < 2 * AMENV_MAX_JOINTS; k++) o[20 + k] = 0.0f;
    o[26] = float(e.eox * T(2)); o[27] = float(e.eoy * T(2)); o[28] = float(e.eoz * T(2));
  }
}

template <typename T, int NROT, int KW, int VAR, int NJ, int ROLE = 0, typename X = NoXchg>
__device__ __forceinline__ uint32_t step_lane(const HotParams<T, NROT>& P, const ColdParams& C, const ArmArg<T, NJ>& AA, Env<T, KW>& e, const float* act, int i,
                                              bool active, T& reward, float* o, const StepIO& io, char* tile, int lane,
                                              bool have_episode, bool& was_reset, int& ep_len_out, float& ep_ret_out, const X& x = X{}) {
  constexpr int OD = ObsDim<VAR, NJ>::value;
  const int K = KW == 1 ? 1 : P.K;
  if constexpr (NJ > 0) {
    if constexpr (ROLE == ARM_ROLE_MAIN || ROLE == ARM_ROLE_HELPER) dynamics_arm<T, NROT, KW, AxesZXX, ROLE, X>(P, AA.p, e, act, x);   // two-wave kernel: z,x,x arm only
    else if constexpr (ROLE == ARM_ROLE_STAGED) dynamics_arm_staged<T, NROT, KW>(P, AA.p, e, act, x);                                   // stage-wave kernel: z,x,x arm only
    else if (AA.p.generic_axes) dynamics_arm<T, NROT, KW, AxesAny>(P, AA.p, e, act);   // wave-uniform: one of the two bodies runs
    else dynamics_arm<T, NROT, KW, AxesZXX>(P, AA.p, e, act);
  } else { dynamics<T, NROT, KW>(P, e, act[0], act[1], act[2], act[3]); }
  constexpr bool EE = NJ > 0;   // arm: forward kinematics of the post-step state feed the task point and the observation
  if constexpr (EE) update_tool_offset<T, KW, ROLE == ARM_ROLE_MAIN || ROLE == ARM_ROLE_HELPER || ROLE == ARM_ROLE_STAGED>(AA.p, e);
  const bool ee_task = EE && P.ee_task != 0;
  uint32_t bits;
  if constexpr (VAR == VAR_V1) { bits = task_step_v1<T, KW>(P, e, reward); } else { bits = task_step<T, KW, EE>(P, e, reward); }
  e.ep_return += reward;
  // two-wave kernel: the helper wave computes and publishes the observation of every lane; this (main) wave forms one only in the
  // cold path below (terminal observation / post-reset observation of the lanes whose episode ended)
  constexpr bool kColdElsewhere = ROLE == ARM_ROLE_FLAGS;   // helper waves write terminal / post-reset rows and the reset state
  constexpr bool kLazyObs = ROLE == ARM_ROLE_MAIN || kColdElsewhere;
  auto obs_now = [&]() {
    if constexpr (VAR == VAR_V1) { observe_v1<T, KW>(P.raw_obs != 0, e, o); } else { observe<T, KW, EE>(K, e, o, ee_task); }
    if constexpr (NJ > 0) observe_joints<T, KW>(e, o);
  };
  if constexpr (!kLazyObs) obs_now();
  was_reset = false;
  ep_len_out = 0; ep_ret_out = 0.0f;
  const bool ended = (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
  const bool resets = ended && (P.flags & AMENV_FLAG_AUTO_RESET);
  constexpr bool kWordsFromLds = ROLE == ARM_ROLE_MAIN || ROLE == ARM_ROLE_WORDS || ROLE == ARM_ROLE_STAGED;
  if constexpr (kWordsFromLds) x.sync();   // two-wave kernels: the helper has left every lane's 12 reset words in LDS (every step: unconditional barrier)
  if (__ballot(ended) != 0ull) {  // wave-uniform: the whole cold path is skipped by waves with no episode end
    uint32_t r[12];
    if constexpr (kColdElsewhere) {
    } else if constexpr (kWordsFromLds) {
#pragma unroll
      for (int k = 0; k < 12; k++) r[k] = x.words[k * 64 + lane];
    } else {
      reset_words_wave(C, resets, C.gid0 + i, e.episode, r);  // all lanes take part, lanes 0..2 do the work
    }
    if (ended) {  // SB3 DummyVecEnv + Monitor contract
      ep_len_out = e.step; ep_ret_out = float(e.ep_return);
      if (active) {
        if (!kLazyObs && io.terminal_obs) {   // two-wave kernel: the terminal observation is the row the helper wave staged; copied after the barrier
          float* t = io.terminal_obs + size_t(i) * OD;
          if constexpr (OD % 4 == 0) {
#pragma unroll
            for (int j = 0; j < OD / 4; j++) reinterpret_cast<float4*>(t)[j] = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
          } else {
#pragma unroll
            for (int j = 0; j < OD; j++) t[j] = o[j];
          }
        }
        if constexpr (!kColdElsewhere) {
          if (io.ep_return) io.ep_return[i] = ep_ret_out;
          if (io.ep_len) io.ep_len[i] = ep_len_out;
        }
      }
      if (resets) {
        if constexpr (kColdElsewhere) { (void)r; }
        else if constexpr (VAR == VAR_V1) { reset_from_words_v1<T, KW>(K, e, r); observe_v1<T, KW>(P.raw_obs != 0, e, o); }
        else {
          reset_from_words<T, KW>(C, K, e, r);
          if constexpr (EE) { e.eox = AA.p.ee_home[0]; e.eoy = AA.p.ee_home[1]; e.eoz = AA.p.ee_home[2]; }   // level, arm at home
          if constexpr (KW == 1) {
            observe_reset<T, KW, EE, NJ>(e, ee_task, o);
          } else {
            observe<T, KW, EE>(K, e, o, ee_task);
            if constexpr (NJ > 0) observe_joints<T, KW>(e, o);
          }
        }
        if constexpr (NJ > 0 && VAR == VAR_V1 && !kColdElsewhere) observe_joints<T, KW>(e, o);
        bits |= AMENV_INFO_WAS_RESET;
        was_reset = true;
      }
    }
  }
  return bits;
}

// Workgroup size is a launch parameter (64..256, multiple of 64): LDS staging area = blockDim.x rows.
// The state blob holds whole tiles, so padding lanes (i >= n) of the last wave run like real
// environments on their own (valid) slots; only their outputs are masked.
// Diagnostic build only (-DAMENV_STAMPS, tools/stamp_profile.py): s_memtime stamps at phase boundaries,
// written by lane 0 of each wave into a buffer of their own behind the stats words.  No stamp executes
// in the product build and no output value is computed from one.
#ifdef AMENV_STAMPS
#define AMENV_STAMP(k)                                                                       \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    unsigned long long t_;                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
    stamps_[k] = t_;                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
#define AMENV_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define AMENV_STAMP(k)
#define AMENV_STAMP_DRAIN()
#endif
constexpr int kStampSlots = 8, kStampWaves = 64, kStampBase = kStatsReplicas * kStatsStride;

struct StepTail { float* terminal_obs; float* ep_return; int32_t* ep_len; unsigned long long* stats; };
struct Head { void* blob; uint32_t tile_bytes; int32_t n; };

#ifndef AMENV_STEP_WAVES_ATTR
#define AMENV_STEP_WAVES_ATTR
#endif
template <typename T, int NROT, int KW, int VAR, int NJ>
__global__ __launch_bounds__(256) AMENV_STEP_WAVES_ATTR void step_kernel(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, const float4* __restrict__ actions,
                                                   float* __restrict__ obs, void* __restrict__ reward_out, uint8_t* __restrict__ done,
                                                   uint32_t* __restrict__ info, const StepTail tl, const HotParams<T, NROT> P, const ColdParams C,
                                                   const ArmArg<T, NJ> AA) {
  constexpr int OD = ObsDim<VAR, NJ>::value, AD = kActDim + NJ;
#ifdef AMENV_STAMPS
  unsigned long long stamps_[kStampSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  AMENV_STAMP(0);
  const Head hd{blob, tile_bytes, n_envs};
  const StepIO io{actions, obs, reward_out, done, info, tl.terminal_obs, tl.ep_return, tl.ep_len, tl.stats};
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int BS = int(blockDim.x);
  const int i = blockIdx.x * BS + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool active = i < hd.n;
  char* tile = const_cast<char*>(tile_base(hd.blob, hd.tile_bytes, i));
  const int K = KW == 1 ? 1 : P.K;
  Env<T, KW> e;
  load_env<T, KW, NJ>(K, tile, lane, e);
  float act[AD];
  if constexpr (NJ == 0) {
    const float4 a = io.actions[min(i, hd.n - 1)];  // padding lanes re-read the last env's action: no exec branch in the prologue
    act[0] = a.x; act[1] = a.y; act[2] = a.z; act[3] = a.w;
  } else {
    const float* ap = reinterpret_cast<const float*>(io.actions) + size_t(min(i, hd.n - 1)) * AD;   // 28-B rows: dword loads
#pragma unroll
    for (int j = 0; j < AD; j++) act[j] = ap[j];
  }
  AMENV_STAMP(1);          // loads issued
  AMENV_STAMP_DRAIN();
  AMENV_STAMP(2);          // loads landed
  T reward; float o[kObsDimMax]; bool was_reset; int ep_len; float ep_ret;
  uint32_t bits = step_lane<T, NROT, KW, VAR, NJ>(P, C, AA, e, act, i, active, reward, o, io, tile, lane, false, was_reset, ep_len, ep_ret);
  const bool is_done = active && (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
  AMENV_STAMP(3);          // dynamics + task + obs computed
  accumulate_stats(io.stats, int((blockIdx.x * blockDim.x + threadIdx.x) >> 6), bits, is_done, ep_len, ep_ret);
  store_env_step<T, KW, NJ>(tile, lane, e, K);
  if (was_reset) store_env_episode<T, KW>(K, tile, lane, e);
  if (active) {
    reinterpret_cast<T*>(io.reward)[i] = reward;
    io.done[i] = is_done ? 1 : 0;
    io.info[i] = bits;
  }
  AMENV_STAMP(4);          // state/outputs stores issued
#ifdef AMENV_DIAG_DIRECT_OBS
  if (active) {
    float4* d = reinterpret_cast<float4*>(io.obs + size_t(i) * OD);
#pragma unroll
    for (int j = 0; j < 5; j++) d[j] = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
  }
  (void)lds;
#else
  stage_obs<OD>(lds + threadIdx.x * OD, o);
  __syncthreads();
  const int row0 = blockIdx.x * BS;
  const int rows = min(BS, hd.n - row0);
  flush_obs<OD>(lds, io.obs + size_t(row0) * OD, rows, BS, int(threadIdx.x));
#endif
  AMENV_STAMP(5);          // obs flushed
  AMENV_STAMP(6);
#ifdef AMENV_STAMPS
  AMENV_STAMP_DRAIN();
  AMENV_STAMP(7);          // all stores acknowledged
  const int wave = (blockIdx.x * BS + threadIdx.x) >> 6;
  if (lane == 0 && wave < kStampWaves)
    for (int k = 0; k < kStampSlots; k++) io.stats[kStampBase + wave * kStampSlots + k] = stamps_[k];
#endif
}

// Rigid vehicles at small batches: helper waves per 64-env tile take the episode-end work off the wave that integrates.  With 4096 envs
// some tile ends an episode in almost every launch and the launch is as slow as its slowest wave (hover-only actions, which end no
// episode, ran 1.1 us faster than the bench workload with a one-wave kernel).
//  * KW > 1 or the v1 tasks (128 threads): wave 1 computes the 12 Philox words of a reset for every lane -- they depend only on (seed,
//    env id, episode) -- while the main wave integrates; the main wave's cold path starts from the words in LDS.
//  * single-waypoint v2 task (256 threads; the observation is a pure function of the post-step state).  Episode-end code runs rarely on
//    any one CU and costs ~7 clocks per instruction there, so it is cut into pieces that run side by side after the ONE barrier:
//      wave 3 loads the tile's replica of the running totals (it owns that replica for the launch); after the barrier it sums the
//             contributions of the lanes that ended in LDS and stores the replica back: no global atomics (five same-line atomics took
//             ~670 clocks to drain);
//      wave 2 integrates the same step only to compute, stage and flush the observation rows (as the arm kernel's helper does) and keeps
//             its rows in registers; after the barrier it writes the terminal-observation rows and Monitor's return / length of the
//             lanes that ended;
//      wave 1 computes, for EVERY lane, the words, the reset state and its observation, and leaves the reset position / final yaw in
//             LDS; after the barrier it writes waypoint and observation row of the lanes that were reset;
//      wave 0 (main) integrates, runs the task step, publishes a flag word per lane (and info bits / length / return of the lanes that
//             ended), passes the barrier, takes the reset position of reset lanes from LDS and stores the state of every lane.
template <typename T, int NROT, int KW, int VAR>
__global__ __launch_bounds__(256) void step_kernel_pw(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, const float4* __restrict__ actions,
                                                      float* __restrict__ obs, void* __restrict__ reward_out, uint8_t* __restrict__ done,
                                                      uint32_t* __restrict__ info, const StepTail tl, const HotParams<T, NROT> P, const ColdParams C) {
  constexpr int OD = ObsDim<VAR, 0>::value;
  constexpr bool kObsWave = KW == 1 && VAR == VAR_V2;              // launched with 256 threads then, else with 128
  const Head hd{blob, tile_bytes, n_envs};
  const StepIO io{actions, obs, reward_out, done, info, tl.terminal_obs, tl.ep_return, tl.ep_len, tl.stats};
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [64 rows x OD] obs staging | [12][64] reset words (or flag words + reset positions)
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);   // 0 main, 1 reset, 2 observation, 3 Monitor
  const int i = blockIdx.x * 64 + lane;
  const bool active = i < hd.n;
  char* tile = const_cast<char*>(tile_base(hd.blob, hd.tile_bytes, i));
  uint32_t* words = reinterpret_cast<uint32_t*>(lds + 64 * OD);
  const int row0 = blockIdx.x * 64;
  const ArmArg<T, 0> AA{0};
  if constexpr (kObsWave) {
    static_assert(OD % 4 == 0, "observation rows are written as float4");
#ifdef AMENV_STAMPS   // diagnostic build: stamps of all four waves of the first 16 tiles (wave slot = 4 * tile + role)
    unsigned long long stamps_[kStampSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamps_out = [&]() {
      AMENV_STAMP_DRAIN();
      AMENV_STAMP(6);
      const int w = int(blockIdx.x) * 4 + role;
      if (lane == 0 && w < kStampWaves)
        for (int k = 0; k < kStampSlots; k++) io.stats[kStampBase + w * kStampSlots + k] = stamps_[k];
    };
#endif
    AMENV_STAMP(0);
    // [64] bit 0: episode ended on a real env, bit 1: the lane is reset | [64] info bits | [64] length | [64] return | [4][64] reset position, final yaw
    uint32_t* flag = words;
    float* rst = reinterpret_cast<float*>(words + 256);
    if (role == 3) {                 // Monitor wave: owns totals replica [tile] during this launch -- loads it now, no atomics later
      unsigned long long* totals = io.stats + size_t(blockIdx.x & (kStatsReplicas - 1)) * kStatsStride;
      unsigned long long* acc = reinterpret_cast<unsigned long long*>(words + 512);   // [S_COUNT] this launch's additions (LDS)
      unsigned long long mine = 0ull;
      if (lane < S_COUNT) { mine = totals[lane]; acc[lane] = 0ull; }
      AMENV_STAMP(1); AMENV_STAMP(2); AMENV_STAMP(3);
      __syncthreads();
      AMENV_STAMP(4);
      const bool ended = (flag[lane] & 1u) != 0;
      if (__ballot(ended) != 0ull) {   // wave-uniform
        if (ended) {
          const uint32_t bits = flag[64 + lane];
          const int ep_len = int(flag[128 + lane]);
          const float ep_ret = __uint_as_float(flag[192 + lane]);
          accumulate_stats_lane(acc, 0, bits, ep_len, ep_ret);   // LDS adds (integers: the order does not matter)
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < S_COUNT) totals[lane] = mine + acc[lane];
      }
      AMENV_STAMP(5);
#ifdef AMENV_STAMPS
      stamps_out();
#endif
      return;
    }
    if (role == 1) {
      Env<T, KW> er;
      er.episode = iptr4(tile, lane)->w;
      uint32_t r[12];
      reset_words_serial(C, C.gid0 + i, er.episode, r);
      AMENV_STAMP(1);
      reset_from_words<T, KW>(C, 1, er, r);
      float ro[kObsDimMax];
      observe_reset<T, KW, false, 0>(er, false, ro);
      rst[lane] = float(er.px); rst[64 + lane] = float(er.py); rst[128 + lane] = float(er.pz); rst[192 + lane] = float(er.final_yaw);
      AMENV_STAMP(2); AMENV_STAMP(3);
      __syncthreads();               // flags published (and this wave's reset positions)
      AMENV_STAMP(4);
      if (flag[lane] & 2u) {         // (padding lanes run real arithmetic on their own slots: their state is reset too, rows are not written)
        store_env_episode<T, KW>(1, tile, lane, er);
        if (active) {
          float4* d = reinterpret_cast<float4*>(io.obs + size_t(i) * OD);
#pragma unroll
          for (int j = 0; j < OD / 4; j++) d[j] = make_float4(ro[4 * j], ro[4 * j + 1], ro[4 * j + 2], ro[4 * j + 3]);
        }
      }
      AMENV_STAMP(5);
#ifdef AMENV_STAMPS
      stamps_out();
#endif
      return;
    }
    Env<T, KW> e;
    load_env<T, KW, 0>(1, tile, lane, e);
    float act[kActDim];
    const float4 a = io.actions[min(i, hd.n - 1)];
    act[0] = a.x; act[1] = a.y; act[2] = a.z; act[3] = a.w;
    if (role == 2) {
#ifdef AMENV_STAMPS
      AMENV_STAMP_DRAIN();
#endif
      AMENV_STAMP(1);
      dynamics<T, NROT, KW>(P, e, act[0], act[1], act[2], act[3]);
      AMENV_STAMP(2);
      float ho[kObsDimMax];
      observe<T, KW>(1, e, ho);
      stage_obs<OD>(lds + lane * OD, ho);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      flush_obs<OD>(lds, io.obs + size_t(row0) * OD, min(64, hd.n - row0), 64, lane);
      AMENV_STAMP(3);
      __syncthreads();               // rows stored and acknowledged (wave 1 may now overwrite those of reset lanes); flags published
      AMENV_STAMP(4);
      if (flag[lane] & 1u) {
        if (io.terminal_obs) {
          float4* t = reinterpret_cast<float4*>(io.terminal_obs + size_t(i) * OD);
#pragma unroll
          for (int j = 0; j < OD / 4; j++) t[j] = make_float4(ho[4 * j], ho[4 * j + 1], ho[4 * j + 2], ho[4 * j + 3]);
        }
        if (io.ep_return) io.ep_return[i] = __uint_as_float(flag[192 + lane]);
        if (io.ep_len) io.ep_len[i] = int(flag[128 + lane]);
      }
      AMENV_STAMP(5);
#ifdef AMENV_STAMPS
      stamps_out();
#endif
      return;
    }
#ifdef AMENV_STAMPS
    AMENV_STAMP_DRAIN();
#endif
    AMENV_STAMP(1);
    T reward; float o[kObsDimMax]; bool was_reset; int ep_len; float ep_ret;
    uint32_t bits = step_lane<T, NROT, KW, VAR, 0, ARM_ROLE_FLAGS>(P, C, AA, e, act, i, active, reward, o, io, tile, lane, false, was_reset, ep_len, ep_ret);
    AMENV_STAMP(2);
    const bool is_done = active && (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
    flag[lane] = (is_done ? 1u : 0u) | (was_reset ? 2u : 0u);
    if (is_done) { flag[64 + lane] = bits; flag[128 + lane] = uint32_t(ep_len); flag[192 + lane] = __float_as_uint(ep_ret); }
    AMENV_STAMP(3);
    __syncthreads();
    AMENV_STAMP(4);
    if (was_reset) {                 // the rest of reset_from_words' result is constant: at rest, level, counters cleared
      e.px = T(rst[lane]); e.py = T(rst[64 + lane]); e.pz = T(rst[128 + lane]); e.final_yaw = T(rst[192 + lane]);
      e.vx = e.vy = e.vz = T(0); e.qw = T(1); e.qx = e.qy = e.qz = T(0); e.wx = e.wy = e.wz = T(0);
      e.last_distance = T(-1); e.ep_return = T(0);
      e.step = 0; e.counter = 0; e.flags = 0; e.episode += 1;
    }
    store_env_step<T, KW>(tile, lane, e);
    if (active) {
      reinterpret_cast<T*>(io.reward)[i] = reward;
      io.done[i] = is_done ? 1 : 0;
      io.info[i] = bits;
    }
    AMENV_STAMP(5);
#ifdef AMENV_STAMPS
    stamps_[7] = __ballot(is_done) != 0ull ? 1ull : 0ull;   // did this tile see an episode end?
    stamps_out();
#endif
  } else {
    if (role == 1) {
      const int32_t episode = iptr4(tile, lane)->w;
      uint32_t r[12];
      reset_words_serial(C, C.gid0 + i, episode, r);
#pragma unroll
      for (int k = 0; k < 12; k++) words[k * 64 + lane] = r[k];
      __syncthreads();   // words published (pairs with the barrier in step_lane)
      __syncthreads();   // pairs with the main wave's last barrier
      return;
    }
    const int K = KW == 1 ? 1 : P.K;
    Env<T, KW> e;
    load_env<T, KW, 0>(K, tile, lane, e);
    float act[kActDim];
    const float4 a = io.actions[min(i, hd.n - 1)];
    act[0] = a.x; act[1] = a.y; act[2] = a.z; act[3] = a.w;
    const LdsXchg x{lds, lane, words};
    T reward; float o[kObsDimMax]; bool was_reset; int ep_len; float ep_ret;
    uint32_t bits = step_lane<T, NROT, KW, VAR, 0, ARM_ROLE_WORDS, LdsXchg>(P, C, AA, e, act, i, active, reward, o, io, tile, lane, false, was_reset, ep_len,
                                                                           ep_ret, x);
    const bool is_done = active && (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
    accumulate_stats(io.stats, int(blockIdx.x), bits, is_done, ep_len, ep_ret);
    store_env_step<T, KW>(tile, lane, e);
    if (was_reset) store_env_episode<T, KW>(K, tile, lane, e);
    if (active) {
      reinterpret_cast<T*>(io.reward)[i] = reward;
      io.done[i] = is_done ? 1 : 0;
      io.info[i] = bits;
    }
    stage_obs<OD>(lds + lane * OD, o);
    __syncthreads();
    flush_obs<OD>(lds, io.obs + size_t(row0) * OD, min(64, hd.n - row0), 64, lane);
  }
}

// Two-wave step kernel for the hexacopter + z,x,x arm at small batches (BASELINE config 3 at 4096 envs = 64 tiles).  One tile of 64
// environments per 128-thread workgroup: wave 0 ("main") does everything the one-wave kernel does except link 3's part of each RHS,
// wave 1 ("helper") runs the same RK4 on the same state and contributes link 3 (see amenv_arm.hpp).  A launch of 64 lone waves
// is bound by one wave's instruction issue (~4.8 cycles per dependent VALU op), so halving the longest per-wave instruction
// stream is what shortens the step; at large batches the one-wave kernel (no redundant work) is used instead.
// Barriers: both waves execute exactly 8 * substeps + 1 of them, all in uniform control flow.
template <typename T, int NROT>
__global__ __launch_bounds__(128) void step_kernel_arm2w(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, const float4* __restrict__ actions,
                                                         float* __restrict__ obs, void* __restrict__ reward_out, uint8_t* __restrict__ done,
                                                         uint32_t* __restrict__ info, const StepTail tl, const HotParams<T, NROT> P, const ColdParams C,
                                                         const ArmArg<T, 3> AA) {
  constexpr int KW = 1, VAR = VAR_V2, NJ = 3;
  constexpr int OD = ObsDim<VAR, NJ>::value, AD = kActDim + NJ;
#ifdef AMENV_STAMPS
  unsigned long long stamps_[kStampSlots] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  AMENV_STAMP(0);
  const Head hd{blob, tile_bytes, n_envs};
  const StepIO io{actions, obs, reward_out, done, info, tl.terminal_obs, tl.ep_return, tl.ep_len, tl.stats};
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [64 rows x OD] obs staging, then [kArmXchgSlots][64] exchange
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);   // wave-uniform: 0 main, 1 helper
  const int i = blockIdx.x * 64 + lane;
  const bool active = i < hd.n;
  char* tile = const_cast<char*>(tile_base(hd.blob, hd.tile_bytes, i));
  const uint32_t* words_ = reinterpret_cast<const uint32_t*>(lds + 64 * OD) + kArmXchgSlots * 64;
#ifdef AMENV_STAMPS
  const LdsXchg x{lds + 64 * OD, lane, words_, stamps_};
#else
  const LdsXchg x{lds + 64 * OD, lane, words_};
#endif
  Env<T, KW> e;
  load_env<T, KW, NJ>(1, tile, lane, e);
  float act[AD];
  const float* ap = reinterpret_cast<const float*>(io.actions) + size_t(min(i, hd.n - 1)) * AD;
#pragma unroll
  for (int j = 0; j < AD; j++) act[j] = ap[j];
  AMENV_STAMP(1);          // loads issued
  if (role == 0) { AMENV_STAMP_DRAIN(); }
  AMENV_STAMP(2);          // (main) loads landed
  if (role != 0) {
    // helper: link 3's share of the RK4, then the observation of every lane (a pure function of the post-step state for the
    // single-waypoint task), staged in LDS and flushed coalesced -- all off the main wave's critical path
    // The 12 Philox words a reset of this lane's env would consume (key: seed, global env id, episode) are known from the start of the
    // launch: the helper computes them for EVERY lane in its idle windows inside the RK4 (block 0 before the first barrier of the step,
    // blocks 1-2 during the main wave's last solve), so the main wave never runs Philox on its cold path.
    uint32_t rw[12];
    const int64_t gid = C.gid0 + i;
    const int32_t ep = e.episode;
    auto philox_blocks = [&](int part) {
      const uint32_t g_lo = uint32_t(uint64_t(gid)), g_hi = uint32_t(uint64_t(gid) >> 32);
      if (part == 0) philox4x32_10(C.seed_lo, C.seed_hi, g_lo, g_hi, uint32_t(ep), 0u, &rw[0]);
      else { philox4x32_10(C.seed_lo, C.seed_hi, g_lo, g_hi, uint32_t(ep), 1u, &rw[4]); philox4x32_10(C.seed_lo, C.seed_hi, g_lo, g_hi, uint32_t(ep), 2u, &rw[8]); }
    };
    dynamics_arm<T, NROT, KW, AxesZXX, ARM_ROLE_HELPER, LdsXchg>(P, AA.p, e, act, x, philox_blocks);
    {
      uint32_t* wl = reinterpret_cast<uint32_t*>(lds + 64 * OD) + kArmXchgSlots * 64 + lane;
#pragma unroll
      for (int k = 0; k < 12; k++) wl[k * 64] = rw[k];
      x.sync();
    }
    float ho[kObsDimMax];
    update_tool_offset<T, KW, true>(AA.p, e);
    observe<T, KW, true>(1, e, ho, P.ee_task != 0);
    observe_joints<T, KW>(e, ho);
    stage_obs<OD>(lds + lane * OD, ho);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();          // one wave stages and flushes: LDS accesses of a wave complete in order
    const int row0h = blockIdx.x * 64;
    flush_obs<OD>(lds, io.obs + size_t(row0h) * OD, min(64, hd.n - row0h), 64, lane);
    __syncthreads();   // stores acknowledged (s_waitcnt vmcnt(0)) before the main wave may overwrite the rows of reset lanes
    return;
  }
  T reward; float o[kObsDimMax]; bool was_reset; int ep_len; float ep_ret;
  uint32_t bits = step_lane<T, NROT, KW, VAR, NJ, ARM_ROLE_MAIN, LdsXchg>(P, C, AA, e, act, i, active, reward, o, io, tile, lane, false, was_reset,
                                                                        ep_len, ep_ret, x);
  const bool is_done = active && (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
  accumulate_stats(io.stats, int(blockIdx.x), bits, is_done, ep_len, ep_ret);
  store_env_step<T, KW, NJ>(tile, lane, e);
  if (was_reset) store_env_episode<T, KW>(1, tile, lane, e);
  if (active) {
    reinterpret_cast<T*>(io.reward)[i] = reward;
    io.done[i] = is_done ? 1 : 0;
    io.info[i] = bits;
  }
  __syncthreads();     // the helper's observation rows have landed
  if (is_done && io.terminal_obs) {   // SB3 terminal_observation = the pre-reset observation = the row the helper staged in LDS for this lane
    float* t = io.terminal_obs + size_t(i) * OD;
#pragma unroll
    for (int j = 0; j < OD; j++) t[j] = lds[lane * OD + j];
  }
  if (was_reset && active) {   // rare: this lane's env was auto-reset -> its row must hold the post-reset observation
    float* d = io.obs + size_t(i) * OD;
#pragma unroll
    for (int j = 0; j < OD; j++) d[j] = o[j];
  }
#ifdef AMENV_STAMPS
  AMENV_STAMP_DRAIN();
  AMENV_STAMP(7);          // all stores acknowledged
  if (lane == 0 && blockIdx.x < kStampWaves)
    for (int kk = 0; kk < kStampSlots; kk++) io.stats[kStampBase + blockIdx.x * kStampSlots + kk] = stamps_[kk];
#endif
}

// Stage-wave step kernel for the hexacopter + z,x,x arm between the lane-team kernel's range and the throughput regime (BASELINE
// config 4: 32768 envs per GPU).  One tile of 64 environments per 320-thread workgroup:
//   waves 0..3  one RK4 STAGE each: the joint servos do not feel the base, so the joint state of every stage follows from (th, thd, cmd)
//               alone; the wave forms its stage's joint configuration and reduces everything the base dynamics needs from it to 36
//               numbers per env (arm_kin_aggregates, ~2/3 of the step's arithmetic) -- all four stages side by side instead of in sequence;
//               wave 3 also integrates the joints.  After the ONE barrier of the RK4, waves 0..2 compute the three Philox blocks a reset
//               of their lanes would consume and leave them in LDS (second barrier), wave 3 is done.
//   wave 4      everything else: mixer, RK4 on the 13 base states with arm_dyn_agg on the aggregates (~160 instructions per stage instead
//               of ~960), forward kinematics, task step, observation, auto-reset, stores.
// The critical path is one stage's kinematics + four short base stages instead of four full right-hand sides (10.4 us with the two-wave
// kernel at 32768 envs).  Barriers: waves 0..2 and 4 execute two, wave 3 one (a finished wave no longer counts), all in uniform flow.
template <typename T, int NROT>
__global__ __launch_bounds__(320) void step_kernel_armk(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, const float4* __restrict__ actions,
                                                        float* __restrict__ obs, void* __restrict__ reward_out, uint8_t* __restrict__ done,
                                                        uint32_t* __restrict__ info, const StepTail tl, const HotParams<T, NROT> P, const ColdParams C,
                                                        const ArmArg<T, 3> AA) {
  constexpr int KW = 1, VAR = VAR_V2, NJ = 3;   // (T = double: the logic-gate build of the same kernel, aggregates exchanged in fp64)
  constexpr int OD = ObsDim<VAR, NJ>::value, AD = kActDim + NJ;
  const Head hd{blob, tile_bytes, n_envs};
  const StepIO io{actions, obs, reward_out, done, info, tl.terminal_obs, tl.ep_return, tl.ep_len, tl.stats};
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [64 rows x OD] obs staging (f32) | [4][kAggSlots][64] aggregates (T) | [6][64] joints (T) | [12][64] reset words
  static_assert((64 * OD * sizeof(float)) % 8 == 0, "the aggregates start 8-byte aligned");
  T* agg = reinterpret_cast<T*>(lds + 64 * OD);
  T* jn = agg + 4 * kAggSlots * 64;
  uint32_t* words = reinterpret_cast<uint32_t*>(jn + 6 * 64);
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);   // wave-uniform: 0..3 stage waves, 4 main
  const int i = blockIdx.x * 64 + lane;
  const bool active = i < hd.n;
  char* tile = const_cast<char*>(tile_base(hd.blob, hd.tile_bytes, i));
  const float* ap = reinterpret_cast<const float*>(io.actions) + size_t(min(i, hd.n - 1)) * AD;
  if (role < 4) {
    const Vec4<T> ja = *gptr<T>(tile, lane, 4 + KW), jr = *gptr<T>(tile, lane, 5 + KW);
    const int32_t episode = iptr4(tile, lane)->w;
    T cmd[3];
#pragma unroll
    for (int k = 0; k < 3; k++) cmd[k] = T(__builtin_fmaf(ap[4 + k], AA.p.half[k], AA.p.mid[k]));
    const T th0[3] = {ja.a, ja.b, ja.c}, td0[3] = {jr.a, jr.b, jr.c};
    arm_kin_stage<T>(P, AA.p, role, th0, td0, cmd, agg, jn, lane);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                     // aggregates (and the integrated joints) are in LDS
    __builtin_amdgcn_sched_barrier(0);
    if (role == 3) return;
    const int64_t gid = C.gid0 + i;
    uint32_t w[4];
    philox4x32_10(C.seed_lo, C.seed_hi, uint32_t(uint64_t(gid)), uint32_t(uint64_t(gid) >> 32), uint32_t(episode), uint32_t(role), w);
#pragma unroll
    for (int k = 0; k < 4; k++) words[(4 * role + k) * 64 + lane] = w[k];
    __syncthreads();                     // reset words published (pairs with the barrier in step_lane)
    return;
  }
  Env<T, KW> e;
  load_env<T, KW, 0>(1, tile, lane, e);    // the joints arrive integrated from wave 3
  float act[AD];
#pragma unroll
  for (int j = 0; j < 4; j++) act[j] = ap[j];
  act[4] = act[5] = act[6] = 0.0f;
  const StagedXchg<T> x{agg, jn, words, lane};
  T reward; float o[kObsDimMax]; bool was_reset; int ep_len; float ep_ret;
  uint32_t bits = step_lane<T, NROT, KW, VAR, NJ, ARM_ROLE_STAGED, StagedXchg<T>>(P, C, AA, e, act, i, active, reward, o, io, tile, lane, false, was_reset,
                                                                             ep_len, ep_ret, x);
  const bool is_done = active && (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
  accumulate_stats(io.stats, int(blockIdx.x), bits, is_done, ep_len, ep_ret);
  store_env_step<T, KW, NJ>(tile, lane, e);
  if (was_reset) store_env_episode<T, KW>(1, tile, lane, e);
  if (active) {
    reinterpret_cast<T*>(io.reward)[i] = reward;
    io.done[i] = is_done ? 1 : 0;
    io.info[i] = bits;
  }
  stage_obs<OD>(lds + lane * OD, o);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();          // one wave stages and flushes: LDS accesses of a wave complete in order
  const int row0 = blockIdx.x * 64;
  flush_obs<OD>(lds, io.obs + size_t(row0) * OD, min(64, hd.n - row0), 64, lane);
}

// n_steps control steps per launch with open-loop actions [T][N][4]; per-step outputs [T][N]...
// State stays in registers across steps: HBM traffic per env-step drops to action + outputs.
template <typename T, int NROT, int KW, int VAR, int NJ>
__global__ __launch_bounds__(256) void rollout_kernel(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, const float4* __restrict__ actions,
                                                      float* __restrict__ obs, void* __restrict__ reward_out, uint8_t* __restrict__ done,
                                                      uint32_t* __restrict__ info, int n_steps, const StepTail tl, const HotParams<T, NROT> P,
                                                      const ColdParams C, const ArmArg<T, NJ> AA) {
  constexpr int OD = ObsDim<VAR, NJ>::value, AD = kActDim + NJ;
  const Head hd{blob, tile_bytes, n_envs};
  const StepIO io{actions, obs, reward_out, done, info, nullptr, nullptr, nullptr, tl.stats};
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int BS = int(blockDim.x);
  const int i = blockIdx.x * BS + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool active = i < hd.n;
  const int row0 = blockIdx.x * BS;
  const int rows = min(BS, hd.n - row0);
  const size_t n = size_t(hd.n);
  char* tile = const_cast<char*>(tile_base(hd.blob, hd.tile_bytes, i));
  const int K = KW == 1 ? 1 : P.K;
  Env<T, KW> e;
  load_env<T, KW, NJ>(K, tile, lane, e);
  bool any_reset = false;
  StepIO io_t = io; io_t.terminal_obs = nullptr; io_t.ep_return = nullptr; io_t.ep_len = nullptr;
  for (int t = 0; t < n_steps; t++) {
    float act[AD];
    {
      const float* ap = reinterpret_cast<const float*>(io.actions) + (size_t(t) * n + min(i, hd.n - 1)) * AD;
      if constexpr (NJ == 0) { const float4 a = *reinterpret_cast<const float4*>(ap); act[0] = a.x; act[1] = a.y; act[2] = a.z; act[3] = a.w; }
      else {
#pragma unroll
        for (int j = 0; j < AD; j++) act[j] = ap[j];
      }
    }
    T reward; float o[kObsDimMax]; bool was_reset; int ep_len; float ep_ret;
    uint32_t bits = step_lane<T, NROT, KW, VAR, NJ>(P, C, AA, e, act, i, active, reward, o, io_t, tile, lane, any_reset, was_reset, ep_len, ep_ret);
    any_reset |= was_reset;
    const bool is_done = active && (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
    accumulate_stats(io.stats, int((blockIdx.x * blockDim.x + threadIdx.x) >> 6), bits, is_done, ep_len, ep_ret);
    if (active) {
      if (io.reward) reinterpret_cast<T*>(io.reward)[size_t(t) * n + i] = reward;
      if (io.done) io.done[size_t(t) * n + i] = is_done ? 1 : 0;
      if (io.info) io.info[size_t(t) * n + i] = bits;
    }
    if (io.obs) {
      stage_obs<OD>(lds + threadIdx.x * OD, o);
      __syncthreads();
      flush_obs<OD>(lds, io.obs + (size_t(t) * n + row0) * OD, rows, int(blockDim.x), int(threadIdx.x));
      __syncthreads();
    }
  }
  store_env_step<T, KW, NJ>(tile, lane, e, K);
  if (any_reset) store_env_episode<T, KW>(K, tile, lane, e);
}

// WaypointQuadEnv.reset for masked envs (mask null = all) + observation of every env.
// Launched over whole tiles: padding lanes (i >= n) are always reset so that they hold a valid state.
// joints (read from the tile: they are not part of the generic Env load) + forward kinematics for the cold kernels
template <typename T, int KW>
__device__ __forceinline__ void joints_and_tool(const ArmParams<T>& A, int K, int nj, char* tile, int lane, Env<T, KW>& e) {
#pragma unroll
  for (int j = 0; j < AMENV_MAX_JOINTS; j++) {
    e.th[j] = j < nj ? *fptr<T>(tile, lane, AMENV_F_WP0 + 3 * K + j, K, nj) : T(0);
    e.thd[j] = j < nj ? *fptr<T>(tile, lane, AMENV_F_WP0 + 3 * K + nj + j, K, nj) : T(0);
  }
  e.eox = e.eoy = e.eoz = T(0);
  if (nj > 0) update_tool_offset<T, KW, false>(A, e);
}

template <typename T>
__global__ void reset_kernel(int n, int n_pad, int K, int variant, int nj, int ee_task, uint32_t tile_bytes, const ColdParams C, const ArmParams<T> A,
                             void* __restrict__ blob, const uint8_t* __restrict__ mask, float* __restrict__ obs, int pad_only) {
  const bool v1 = variant != AMENV_TASK_V2_SCALED20;
  const int od = v1 ? 17 : 20 + 2 * nj + (nj > 0 ? 3 : 0);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pad) return;   // the blob holds n_pad / 64 tiles: nothing exists beyond the last padded lane
  const int lane = threadIdx.x & 63;
  const bool active = i < n;
  char* tile = const_cast<char*>(tile_base(blob, tile_bytes, i));
  Env<T, AMENV_MAX_WAYPOINTS> e;
  load_env<T, AMENV_MAX_WAYPOINTS>(K, tile, lane, e);
  const bool do_reset = active ? (!pad_only && (!mask || mask[i])) : true;  // pad_only: amenv_create's init pass
  if (do_reset) {
    reset_env<T, AMENV_MAX_WAYPOINTS>(C, K, v1, e, C.gid0 + i);
    store_env_step<T, AMENV_MAX_WAYPOINTS>(tile, lane, e);
    store_env_episode<T, AMENV_MAX_WAYPOINTS>(K, tile, lane, e);
    for (int j = 0; j < 2 * nj; j++) *fptr<T>(tile, lane, AMENV_F_WP0 + 3 * K + j, K, nj) = T(0);   // arm at home, at rest
  }
  if (obs && active) {
    float o[kObsDimMax];
    joints_and_tool<T, AMENV_MAX_WAYPOINTS>(A, K, nj, tile, lane, e);   // (zero joints after a reset)
    if (do_reset && nj > 0) { e.eox = A.ee_home[0]; e.eoy = A.ee_home[1]; e.eoz = A.ee_home[2]; }   // as the step kernels' reset path
    if (v1) observe_v1<T, AMENV_MAX_WAYPOINTS>(variant == AMENV_TASK_V1_RAW17, e, o);
    else observe<T, AMENV_MAX_WAYPOINTS, true>(K, e, o, nj > 0 && ee_task != 0);
    if (nj > 0) observe_joints<T, AMENV_MAX_WAYPOINTS>(e, o);
    for (int j = 0; j < od; j++) obs[size_t(i) * od + j] = o[j];
  }
}

// _get_observation of the current state for every env (no stepping).
template <typename T>
__global__ void observe_kernel(int n, int K, int variant, int nj, int ee_task, uint32_t tile_bytes, const ArmParams<T> A, const void* __restrict__ blob,
                               float* __restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool v1 = variant != AMENV_TASK_V2_SCALED20;
  const int od = v1 ? 17 : 20 + 2 * nj + (nj > 0 ? 3 : 0);
  Env<T, AMENV_MAX_WAYPOINTS> e;
  char* tile = const_cast<char*>(tile_base(blob, tile_bytes, i));
  load_env<T, AMENV_MAX_WAYPOINTS>(K, tile, threadIdx.x & 63, e);
  joints_and_tool<T, AMENV_MAX_WAYPOINTS>(A, K, nj, tile, threadIdx.x & 63, e);
  float o[kObsDimMax];
  if (v1) observe_v1<T, AMENV_MAX_WAYPOINTS>(variant == AMENV_TASK_V1_RAW17, e, o);
  else observe<T, AMENV_MAX_WAYPOINTS, true>(K, e, o, nj > 0 && ee_task != 0);
  if (nj > 0) observe_joints<T, AMENV_MAX_WAYPOINTS>(e, o);
  for (int j = 0; j < od; j++) obs[size_t(i) * od + j] = o[j];
}

// amenv_ee_position: world position of the tool point of every env (forward kinematics of the stored state), out [N][3] f32
template <typename T>
__global__ void ee_position_kernel(int n, int K, int nj, uint32_t tile_bytes, const ArmParams<T> A, const void* __restrict__ blob, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Env<T, AMENV_MAX_WAYPOINTS> e;
  char* tile = const_cast<char*>(tile_base(blob, tile_bytes, i));
  load_env<T, AMENV_MAX_WAYPOINTS>(K, tile, threadIdx.x & 63, e);
  joints_and_tool<T, AMENV_MAX_WAYPOINTS>(A, K, nj, tile, threadIdx.x & 63, e);
  out[size_t(i) * 3] = float(e.px + e.eox); out[size_t(i) * 3 + 1] = float(e.py + e.eoy); out[size_t(i) * 3 + 2] = float(e.pz + e.eoz);
}

// amenv_get_state / amenv_set_state: the public struct-of-arrays view (fstate [NF][N] T, istate [4][N] i32)
// <-> the internal tile layout.  to_api != 0: tiles -> SoA; else SoA -> tiles.
template <typename T>
__global__ void transpose_state_kernel(int n, int nf, int K, int nj, uint32_t tile_bytes, void* __restrict__ blob, T* __restrict__ fapi,
                                       int32_t* __restrict__ iapi, int to_api) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  char* tile = const_cast<char*>(tile_base(blob, tile_bytes, i));
  const int lane = threadIdx.x & 63;
  if (fapi)
    for (int f = 0; f < nf; f++) {
      if (to_api) fapi[size_t(f) * n + i] = *fptr<T>(tile, lane, f, K, nj);
      else *fptr<T>(tile, lane, f, K, nj) = fapi[size_t(f) * n + i];
    }
  if (iapi)
    for (int f = 0; f < AMENV_I_NFIELDS; f++) {
      if (to_api) iapi[size_t(f) * n + i] = *iptr(tile, lane, f);
      else *iptr(tile, lane, f) = iapi[size_t(f) * n + i];
    }
}

}  // namespace amenv_dev
