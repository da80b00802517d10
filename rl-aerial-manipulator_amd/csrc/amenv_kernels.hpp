// amenv_kernels.hpp -- gfx950 kernels of the batched waypoint environment.
//
// Data layout in HBM (DESIGN.md "Layout"):
//   fstate  T   [n_float_fields][N]   struct-of-arrays: lane i of a wave touches element i of
//   istate  i32 [4][N]                every field => each field access is one coalesced 256-B
//                                      (fp32) wave transaction
//   actions f32 [N][4]                one float4 per lane, coalesced
//   obs     f32 [N][20]               row-major for the policy MLP: rows are staged through LDS
//                                      and written as contiguous 1-KiB float4 wave stores
// One lane owns one environment for the whole step: load -> mixer -> RK4 -> reward -> state
// machine -> (masked) auto-reset -> observation, a single launch per control step.
#pragma once
#include "amenv_model.hpp"

namespace amenv_dev {

enum StatSlot { S_EPISODES = 0, S_TERMINATED, S_TRUNCATED, S_SUCCESS, S_CRASHED, S_OOB, S_NONFINITE, S_LENGTH, S_RETURN_Q10, S_COUNT };

template <typename T>
__device__ __forceinline__ void load_env(const Params<T>& P, const T* __restrict__ fs, const int32_t* __restrict__ is, int i,
                                         Env<T>& e) {
  const size_t n = size_t(P.n);
  const T* f = fs + i;
  e.px = f[0 * n]; e.py = f[1 * n]; e.pz = f[2 * n];
  e.vx = f[3 * n]; e.vy = f[4 * n]; e.vz = f[5 * n];
  e.qw = f[6 * n]; e.qx = f[7 * n]; e.qy = f[8 * n]; e.qz = f[9 * n];
  e.wx = f[10 * n]; e.wy = f[11 * n]; e.wz = f[12 * n];
  e.final_yaw = f[AMENV_F_FINAL_YAW * n];
  e.last_distance = f[AMENV_F_LAST_DISTANCE * n];
  e.ep_return = f[AMENV_F_EP_RETURN * n];
#pragma unroll
  for (int k = 0; k < AMENV_MAX_WAYPOINTS; k++) {
    if (k < P.K) {
      e.wp[k][0] = f[(AMENV_F_WP0 + 3 * k + 0) * n]; e.wp[k][1] = f[(AMENV_F_WP0 + 3 * k + 1) * n];
      e.wp[k][2] = f[(AMENV_F_WP0 + 3 * k + 2) * n];
    } else {
      e.wp[k][0] = e.wp[k][1] = e.wp[k][2] = T(0);
    }
  }
  const int32_t* s = is + i;
  e.step = s[AMENV_I_STEP * n]; e.counter = s[AMENV_I_COUNTER * n]; e.flags = s[AMENV_I_FLAGS * n];
  e.episode = 0;  // the episode counter is only needed by a reset: loaded there (load_episode)
}

template <typename T>
__device__ __forceinline__ void load_episode(const Params<T>& P, const int32_t* __restrict__ is, int i, Env<T>& e) {
  e.episode = is[size_t(AMENV_I_EPISODE) * size_t(P.n) + i];
}

// per-step mutable part of the state
template <typename T>
__device__ __forceinline__ void store_env_step(const Params<T>& P, T* __restrict__ fs, int32_t* __restrict__ is, int i,
                                               const Env<T>& e) {
  const size_t n = size_t(P.n);
  T* f = fs + i;
  f[0 * n] = e.px; f[1 * n] = e.py; f[2 * n] = e.pz;
  f[3 * n] = e.vx; f[4 * n] = e.vy; f[5 * n] = e.vz;
  f[6 * n] = e.qw; f[7 * n] = e.qx; f[8 * n] = e.qy; f[9 * n] = e.qz;
  f[10 * n] = e.wx; f[11 * n] = e.wy; f[12 * n] = e.wz;
  f[AMENV_F_LAST_DISTANCE * n] = e.last_distance;
  f[AMENV_F_EP_RETURN * n] = e.ep_return;
  int32_t* s = is + i;
  s[AMENV_I_STEP * n] = e.step; s[AMENV_I_COUNTER * n] = e.counter; s[AMENV_I_FLAGS * n] = e.flags;
}

// per-episode constants, written only by lanes that were reset
template <typename T>
__device__ __forceinline__ void store_env_episode(const Params<T>& P, T* __restrict__ fs, int32_t* __restrict__ is, int i,
                                                  const Env<T>& e) {
  const size_t n = size_t(P.n);
  T* f = fs + i;
  f[AMENV_F_FINAL_YAW * n] = e.final_yaw;
#pragma unroll
  for (int k = 0; k < AMENV_MAX_WAYPOINTS; k++)
    if (k < P.K) {
      f[(AMENV_F_WP0 + 3 * k + 0) * n] = e.wp[k][0]; f[(AMENV_F_WP0 + 3 * k + 1) * n] = e.wp[k][1];
      f[(AMENV_F_WP0 + 3 * k + 2) * n] = e.wp[k][2];
    }
  is[AMENV_I_EPISODE * n + i] = e.episode;
}

// Stage this lane's 80-B observation row in LDS.  Row stride 80 B: the 8 lanes of a
// ds_write_b128 group land on banks {0,20,8,28,16,4,24,12}+0..3 -> conflict-free.
__device__ __forceinline__ void stage_obs(float* lds_row, const float* o) {
  float4* d = reinterpret_cast<float4*>(lds_row);
#pragma unroll
  for (int j = 0; j < 5; j++) d[j] = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
}

// Copy the block's staged rows to global memory: consecutive lanes write consecutive float4
// (1 KiB per wave instruction), rows_valid*5 float4 in all.
template <int BS>
__device__ __forceinline__ void flush_obs(const float* lds, float* __restrict__ obs_block, int rows_valid) {
  const float4* s = reinterpret_cast<const float4*>(lds);
  float4* d = reinterpret_cast<float4*>(obs_block);
  const int nvec = rows_valid * (kObsDim / 4);
#pragma unroll
  for (int j = 0; j < kObsDim / 4; j++) {
    const int f = j * BS + int(threadIdx.x);
    if (f < nvec) d[f] = s[f];
  }
}

__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Monitor-style running totals: ballot+popcount for the counters, a shuffle reduction only in
// waves that finished an episode this step, one atomic per wave and counter.
__device__ __forceinline__ void accumulate_stats(unsigned long long* __restrict__ stats, uint32_t bits, bool is_done, int ep_len,
                                                 float ep_ret) {
  const unsigned long long m_done = __ballot(is_done);
  if (m_done == 0ull) return;  // wave-uniform
  const unsigned long long m_term = __ballot(is_done && (bits & AMENV_INFO_TERMINATED));
  const unsigned long long m_trunc = __ballot(is_done && (bits & AMENV_INFO_TRUNCATED) && !(bits & AMENV_INFO_TERMINATED));
  const unsigned long long m_succ = __ballot(is_done && (bits & AMENV_INFO_SUCCESS));
  const unsigned long long m_crash = __ballot(is_done && (bits & AMENV_INFO_CRASHED));
  const unsigned long long m_oob = __ballot(is_done && (bits & AMENV_INFO_OOB));
  const unsigned long long m_nf = __ballot(is_done && (bits & AMENV_INFO_NONFINITE));
  const float r = is_done ? ep_ret : 0.0f;
  const long long q = __builtin_isfinite(r) ? llrintf(r * 1024.0f) : 0ll;
  const long long len_sum = wave_sum(is_done ? (long long)ep_len : 0ll);
  const long long ret_sum = wave_sum(q);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&stats[S_EPISODES], (unsigned long long)__popcll(m_done));
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
template <typename T, int NROT>
__device__ __forceinline__ uint32_t step_lane(const Params<T>& P, Env<T>& e, const float4 a, int i, T& reward, float* o,
                                              const StepIO& io, const int32_t* __restrict__ is, bool was_reset_before,
                                              bool& was_reset, int& ep_len_out, float& ep_ret_out) {
  dynamics<T, NROT>(P, e, a.x, a.y, a.z, a.w);
  uint32_t bits = task_step(P, e, reward);
  e.ep_return += reward;
  observe(P, e, o);
  was_reset = false;
  ep_len_out = 0; ep_ret_out = 0.0f;
  if (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) {  // SB3 DummyVecEnv + Monitor contract
    ep_len_out = e.step; ep_ret_out = float(e.ep_return);
    if (io.terminal_obs) {
      float4* t = reinterpret_cast<float4*>(io.terminal_obs + size_t(i) * kObsDim);
#pragma unroll
      for (int j = 0; j < 5; j++) t[j] = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
    }
    if (io.ep_return) io.ep_return[i] = ep_ret_out;
    if (io.ep_len) io.ep_len[i] = ep_len_out;
    if (P.flags & AMENV_FLAG_AUTO_RESET) {
      if (!was_reset_before) load_episode(P, is, i, e);
      reset_env(P, e, P.gid0 + i);
      observe(P, e, o);
      bits |= AMENV_INFO_WAS_RESET;
      was_reset = true;
    }
  }
  return bits;
}

template <typename T, int NROT, int BS>
__global__ __launch_bounds__(BS) void step_kernel(const Params<T> P, T* __restrict__ fs, int32_t* __restrict__ is, const StepIO io) {
  __shared__ __attribute__((aligned(16))) float lds[BS * kObsDim];
  const int i = blockIdx.x * BS + threadIdx.x;
  const bool active = i < P.n;
  uint32_t bits = 0; bool is_done = false; int ep_len = 0; float ep_ret = 0.0f;
  if (active) {
    Env<T> e;
    load_env(P, fs, is, i, e);
    const float4 a = io.actions[i];
    T reward; float o[kObsDim]; bool was_reset;
    bits = step_lane<T, NROT>(P, e, a, i, reward, o, io, is, false, was_reset, ep_len, ep_ret);
    is_done = (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
    store_env_step(P, fs, is, i, e);
    if (was_reset) store_env_episode(P, fs, is, i, e);
    reinterpret_cast<T*>(io.reward)[i] = reward;
    io.done[i] = is_done ? 1 : 0;
    io.info[i] = bits;
    stage_obs(lds + threadIdx.x * kObsDim, o);
  }
  __syncthreads();
  const int row0 = blockIdx.x * BS;
  const int rows = min(BS, P.n - row0);
  flush_obs<BS>(lds, io.obs + size_t(row0) * kObsDim, rows);
  accumulate_stats(io.stats, bits, is_done, ep_len, ep_ret);
}

// n_steps control steps per launch with open-loop actions [T][N][4]; per-step outputs [T][N]...
// State stays in registers across steps: HBM traffic per env-step drops to action + outputs.
template <typename T, int NROT, int BS>
__global__ __launch_bounds__(BS) void rollout_kernel(const Params<T> P, T* __restrict__ fs, int32_t* __restrict__ is, const StepIO io,
                                                     int n_steps) {
  __shared__ __attribute__((aligned(16))) float lds[BS * kObsDim];
  const int i = blockIdx.x * BS + threadIdx.x;
  const bool active = i < P.n;
  const int row0 = blockIdx.x * BS;
  const int rows = min(BS, P.n - row0);
  const size_t n = size_t(P.n);
  Env<T> e;
  if (active) load_env(P, fs, is, i, e);
  bool any_reset = false;
  for (int t = 0; t < n_steps; t++) {
    uint32_t bits = 0; bool is_done = false; int ep_len = 0; float ep_ret = 0.0f;
    if (active) {
      const float4 a = io.actions[size_t(t) * n + i];
      T reward; float o[kObsDim]; bool was_reset;
      StepIO io_t = io; io_t.terminal_obs = nullptr; io_t.ep_return = nullptr; io_t.ep_len = nullptr;
      bits = step_lane<T, NROT>(P, e, a, i, reward, o, io_t, is, any_reset, was_reset, ep_len, ep_ret);
      any_reset |= was_reset;
      is_done = (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
      if (io.reward) reinterpret_cast<T*>(io.reward)[size_t(t) * n + i] = reward;
      if (io.done) io.done[size_t(t) * n + i] = is_done ? 1 : 0;
      if (io.info) io.info[size_t(t) * n + i] = bits;
      if (io.obs) stage_obs(lds + threadIdx.x * kObsDim, o);
    }
    if (io.obs) {
      __syncthreads();
      flush_obs<BS>(lds, io.obs + (size_t(t) * n + row0) * kObsDim, rows);
      __syncthreads();
    }
    accumulate_stats(io.stats, bits, is_done, ep_len, ep_ret);
  }
  if (active) {
    store_env_step(P, fs, is, i, e);
    if (any_reset) store_env_episode(P, fs, is, i, e);
  }
}

// WaypointQuadEnv.reset for masked envs (mask null = all) + observation of every env.
template <typename T>
__global__ void reset_kernel(const Params<T> P, T* __restrict__ fs, int32_t* __restrict__ is, const uint8_t* __restrict__ mask,
                             float* __restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  Env<T> e;
  load_env(P, fs, is, i, e);
  if (!mask || mask[i]) {
    load_episode(P, is, i, e);
    reset_env(P, e, P.gid0 + i);
    store_env_step(P, fs, is, i, e);
    store_env_episode(P, fs, is, i, e);
  }
  if (obs) {
    float o[kObsDim];
    observe(P, e, o);
#pragma unroll
    for (int j = 0; j < kObsDim; j++) obs[size_t(i) * kObsDim + j] = o[j];
  }
}

// _get_observation of the current state for every env (no stepping).
template <typename T>
__global__ void observe_kernel(const Params<T> P, const T* __restrict__ fs, const int32_t* __restrict__ is, float* __restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  Env<T> e;
  load_env(P, fs, is, i, e);
  float o[kObsDim];
  observe(P, e, o);
#pragma unroll
  for (int j = 0; j < kObsDim; j++) obs[size_t(i) * kObsDim + j] = o[j];
}

}  // namespace amenv_dev
