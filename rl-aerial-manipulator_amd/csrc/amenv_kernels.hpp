// amenv_kernels.hpp -- gfx950 kernels of the batched waypoint environment.
//
// Data layout in HBM (DESIGN.md "Layout")
//   state blob : tiles of 64 environments (one wavefront each), tile t at byte t * tile_bytes:
//                  [4 int fields   ][64 lanes] i32      offsets 0, 256, 512, 768
//                  [NF float fields][64 lanes] T        offset 1024 + f * 64 * sizeof(T)
//                One wave's whole state is ONE contiguous block (4.9 KiB for fp32, K = 1); every field
//                access is a fully coalesced 256-B (fp32) wave transaction whose field offset is an
//                instruction immediate: one base address per lane, no per-field address arithmetic.
//   actions    : f32 [N][4]   one float4 per lane, coalesced
//   obs        : f32 [N][20]  row-major for the policy MLP: rows are staged through LDS and written
//                             as contiguous 1-KiB float4 wave stores
// One lane owns one environment for the whole step: load -> mixer -> RK4 -> reward -> state machine
// -> (masked) auto-reset -> observation, a single launch per control step.
#pragma once
#include "amenv_model.hpp"

namespace amenv_dev {

enum StatSlot { S_EPISODES = 0, S_TERMINATED, S_TRUNCATED, S_SUCCESS, S_CRASHED, S_OOB, S_NONFINITE, S_LENGTH, S_RETURN_Q10, S_COUNT };

constexpr uint32_t kIntBytes = AMENV_I_NFIELDS * 64 * sizeof(int32_t);  // 1024

__host__ __device__ inline uint32_t tile_bytes_for(int n_float_fields, int tsize) { return kIntBytes + uint32_t(n_float_fields) * 64u * uint32_t(tsize); }

// byte address of the tile that holds env i
__device__ __forceinline__ const char* tile_base(const void* blob, uint32_t tile_bytes, int i) {
  return static_cast<const char*>(blob) + size_t(i >> 6) * tile_bytes;
}
template <typename T> __device__ __forceinline__ T* fptr(char* tile, int lane, int field) {
  return reinterpret_cast<T*>(tile + kIntBytes + size_t(field) * 64 * sizeof(T)) + lane;
}
__device__ __forceinline__ int32_t* iptr(char* tile, int lane, int field) { return reinterpret_cast<int32_t*>(tile + size_t(field) * 256) + lane; }

template <typename T, int KW>
__device__ __forceinline__ void load_env(int K, const char* tile_c, int lane, Env<T, KW>& e) {
  char* tile = const_cast<char*>(tile_c);
  e.px = *fptr<T>(tile, lane, 0); e.py = *fptr<T>(tile, lane, 1); e.pz = *fptr<T>(tile, lane, 2);
  e.vx = *fptr<T>(tile, lane, 3); e.vy = *fptr<T>(tile, lane, 4); e.vz = *fptr<T>(tile, lane, 5);
  e.qw = *fptr<T>(tile, lane, 6); e.qx = *fptr<T>(tile, lane, 7); e.qy = *fptr<T>(tile, lane, 8); e.qz = *fptr<T>(tile, lane, 9);
  e.wx = *fptr<T>(tile, lane, 10); e.wy = *fptr<T>(tile, lane, 11); e.wz = *fptr<T>(tile, lane, 12);
  e.final_yaw = *fptr<T>(tile, lane, AMENV_F_FINAL_YAW);
  e.last_distance = *fptr<T>(tile, lane, AMENV_F_LAST_DISTANCE);
  e.ep_return = *fptr<T>(tile, lane, AMENV_F_EP_RETURN);
#pragma unroll
  for (int k = 0; k < KW; k++) {
    if (k < K) {
      e.wp[k][0] = *fptr<T>(tile, lane, AMENV_F_WP0 + 3 * k + 0); e.wp[k][1] = *fptr<T>(tile, lane, AMENV_F_WP0 + 3 * k + 1);
      e.wp[k][2] = *fptr<T>(tile, lane, AMENV_F_WP0 + 3 * k + 2);
    } else {
      e.wp[k][0] = e.wp[k][1] = e.wp[k][2] = T(0);
    }
  }
  e.step = *iptr(tile, lane, AMENV_I_STEP); e.counter = *iptr(tile, lane, AMENV_I_COUNTER); e.flags = *iptr(tile, lane, AMENV_I_FLAGS);
  e.episode = 0;  // the episode counter is only needed by a reset: loaded there
}

// per-step mutable part of the state
template <typename T, int KW>
__device__ __forceinline__ void store_env_step(char* tile, int lane, const Env<T, KW>& e) {
  *fptr<T>(tile, lane, 0) = e.px; *fptr<T>(tile, lane, 1) = e.py; *fptr<T>(tile, lane, 2) = e.pz;
  *fptr<T>(tile, lane, 3) = e.vx; *fptr<T>(tile, lane, 4) = e.vy; *fptr<T>(tile, lane, 5) = e.vz;
  *fptr<T>(tile, lane, 6) = e.qw; *fptr<T>(tile, lane, 7) = e.qx; *fptr<T>(tile, lane, 8) = e.qy; *fptr<T>(tile, lane, 9) = e.qz;
  *fptr<T>(tile, lane, 10) = e.wx; *fptr<T>(tile, lane, 11) = e.wy; *fptr<T>(tile, lane, 12) = e.wz;
  *fptr<T>(tile, lane, AMENV_F_LAST_DISTANCE) = e.last_distance;
  *fptr<T>(tile, lane, AMENV_F_EP_RETURN) = e.ep_return;
  *iptr(tile, lane, AMENV_I_STEP) = e.step; *iptr(tile, lane, AMENV_I_COUNTER) = e.counter; *iptr(tile, lane, AMENV_I_FLAGS) = e.flags;
}

// per-episode constants, written only by lanes that were reset
template <typename T, int KW>
__device__ __forceinline__ void store_env_episode(int K, char* tile, int lane, const Env<T, KW>& e) {
  *fptr<T>(tile, lane, AMENV_F_FINAL_YAW) = e.final_yaw;
#pragma unroll
  for (int k = 0; k < KW; k++)
    if (k < K) {
      *fptr<T>(tile, lane, AMENV_F_WP0 + 3 * k + 0) = e.wp[k][0]; *fptr<T>(tile, lane, AMENV_F_WP0 + 3 * k + 1) = e.wp[k][1];
      *fptr<T>(tile, lane, AMENV_F_WP0 + 3 * k + 2) = e.wp[k][2];
    }
  *iptr(tile, lane, AMENV_I_EPISODE) = e.episode;
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
__device__ __forceinline__ void flush_obs(const float* lds, float* __restrict__ obs_block, int rows_valid) {
  const float4* s = reinterpret_cast<const float4*>(lds);
  float4* d = reinterpret_cast<float4*>(obs_block);
  const int nvec = rows_valid * (kObsDim / 4);
  const int bs = int(blockDim.x);
#pragma unroll
  for (int j = 0; j < kObsDim / 4; j++) {
    const int f = j * bs + int(threadIdx.x);
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
template <typename T, int NROT, int KW>
__device__ __forceinline__ uint32_t step_lane(const HotParams<T, NROT>& P, const ColdParams& C, Env<T, KW>& e, const float4 a, int i,
                                              bool active, T& reward, float* o, const StepIO& io, char* tile, int lane,
                                              bool have_episode, bool& was_reset, int& ep_len_out, float& ep_ret_out) {
  const int K = KW == 1 ? 1 : P.K;
  dynamics<T, NROT, KW>(P, e, a.x, a.y, a.z, a.w);
  uint32_t bits = task_step<T, KW>(P, e, reward);
  e.ep_return += reward;
  observe<T, KW>(K, e, o);
  was_reset = false;
  ep_len_out = 0; ep_ret_out = 0.0f;
  if (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) {  // SB3 DummyVecEnv + Monitor contract
    ep_len_out = e.step; ep_ret_out = float(e.ep_return);
    if (active) {
      if (io.terminal_obs) {
        float4* t = reinterpret_cast<float4*>(io.terminal_obs + size_t(i) * kObsDim);
#pragma unroll
        for (int j = 0; j < 5; j++) t[j] = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
      }
      if (io.ep_return) io.ep_return[i] = ep_ret_out;
      if (io.ep_len) io.ep_len[i] = ep_len_out;
    }
    if (P.flags & AMENV_FLAG_AUTO_RESET) {
      if (!have_episode) e.episode = *iptr(tile, lane, AMENV_I_EPISODE);
      reset_env<T, KW>(C, K, e, C.gid0 + i);
      observe<T, KW>(K, e, o);
      bits |= AMENV_INFO_WAS_RESET;
      was_reset = true;
    }
  }
  return bits;
}

// Workgroup size is a launch parameter (64..256, multiple of 64): LDS staging area = blockDim.x rows.
// The state blob holds whole tiles, so padding lanes (i >= n) of the last wave run like real
// environments on their own (valid) slots; only their outputs are masked.
template <typename T, int NROT, int KW>
__global__ __launch_bounds__(256) void step_kernel(const HotParams<T, NROT> P, const ColdParams C, void* __restrict__ blob, const StepIO io) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int BS = int(blockDim.x);
  const int i = blockIdx.x * BS + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool active = i < P.n;
  char* tile = const_cast<char*>(tile_base(blob, P.tile_bytes, i));
  const int K = KW == 1 ? 1 : P.K;
  Env<T, KW> e;
  load_env<T, KW>(K, tile, lane, e);
  const float4 a = active ? io.actions[i] : make_float4(1.0f, 0.f, 0.f, 0.f);
  T reward; float o[kObsDim]; bool was_reset; int ep_len; float ep_ret;
  uint32_t bits = step_lane<T, NROT, KW>(P, C, e, a, i, active, reward, o, io, tile, lane, false, was_reset, ep_len, ep_ret);
  const bool is_done = active && (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
  store_env_step<T, KW>(tile, lane, e);
  if (was_reset) store_env_episode<T, KW>(K, tile, lane, e);
  if (active) {
    reinterpret_cast<T*>(io.reward)[i] = reward;
    io.done[i] = is_done ? 1 : 0;
    io.info[i] = bits;
  }
  stage_obs(lds + threadIdx.x * kObsDim, o);
  __syncthreads();
  const int row0 = blockIdx.x * BS;
  const int rows = min(BS, P.n - row0);
  flush_obs(lds, io.obs + size_t(row0) * kObsDim, rows);
  accumulate_stats(io.stats, bits, is_done, ep_len, ep_ret);
}

// n_steps control steps per launch with open-loop actions [T][N][4]; per-step outputs [T][N]...
// State stays in registers across steps: HBM traffic per env-step drops to action + outputs.
template <typename T, int NROT, int KW>
__global__ __launch_bounds__(256) void rollout_kernel(const HotParams<T, NROT> P, const ColdParams C, void* __restrict__ blob,
                                                      const StepIO io, int n_steps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int BS = int(blockDim.x);
  const int i = blockIdx.x * BS + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool active = i < P.n;
  const int row0 = blockIdx.x * BS;
  const int rows = min(BS, P.n - row0);
  const size_t n = size_t(P.n);
  char* tile = const_cast<char*>(tile_base(blob, P.tile_bytes, i));
  const int K = KW == 1 ? 1 : P.K;
  Env<T, KW> e;
  load_env<T, KW>(K, tile, lane, e);
  bool any_reset = false;
  StepIO io_t = io; io_t.terminal_obs = nullptr; io_t.ep_return = nullptr; io_t.ep_len = nullptr;
  for (int t = 0; t < n_steps; t++) {
    const float4 a = active ? io.actions[size_t(t) * n + i] : make_float4(1.0f, 0.f, 0.f, 0.f);
    T reward; float o[kObsDim]; bool was_reset; int ep_len; float ep_ret;
    uint32_t bits = step_lane<T, NROT, KW>(P, C, e, a, i, active, reward, o, io_t, tile, lane, any_reset, was_reset, ep_len, ep_ret);
    any_reset |= was_reset;
    const bool is_done = active && (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
    if (active) {
      if (io.reward) reinterpret_cast<T*>(io.reward)[size_t(t) * n + i] = reward;
      if (io.done) io.done[size_t(t) * n + i] = is_done ? 1 : 0;
      if (io.info) io.info[size_t(t) * n + i] = bits;
    }
    if (io.obs) {
      stage_obs(lds + threadIdx.x * kObsDim, o);
      __syncthreads();
      flush_obs(lds, io.obs + (size_t(t) * n + row0) * kObsDim, rows);
      __syncthreads();
    }
    accumulate_stats(io.stats, bits, is_done, ep_len, ep_ret);
  }
  store_env_step<T, KW>(tile, lane, e);
  if (any_reset) store_env_episode<T, KW>(K, tile, lane, e);
}

// WaypointQuadEnv.reset for masked envs (mask null = all) + observation of every env.
// Launched over whole tiles: padding lanes (i >= n) are always reset so that they hold a valid state.
template <typename T>
__global__ void reset_kernel(int n, int K, uint32_t tile_bytes, const ColdParams C, void* __restrict__ blob,
                             const uint8_t* __restrict__ mask, float* __restrict__ obs, int pad_only) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool active = i < n;
  char* tile = const_cast<char*>(tile_base(blob, tile_bytes, i));
  Env<T, AMENV_MAX_WAYPOINTS> e;
  load_env<T, AMENV_MAX_WAYPOINTS>(K, tile, lane, e);
  const bool do_reset = active ? (!pad_only && (!mask || mask[i])) : true;  // pad_only: amenv_create's init pass
  if (do_reset) {
    e.episode = *iptr(tile, lane, AMENV_I_EPISODE);
    reset_env<T, AMENV_MAX_WAYPOINTS>(C, K, e, C.gid0 + i);
    store_env_step<T, AMENV_MAX_WAYPOINTS>(tile, lane, e);
    store_env_episode<T, AMENV_MAX_WAYPOINTS>(K, tile, lane, e);
  }
  if (obs && active) {
    float o[kObsDim];
    observe<T, AMENV_MAX_WAYPOINTS>(K, e, o);
#pragma unroll
    for (int j = 0; j < kObsDim; j++) obs[size_t(i) * kObsDim + j] = o[j];
  }
}

// _get_observation of the current state for every env (no stepping).
template <typename T>
__global__ void observe_kernel(int n, int K, uint32_t tile_bytes, const void* __restrict__ blob, float* __restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Env<T, AMENV_MAX_WAYPOINTS> e;
  load_env<T, AMENV_MAX_WAYPOINTS>(K, tile_base(blob, tile_bytes, i), threadIdx.x & 63, e);
  float o[kObsDim];
  observe<T, AMENV_MAX_WAYPOINTS>(K, e, o);
#pragma unroll
  for (int j = 0; j < kObsDim; j++) obs[size_t(i) * kObsDim + j] = o[j];
}

// amenv_get_state / amenv_set_state: the public struct-of-arrays view (fstate [NF][N] T, istate [4][N] i32)
// <-> the internal tile layout.  to_api != 0: tiles -> SoA; else SoA -> tiles.
template <typename T>
__global__ void transpose_state_kernel(int n, int nf, uint32_t tile_bytes, void* __restrict__ blob, T* __restrict__ fapi,
                                       int32_t* __restrict__ iapi, int to_api) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  char* tile = const_cast<char*>(tile_base(blob, tile_bytes, i));
  const int lane = threadIdx.x & 63;
  if (fapi)
    for (int f = 0; f < nf; f++) {
      if (to_api) fapi[size_t(f) * n + i] = *fptr<T>(tile, lane, f);
      else *fptr<T>(tile, lane, f) = fapi[size_t(f) * n + i];
    }
  if (iapi)
    for (int f = 0; f < AMENV_I_NFIELDS; f++) {
      if (to_api) iapi[size_t(f) * n + i] = *iptr(tile, lane, f);
      else *iptr(tile, lane, f) = iapi[size_t(f) * n + i];
    }
}

}  // namespace amenv_dev
