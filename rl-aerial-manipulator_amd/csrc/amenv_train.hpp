// amenv_train.hpp -- the two non-GEMM pieces of the GPU-resident PPO loop (BASELINE config 5):
//   * gae_kernel:          SB3 RolloutBuffer.compute_returns_and_advantage (stable-baselines3 2.6.0, third-party; the
//                          reference calls it through PPO.learn, v2/rl_train.py:56) with gamma = .995, lambda = .9 (:46-47)
//   * gaussian_act_kernel: DiagGaussianDistribution.sample / log_prob + the action-space clip SB3's collect_rollouts
//                          applies before env.step (action bounds v2/rl_env_scaledObs.py:20-24)
// The MLP forward/backward stays in PyTorch-ROCm (rocBLAS GEMMs); everything here is per-env elementwise work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "amenv_model.hpp"

namespace amenv_dev {

// One lane per environment walks its column of the [T, N] rollout buffer from the last step to the first:
//   delta_t = r_t + gamma * V_{t+1} * (1 - done_t) - V_t ;  A_t = delta_t + gamma * lambda * (1 - done_t) * A_{t+1}
// done_t = "the episode ended at step t" (SB3 stores the same bit shifted by one as episode_starts[t+1]; the last
// row's bit is SB3's `dones` argument), V_T = last_values.  returns = A + V.  Loads of step t-1 are issued before the
// arithmetic of step t so the dependent chain is the two FMAs, not the memory latency.
__global__ __launch_bounds__(256) void gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
                                                  const uint8_t* __restrict__ dones, const float* __restrict__ last_values,
                                                  float* __restrict__ adv, float* __restrict__ ret, int T, int64_t N,
                                                  float gamma, float lam) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float next_v = last_values[i], a = 0.0f;
  const float gl = gamma * lam;
  int64_t o = int64_t(T - 1) * N + i;
  float r = rewards[o], v = values[o];
  uint8_t d = dones[o];
  for (int t = T - 1; t >= 0; t--) {
    float r_n = 0.0f, v_n = 0.0f;
    uint8_t d_n = 0;
    if (t > 0) { r_n = rewards[o - N]; v_n = values[o - N]; d_n = dones[o - N]; }
    const float nnt = d ? 0.0f : 1.0f;
    const float delta = fma_(gamma * next_v, nnt, r) - v;
    a = fma_(gl * nnt, a, delta);
    adv[o] = a;
    ret[o] = a + v;
    next_v = v;
    r = r_n; v = v_n; d = d_n;
    o -= N;
  }
}

// One lane per environment: a_raw = mean + exp(log_std) * z, z ~ N(0, I) from Philox4x32-10 keyed by `seed` with counter
// (global env id lo, hi, draw index, block) -> Box-Muller; logp = sum_k(-z_k^2/2 - log_std_k - log(2 pi)/2);
// a_env = clip(a_raw, low, high).  Keyed by the GLOBAL env id and the caller's draw index, so the noise of an env does not
// depend on how envs are sharded over GPUs.
template <int A>
__global__ __launch_bounds__(256) void gaussian_act_kernel(const float* __restrict__ mean, const float* __restrict__ log_std,
                                                           const float* __restrict__ low, const float* __restrict__ high,
                                                           float* __restrict__ raw, float* __restrict__ clipped,
                                                           float* __restrict__ logp, int64_t n, uint32_t seed_lo, uint32_t seed_hi,
                                                           uint32_t draw, int64_t gid0) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t gid = uint64_t(gid0 + i);
  constexpr int NB = (A + 3) / 4;
  float z[NB * 4];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    uint32_t w[4];
    philox4x32_10(seed_lo ^ 0x5bd1e995u, seed_hi ^ 0x27d4eb2fu, uint32_t(gid), uint32_t(gid >> 32), draw, uint32_t(b), w);
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const float u1 = float((w[2 * p] >> 8) + 1u) * 5.9604644775390625e-08f;      // (0, 1]
      const float u2 = float(w[2 * p + 1] >> 8) * 5.9604644775390625e-08f;         // [0, 1)
      const float rad = sqrtf(-2.0f * logf(u1));
      float s, c;
      sincosf(6.28318530717958647692f * u2, &s, &c);
      z[4 * b + 2 * p] = rad * c;
      z[4 * b + 2 * p + 1] = rad * s;
    }
  }
  float lp = 0.0f;
#pragma unroll
  for (int k = 0; k < A; k++) {
    const float ls = log_std[k];
    const float a = fma_(expf(ls), z[k], mean[i * A + k]);
    raw[i * A + k] = a;
    clipped[i * A + k] = fminf(fmaxf(a, low[k]), high[k]);
    lp += fma_(-0.5f * z[k], z[k], -ls) - 0.918938533204672742f;
  }
  logp[i] = lp;
}

}  // namespace amenv_dev
