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


// ---- fused PPO loss and its gradient (SB3 PPO.train, the part between the network outputs and the backward pass) ----------
// torch expresses it as ~60 elementwise / reduction kernels per minibatch (advantage normalisation, Gaussian log-prob, ratio,
// clipped surrogate, value MSE, entropy and the backward of each), ~5 us apiece inside a replayed graph = a quarter of the update
// at 65,536 samples.  Here: three launches.
//   1. ppo_adv_partials : per-block fp64 sum / sum of squares of the raw advantages
//   2. ppo_loss_grad    : every block reduces those partials itself (fixed order) -> mean, 1/(std + 1e-8) (unbiased std, as
//                         torch.std); one lane per sample: d loss / d mean[i,:], d loss / d value[i]; per-block partials of
//                         d loss / d log_std[:] and of the four reported scalars
//   3. ppo_finalize     : fixed-order sum of the per-block partials (deterministic, no atomics)
// Loss (SB3 2.6.0): L = mean(-min(A r, A clip(r, 1-c, 1+c))) + ent_coef * (-mean(entropy)) + vf_coef * mean((R - V)^2),
// r = exp(logp - logp_old), A normalised per minibatch.  torch.min's backward splits ties evenly between its arguments and
// clamp passes gradient on the closed interval, which together give: d/dr = -A where r is inside [1-c, 1+c] or where the
// unclipped term is the smaller one, else 0.
constexpr int kPpoBlock = 256;
constexpr int kPpoMaxBlocks = 1024;
constexpr int kPpoPartial = AMENV_MAX_JOINTS + kActDim + 4;   // d log_std[<=7] + policy / value / clip-fraction sums (+1 spare)

// body of ppo_adv_partials for block `blk` of `nblk` (also called from the fused minibatch step's prologue kernel, amenv_mlp_train.hpp)
__device__ __forceinline__ void ppo_adv_partials_block(const float* __restrict__ adv, int64_t n, double* __restrict__ part, const int64_t* __restrict__ index,
                                                       int blk, int nblk) {
  __shared__ double sh[2][kPpoBlock / 64];
  double s = 0.0, q = 0.0;
  for (int64_t i = int64_t(blk) * kPpoBlock + threadIdx.x; i < n; i += int64_t(nblk) * kPpoBlock) {
    const double a = adv[index ? index[i] : i];
    s += a; q += a * a;
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o); q += __shfl_down(q, o); }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[0][w] = s; sh[1][w] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double S = 0.0, Q = 0.0;
    for (int k = 0; k < kPpoBlock / 64; k++) { S += sh[0][k]; Q += sh[1][k]; }
    part[2 * blk] = S; part[2 * blk + 1] = Q;
  }
}
__global__ __launch_bounds__(kPpoBlock) void ppo_adv_partials(const float* __restrict__ adv, int64_t n, double* __restrict__ part,
                                                              const int64_t* __restrict__ index = nullptr) {
  ppo_adv_partials_block(adv, n, part, index, int(blockIdx.x), int(gridDim.x));
}

template <int A>
__global__ __launch_bounds__(kPpoBlock) void ppo_loss_grad(const float* __restrict__ mean, const float* __restrict__ value,
                                                           const float* __restrict__ log_std, const float* __restrict__ actions,
                                                           const float* __restrict__ old_logp, const float* __restrict__ adv,
                                                           const float* __restrict__ ret, int64_t n, float clip, float vf_coef,
                                                           int normalize, const double* __restrict__ adv_part, int adv_blocks,
                                                           float* __restrict__ d_mean, float* __restrict__ d_value,
                                                           float* __restrict__ part) {
  __shared__ float sh[kPpoBlock / 64][kPpoPartial];
  // advantage statistics: every block sums the (few hundred) partials in the same fixed order
  float mu = 0.0f, inv_sd = 1.0f;
  if (normalize && n > 1) {
    double S = 0.0, Q = 0.0;
    for (int k = 0; k < adv_blocks; k++) { S += adv_part[2 * k]; Q += adv_part[2 * k + 1]; }
    const double m = S / double(n);
    const double var = fmax((Q - S * m) / double(n - 1), 0.0);
    mu = float(m);
    inv_sd = 1.0f / (float(sqrt(var)) + 1e-8f);
  }
  float els[A], isd[A];
#pragma unroll
  for (int k = 0; k < A; k++) { els[k] = log_std[k]; isd[k] = expf(-els[k]); }
  const float inv_n = 1.0f / float(n);
  float acc[kPpoPartial];
#pragma unroll
  for (int k = 0; k < kPpoPartial; k++) acc[k] = 0.0f;
  for (int64_t i = int64_t(blockIdx.x) * kPpoBlock + threadIdx.x; i < n; i += int64_t(gridDim.x) * kPpoBlock) {
    float z[A], lp = 0.0f;
#pragma unroll
    for (int k = 0; k < A; k++) {
      z[k] = (actions[i * A + k] - mean[i * A + k]) * isd[k];
      lp += fma_(-0.5f * z[k], z[k], -els[k]) - 0.918938533204672742f;
    }
    const float r = expf(lp - old_logp[i]);
    const float a = (adv[i] - mu) * inv_sd;
    const float s1 = a * r, s2 = a * fminf(fmaxf(r, 1.0f - clip), 1.0f + clip);
    const bool inside = r >= 1.0f - clip && r <= 1.0f + clip;
    const float g_lp = (inside || s1 < s2) ? -a * r * inv_n : 0.0f;          // d L / d logp_i
#pragma unroll
    for (int k = 0; k < A; k++) {
      d_mean[i * A + k] = g_lp * z[k] * isd[k];
      acc[k] += g_lp * fma_(z[k], z[k], -1.0f);
    }
    const float dv = ret[i] - value[i];
    d_value[i] = -2.0f * vf_coef * dv * inv_n;
    acc[A] += -fminf(s1, s2);
    acc[A + 1] += dv * dv;
    acc[A + 2] += fabsf(r - 1.0f) > clip ? 1.0f : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < A + 3; k++)
    for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_down(acc[k], o);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < A + 3; k++) sh[w][k] = acc[k];
  __syncthreads();
  if (threadIdx.x < A + 3) {
    float t = 0.0f;
    for (int k = 0; k < kPpoBlock / 64; k++) t += sh[k][threadIdx.x];
    part[blockIdx.x * kPpoPartial + threadIdx.x] = t;
  }
}

// d_log_std[k] = sum of partials + ent_coef * d(-mean entropy)/d log_std = ... - ent_coef ; stats = {policy loss, value loss,
// entropy loss, clip fraction}
__global__ void ppo_finalize(const float* __restrict__ part, int blocks, int A, int64_t n, const float* __restrict__ log_std, float ent_coef,
                             float* __restrict__ d_log_std, float* __restrict__ stats) {
  const int k = threadIdx.x;
  if (k >= A + 3) return;
  float t = 0.0f;
  for (int b = 0; b < blocks; b++) t += part[b * kPpoPartial + k];
  if (k < A) d_log_std[k] = t - ent_coef;
  else if (k == A) stats[0] = t / float(n);
  else if (k == A + 1) stats[1] = t / float(n);
  else {
    stats[3] = t / float(n);
    float e = 0.0f;
    for (int j = 0; j < A; j++) e += 1.418938533204672742f + log_std[j];
    stats[2] = -e;
  }
}

}  // namespace amenv_dev
