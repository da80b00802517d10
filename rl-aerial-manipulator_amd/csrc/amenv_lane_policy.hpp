// amenv_lane_policy.hpp -- the closed-loop rollout of amenv_team_policy.hpp for LARGE batches (amenv_rollout_policy above the lane-team
// kernel's range): observation -> actor / critic MLPs -> Gaussian sample -> clip -> env step, T times in one launch, with ONE LANE PER ENV
// for the environment.  The lane-team form spends 16 lanes on an env to be short; from ~8192 envs on every SIMD holds several of its
// wavefronts and the step is bound by their vector-ALU work (~360 wave-instructions per env-step against ~60 for a lane that owns an env).
//   * a workgroup = EW x 64 envs (EW = 1 or 2 env wavefronts, 320 or 384 threads): wavefronts 4.. integrate them (the lane kernel's
//     step_lane: same arithmetic as amenv_step's lane / helper kernels, bit for bit), wavefronts 0..3 run the two MLPs on the matrix
//     cores exactly as the team kernel does (same packed fragments in registers, bf16 inputs, fp32 accumulation, D = W . X^T) over the
//     workgroup's 4 EW column tiles of 16 envs.  The env code needs ~210 registers, so a SIMD holds two wavefronts and a CU one
//     workgroup: EW = 2 covers 32768 envs with one wave of workgroups on the 256 CUs;
//   * the two halves alternate: five workgroup barriers per step, activations / action means / values through LDS;
//   * the action noise is drawn with the team kernel's Philox keying and Box-Muller mapping, the log-probability is summed in its order:
//     for the same observation both kernels produce the same action, bit for bit.
#pragma once
#include "amenv_kernels.hpp"
#include "amenv_team_policy.hpp"

namespace amenv_dev {

template <int NROT, int EW>
__global__ __launch_bounds__(256 + 64 * EW) void rollout_policy_kernel_lane(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, int n_steps, const PolicyIO io,
                                                                  unsigned long long* __restrict__ stats, const HotParams<float, NROT> P, const ColdParams C,
                                                                  const ArmArg<float, 3> AA) {
  constexpr int OD = 29, AD = 7, NE = 64 * EW, NT = 4 * EW, KW = 1, NJ = 3;   // envs / 16-env column tiles per workgroup
  __shared__ __attribute__((aligned(16))) __bf16 xin[NE * kXS];
  __shared__ __attribute__((aligned(16))) __bf16 h1[2 * NE * kH1S];       // layer-1 activations; layer 3's reuse the front of it (h1 is dead by then)
  __shared__ __attribute__((aligned(16))) __bf16 h2[2 * NE * kH2S];
  __shared__ __attribute__((aligned(16))) float meanb[NE * 8];
  __shared__ float valb[NE];
  __bf16* h3 = h1;
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  const int lane = int(threadIdx.x) & 63;
  if (wave < 4) {
    // ---- MLP wavefronts: this wavefront's neuron subset (as packed for the team kernel) for all four column tiles of 16 envs
    uint4 wf[kPolFrags];
    f32x4 bias[kPolBias];
    {
      const uint4* src = io.pack + size_t(wave) * (kPolFrags + kPolBias) * 64 + lane;
#pragma unroll
      for (int k = 0; k < kPolFrags; k++) wf[k] = src[k * 64];
#pragma unroll
      for (int k = 0; k < kPolBias; k++) { const uint4 b = src[(kPolFrags + k) * 64]; bias[k] = f32x4{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)}; }
    }
    const int nrow = lane & 15, kq = lane >> 4;
    auto load_b = [&](const __bf16* base, int stride, int et, int ks) {   // operand B: 8 consecutive inputs of env 16 et + nrow
      return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + (16 * et + nrow) * stride + 32 * ks + 8 * kq));
    };
    auto store_d = [&](__bf16* base, int stride, int et, int tile16, const f32x4& acc) {   // tanh, 4 consecutive neurons of that env
      const f32x4 t{fast_tanh(acc[0]), fast_tanh(acc[1]), fast_tanh(acc[2]), fast_tanh(acc[3])};
      *reinterpret_cast<uint2*>(base + (16 * et + nrow) * stride + 16 * tile16 + 4 * kq) = __builtin_bit_cast(uint2, __builtin_convertvector(t, bf16x4));
    };
    const int net23 = wave >> 1;                                  // layers 2, 3: wavefronts 0, 1 the actor, 2, 3 the critic
    for (int t = 0; t < n_steps; t++) {
      __syncthreads();                                            // (B0) the observation tile is complete
#pragma unroll
      for (int et = 0; et < NT; et++) {                            // layer 1: 4 tiles of the combined [actor | critic] 256 neurons, K = 32
        const bf16x8 B = load_b(xin, kXS, et, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int T = 4 * wave + j, net = T >> 3;
          f32x4 acc{0.0f, 0.0f, 0.0f, 0.0f};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]), B, acc, 0, 0, 0);
          store_d(h1 + net * NE * kH1S, kH1S, et, T & 7, acc);
        }
      }
      __syncthreads();                                            // (B1)
#pragma unroll
      for (int et = 0; et < NT; et++) {
        bf16x8 B[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) B[ks] = load_b(h1 + net23 * NE * kH1S, kH1S, et, ks);
#pragma unroll
        for (int j = 0; j < 2; j++) {
          f32x4 acc = bias[j];
#pragma unroll
          for (int ks = 0; ks < 4; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[4 + 4 * j + ks]), B[ks], acc, 0, 0, 0);
          store_d(h2 + net23 * NE * kH2S, kH2S, et, (2 * wave + j) & 3, acc);
        }
      }
      __syncthreads();                                            // (B2) h1 has been read by everyone: layer 3 may overwrite it
#pragma unroll
      for (int et = 0; et < NT; et++) {
        bf16x8 B[2];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) B[ks] = load_b(h2 + net23 * NE * kH2S, kH2S, et, ks);
#pragma unroll
        for (int j = 0; j < 2; j++) {
          f32x4 acc = bias[2 + j];
#pragma unroll
          for (int ks = 0; ks < 2; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[12 + 2 * j + ks]), B[ks], acc, 0, 0, 0);
          store_d(h3 + net23 * NE * kH2S, kH2S, et, (2 * wave + j) & 3, acc);
        }
      }
      __syncthreads();                                            // (B3)
      if (wave < 2) {   // heads: wavefront 0 the action mean (rows 0..6), wavefront 1 the value (row 0)
#pragma unroll
        for (int et = 0; et < NT; et++) {
          f32x4 acc = bias[4];
#pragma unroll
          for (int ks = 0; ks < 2; ks++)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[16 + ks]), load_b(h3 + wave * NE * kH2S, kH2S, et, ks), acc, 0, 0, 0);
          if (wave == 0) { if (kq < 2) *reinterpret_cast<float4*>(meanb + (16 * et + nrow) * 8 + 4 * kq) = make_float4(acc[0], acc[1], acc[2], acc[3]); }
          else if (kq == 0) valb[16 * et + nrow] = acc[0];
        }
      }
      __syncthreads();                                            // (B4) means and values are there
    }
    return;
  }
  // ---- env wavefronts: one lane per env
  const int el = (wave - 4) * 64 + lane;                          // env within the workgroup
  const int i = int(blockIdx.x) * NE + el;
  const bool active = i < n_envs;
  const size_t n = size_t(n_envs);
  char* tile = static_cast<char*>(blob) + size_t(i >> 6) * tile_bytes;
  Env<float, KW> e;
  load_env<float, KW, NJ>(1, tile, lane, e);
  // per-entry action constants (policy_pack_kernel's last block: lanes 0..3 hold entry c of the wrench and joint min(c, 2))
  float std_a[AD], ls_a[AD];
  {
    const uint4* ac = io.pack + size_t(4) * (kPolFrags + kPolBias) * 64;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint4 v = ac[c];
      std_a[c] = __uint_as_float(v.x); ls_a[c] = __uint_as_float(v.y);
      if (c < 3) { std_a[4 + c] = __uint_as_float(v.z); ls_a[4 + c] = __uint_as_float(v.w); }
    }
  }
  const int64_t gid = C.gid0 + i;
  const uint32_t g_lo = uint32_t(uint64_t(gid)), g_hi = uint32_t(uint64_t(gid) >> 32);
  const bool ee_task = P.ee_task != 0;
  float o[kObsDimMax];
  update_tool_offset<float, KW, false>(AA.p, e);
  observe<float, KW, true>(1, e, o, ee_task);
  observe_joints<float, KW>(e, o);
  auto publish_obs = [&](float* grow) {                          // observation row -> rollout buffer and (bf16) the MLP's input tile
    if (active) {
#pragma unroll
      for (int j = 0; j < OD; j++) grow[j] = o[j];
    }
    __bf16* xr = xin + el * kXS;
#pragma unroll
    for (int j = 0; j < OD; j++) xr[j] = (__bf16)o[j];
  };
  { __bf16* xr = xin + el * kXS; xr[OD] = (__bf16)1.0f; xr[OD + 1] = (__bf16)0.0f; xr[OD + 2] = (__bf16)0.0f; }   // bias column, K padding
  publish_obs(io.obs + size_t(i) * OD);
  bool any_reset = false;
  StepIO sio{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stats};
  for (int t = 0; t < n_steps; t++) {
    __syncthreads();   // (B0) observation tile published
    __syncthreads();   // (B1)
    __syncthreads();   // (B2)
    __syncthreads();   // (B3)
    __syncthreads();   // (B4) action means and value of this lane's env are in LDS
    float mean[8];
    {
      const float4 m0 = *reinterpret_cast<const float4*>(meanb + el * 8), m1 = *reinterpret_cast<const float4*>(meanb + el * 8 + 4);
      mean[0] = m0.x; mean[1] = m0.y; mean[2] = m0.z; mean[3] = m0.w; mean[4] = m1.x; mean[5] = m1.y; mean[6] = m1.z; mean[7] = m1.w;
    }
    const float value = valb[el];
    // ---- sample: raw = mean + std z, logp, clip (DiagGaussianDistribution + collect_rollouts' clip); Philox block 0 -> wrench entries,
    // block 1 -> joints, pairs (w0, w1) -> entries 0 (cos), 1 (sin), (w2, w3) -> entries 2, 3: the team kernel's mapping
    float z[8];
#pragma unroll
    for (int b = 0; b < 2; b++) {
      uint32_t w4[4];
      philox4x32_10(io.seed_lo ^ 0x5bd1e995u, io.seed_hi ^ 0x27d4eb2fu, g_lo, g_hi, io.draw0 + uint32_t(t), uint32_t(b), w4);
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const float u1 = float((w4[2 * p] >> 8) + 1u) * 5.9604644775390625e-08f, u2 = float(w4[2 * p + 1] >> 8) * 5.9604644775390625e-08f;
        const float rad = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1));
        const float ang = 6.28318530717958647692f * u2;
        z[4 * b + 2 * p] = rad * __cosf(ang);
        z[4 * b + 2 * p + 1] = rad * __sinf(ang);
      }
    }
    float act[AD], raw[AD], lp[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      raw[c] = fma_(std_a[c], z[c], mean[c]);
      lp[c] = fma_(-0.5f * z[c], z[c], -ls_a[c]) - 0.918938533204672742f;
      act[c] = clamp_(raw[c], c == 0 ? 0.0f : -1.0f, c == 0 ? 2.0f : 1.0f);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      raw[4 + c] = fma_(std_a[4 + c], z[4 + c], mean[4 + c]);
      lp[c] = lp[c] + (fma_(-0.5f * z[4 + c], z[4 + c], -ls_a[4 + c]) - 0.918938533204672742f);
      act[4 + c] = clamp_(raw[4 + c], -1.0f, 1.0f);
    }
    const float logp = (lp[0] + lp[1]) + (lp[2] + lp[3]);       // (the team kernel's quad sum, same association)
    const size_t tn = size_t(t) * n;
    if (active) {
      float* ar = io.actions + (tn + i) * AD;
#pragma unroll
      for (int c = 0; c < AD; c++) ar[c] = raw[c];
      io.logp[tn + i] = logp; io.values[tn + i] = value;
    }
    // ---- env step (amenv_step's lane kernel code)
    sio.terminal_obs = io.terminal_obs ? io.terminal_obs + tn * OD : nullptr;
    float reward; bool was_reset; int ep_len; float ep_ret;
    const uint32_t bits = step_lane<float, NROT, KW, VAR_V2, NJ>(P, C, AA, e, act, i, active, reward, o, sio, tile, lane, any_reset, was_reset, ep_len, ep_ret);
    any_reset |= was_reset;
    const bool is_done = active && (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
    accumulate_stats(stats, int(blockIdx.x) * EW + (wave - 4), bits, is_done, ep_len, ep_ret);
    if (active) {
      io.rewards[tn + i] = reward;
      io.dones[tn + i] = is_done ? 1 : 0;
      if (io.info) io.info[tn + i] = bits;
    }
    publish_obs(io.obs + (tn + n + i) * OD);                     // row t + 1, and the next step's MLP input
  }
  store_env_step<float, KW, NJ>(tile, lane, e);
  if (any_reset) store_env_episode<float, KW>(1, tile, lane, e);
}

}  // namespace amenv_dev
