// amenv_quad_policy.hpp -- closed-loop rollout in ONE launch for the RIGID vehicles (the reference's quadrotor, the hexacopter): observation ->
// actor / critic MLPs -> Gaussian sample -> clip -> env step, T times (v2/rl_train.py:38-56 through SB3's collect_rollouts; the reference trains
// exactly this vehicle, runsim_scaledObs.py:15 flies its checkpoint).  The MLP part is amenv_team_policy.hpp's: a 256-thread workgroup = 16 envs,
// the two [128, 64, 64] tanh MLPs on v_mfma_f32_16x16x32_bf16 with the weight fragments resident in the four wavefronts' registers (20 inputs +
// bias column = one K step), activations through LDS, 5 barriers per step.  The env part is amenv_quad.hpp's lane-quad step (4 lanes per env, DPP):
// the workgroup's 16 envs are ONE wavefront's work, so wavefront 0 samples, steps and publishes while the other three wait at the step's barrier.
// Same Philox keying / Box-Muller mapping / clip as the lane-team form (block 0 = the four wrench entries).
#pragma once
#include "amenv_quad.hpp"
#include "amenv_team_policy.hpp"

namespace amenv_dev {

template <int NROT>
__global__ __launch_bounds__(256) void rollout_policy_kernel_quad(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, int n_steps, const PolicyIO io,
                                                                  unsigned long long* __restrict__ stats, const ColdParams C, const QuadParams P) {
  constexpr int OD = 20, AD = 4;
  __shared__ __attribute__((aligned(16))) __bf16 xin[16 * kXS];
  __shared__ __attribute__((aligned(16))) __bf16 xlo[16 * kXS];   // the observation's second bf16 part (x - bf16(x)): see layer 1
  __shared__ __attribute__((aligned(16))) __bf16 h1[2 * 16 * kH1S];
  __shared__ __attribute__((aligned(16))) __bf16 h2[2 * 16 * kH2S];
  __shared__ __attribute__((aligned(16))) __bf16 h3[2 * 16 * kH2S];
  __shared__ __attribute__((aligned(16))) float meanb[16 * 8];
  __shared__ float valb[16];
  QuadLane L;
  L.init(P);
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  const int el = L.lane >> 2;                                // env within the workgroup (wavefront 0's quads)
  const int i = int(blockIdx.x) * 16 + el;
  const bool active = i < n_envs, envw = wave == 0;
  char* tile = static_cast<char*>(blob) + size_t(i >> 6) * tile_bytes;
  // this wavefront's weight fragments and biases: loop-invariant, in registers
  uint4 wf[kPolFrags], wlo[4];                               // (wlo: the second bf16 part of this wavefront's first-layer weights)
  f32x4 bias[kPolBias];
#pragma unroll
  for (int k = 0; k < 4; k++) wlo[k] = io.pack[size_t(kPolLoBase) + size_t(4 * wave + k) * 64 + L.lane];
  {
    const uint4* src = io.pack + size_t(wave) * (kPolFrags + kPolBias) * 64 + L.lane;
#pragma unroll
    for (int k = 0; k < kPolFrags; k++) wf[k] = src[k * 64];
#pragma unroll
    for (int k = 0; k < kPolBias; k++) { const uint4 b = src[(kPolFrags + k) * 64]; bias[k] = f32x4{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)}; }
  }
  QuadEnv E;
  quad_load(tile, i, L, E);                                  // (every wavefront loads; only wavefront 0 uses and stores it)
  const uint4 ac = io.pack[size_t(4) * (kPolFrags + kPolBias) * 64 + L.lane];
  const float std_w = __uint_as_float(ac.x), ls_w = __uint_as_float(ac.y);
  const float lo_w = L.cc == 0 ? 0.0f : -1.0f, hi_w = L.cc == 0 ? 2.0f : 1.0f;
  const int64_t gid = C.gid0 + i;
  const uint32_t g_lo = uint32_t(uint64_t(gid)), g_hi = uint32_t(uint64_t(gid) >> 32);
  const size_t n = size_t(n_envs);
  const int nrow = L.lane & 15, kq = L.lane >> 4;              // MFMA roles of this lane: env column / k-quarter (operand B), neuron quarter (result D)
  // the observation row of the state in registers -> global row (fp32) and the MLP input tile (bf16): six values per lane (quad_store_obs's)
  auto publish_obs = [&](float* grow) {
    const bool v3 = !L.c3;
    const float o0 = E.P * 0.1f, o1 = E.V * 0.2f, o2 = E.Q, o3 = E.W * 0.2f, o4 = (E.WP - E.P) * 0.5f, o5 = L.c3 ? E.final_yaw * 0.31830988618379067154f : 0.0f;
    if (active) {
      if (v3) { grow[L.cc] = o0; grow[3 + L.cc] = o1; grow[10 + L.cc] = o3; grow[13 + L.cc] = o4; }
      grow[6 + L.cc] = o2; grow[16 + L.cc] = o5;
    }
    __bf16* xr = xin + el * kXS;
    __bf16* xl = xlo + el * kXS;
    auto put = [&](int col, float v) { const __bf16 hi = (__bf16)v; xr[col] = hi; xl[col] = (__bf16)(v - float(hi)); };
    if (v3) { put(L.cc, o0); put(3 + L.cc, o1); put(10 + L.cc, o3); put(13 + L.cc, o4); }
    put(6 + L.cc, o2); put(16 + L.cc, o5);
  };
  if (envw) {
    __bf16* xr = xin + el * kXS;
    __bf16* xl = xlo + el * kXS;
    if (L.lead) { xr[OD] = (__bf16)1.0f; xl[OD] = (__bf16)0.0f; }   // bias column
#pragma unroll
    for (int k = 0; k < 3; k++) { const int col = OD + 1 + 4 * k + L.cc; if (col < 32) { xr[col] = (__bf16)0.0f; xl[col] = (__bf16)0.0f; } }   // K padding 21..31
    publish_obs(io.obs + size_t(i) * OD);
  }
  auto load_b = [&](const __bf16* base, int stride, int ks) {   // operand B: 8 consecutive inputs of env `nrow`
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + nrow * stride + 32 * ks + 8 * kq));
  };
  auto store_d = [&](__bf16* base, int stride, int tile16, const f32x4& acc) {   // tanh, 4 consecutive neurons of env `nrow`
    const f32x4 t{fast_tanh(acc[0]), fast_tanh(acc[1]), fast_tanh(acc[2]), fast_tanh(acc[3])};
    *reinterpret_cast<uint2*>(base + nrow * stride + 16 * tile16 + 4 * kq) = __builtin_bit_cast(uint2, __builtin_convertvector(t, bf16x4));
  };
  bool any_reset = false;
  for (int t = 0; t < n_steps; t++) {
    __syncthreads();                                            // the observation tile is complete
    {   // layer 1: 4 tiles of the combined [actor | critic] 256 neurons, K = 32 (20 inputs + bias column + padding).  The observation enters as
        // TWO bf16 parts (x = hi + lo to ~2^-17 relative): a trained controller's first layer amplifies the 2^-9 rounding of a single bf16 input
        // to ~0.1 in the action (measured on the reference checkpoint, 0.03 with the input split); so do the first layer's weights (W1 = hi + lo):
        // three products hi.hi + hi.lo + lo.hi per tile, the first layer at ~fp32 accuracy for two more MFMAs
      const bf16x8 B = load_b(xin, kXS, 0), Bl = load_b(xlo, kXS, 0);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int T = 4 * wave + j, net = T >> 3;
        f32x4 acc{0.0f, 0.0f, 0.0f, 0.0f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wlo[j]), B, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]), Bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]), B, acc, 0, 0, 0);
        store_d(h1 + net * 16 * kH1S, kH1S, T & 7, acc);
      }
    }
    __syncthreads();
    const int net23 = wave >> 1;                                // layers 2, 3: wavefronts 0, 1 the actor, 2, 3 the critic
    {
      bf16x8 B[4];
#pragma unroll
      for (int ks = 0; ks < 4; ks++) B[ks] = load_b(h1 + net23 * 16 * kH1S, kH1S, ks);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        f32x4 acc = bias[j];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[4 + 4 * j + ks]), B[ks], acc, 0, 0, 0);
        store_d(h2 + net23 * 16 * kH2S, kH2S, (2 * wave + j) & 3, acc);
      }
    }
    __syncthreads();
    {
      bf16x8 B[2];
#pragma unroll
      for (int ks = 0; ks < 2; ks++) B[ks] = load_b(h2 + net23 * 16 * kH2S, kH2S, ks);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        f32x4 acc = bias[2 + j];
#pragma unroll
        for (int ks = 0; ks < 2; ks++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[12 + 2 * j + ks]), B[ks], acc, 0, 0, 0);
        store_d(h3 + net23 * 16 * kH2S, kH2S, (2 * wave + j) & 3, acc);
      }
    }
    __syncthreads();
    if (wave < 2) {   // heads: wavefront 0 the action mean (rows 0..3), wavefront 1 the value (row 0)
      f32x4 acc = bias[4];
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[16 + ks]), load_b(h3 + wave * 16 * kH2S, kH2S, ks), acc, 0, 0, 0);
      if (wave == 0) { if (kq == 0) *reinterpret_cast<float4*>(meanb + nrow * 8) = make_float4(acc[0], acc[1], acc[2], acc[3]); }
      else if (kq == 0) valb[nrow] = acc[0];
    }
    __syncthreads();
    if (envw) {
      // ---- sample: raw = mean + std z, logp, clip (DiagGaussianDistribution + collect_rollouts' clip); Philox block 0 = the four wrench entries
      const float mean_w = meanb[el * 8 + L.cc], value = valb[el];
      float z;
      {
        uint32_t w4[4];
        philox4x32_10(io.seed_lo ^ 0x5bd1e995u, io.seed_hi ^ 0x27d4eb2fu, g_lo, g_hi, io.draw0 + uint32_t(t), 0u, w4);
        const uint32_t wa = L.cc < 2 ? w4[0] : w4[2], wb = L.cc < 2 ? w4[1] : w4[3];
        const float u1 = float((wa >> 8) + 1u) * 5.9604644775390625e-08f, u2 = float(wb >> 8) * 5.9604644775390625e-08f;
        const float rad = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1));
        const float ang = 6.28318530717958647692f * u2;
        z = rad * ((L.cc & 1) ? __sinf(ang) : __cosf(ang));
      }
      const float raw = fma_(std_w, z, mean_w);
      const float logp = sum4(fma_(-0.5f * z, z, -ls_w) - 0.918938533204672742f);
      const float act = clamp_(raw, lo_w, hi_w);
      const size_t tn = size_t(t) * n;
      if (active) {
        io.actions[(tn + i) * AD + L.cc] = raw;
        if (L.lead) { io.logp[tn + i] = logp; io.values[tn + i] = value; }
      }
      // ---- env step (amenv_quad.hpp: the step kernel's arithmetic), episode end inline
      const QuadOut o = quad_advance<NROT>(P, L, E, act);
      accumulate_stats(stats, int(blockIdx.x), o.bits, active && o.ended && L.lead, o.ep_len, o.ep_ret);
      uint32_t bits = o.bits;
      if (o.ended) {                                            // uniform within the quad
        if (active && io.terminal_obs) quad_store_obs(L, E, io.terminal_obs + (tn + i) * OD);
        if (P.flags & AMENV_FLAG_AUTO_RESET) {
          quad_apply_reset(L, quad_reset(C, L, E.episode, i), E);
          bits |= AMENV_INFO_WAS_RESET;
          any_reset = true;
        }
      }
      if (active && L.lead) {
        io.rewards[tn + i] = o.reward;
        io.dones[tn + i] = o.ended ? 1 : 0;
        if (io.info) io.info[tn + i] = bits;
      }
      publish_obs(io.obs + (tn + n + i) * OD);                  // row t + 1, and the next step's MLP input (xin was last read before the second barrier)
    }
  }
  if (envw) quad_store(tile, i, L, E, any_reset);
}

}  // namespace amenv_dev
