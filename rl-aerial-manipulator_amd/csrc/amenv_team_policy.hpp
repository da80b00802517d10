// amenv_team_policy.hpp -- closed-loop rollout in ONE launch: observation -> actor / critic MLPs -> Gaussian sample -> clip -> env step,
// T times, for the hexacopter + arm on the lane-team layout (amenv_team.hpp).  This is the loop SB3's collect_rollouts runs
// (v2/rl_train.py:38-56: MlpPolicy [128, 64, 64] tanh, separate actor / critic trunks) with nothing leaving the chip between steps:
//   * a 256-thread workgroup = 4 wavefronts = 16 envs (16 lanes per env); env state and the per-lane constants stay in registers;
//   * the two MLPs run on the matrix cores in bf16 with fp32 accumulation (v_mfma_f32_16x16x32_bf16): this IS a dense contraction --
//     north_star's "no MFMA" is about step().  Per step and workgroup: 18 MFMAs per wavefront.  Orientation D = W . X^T: M = 16 output
//     neurons, N = the workgroup's 16 envs, K = inputs, so a lane ends up with 4 consecutive neurons of one env (one 8-byte LDS store)
//     and the next layer's B operand is 8 consecutive inputs of one env (one 16-byte LDS load).  The weights are loop-invariant A
//     operands: each wavefront keeps its 18 fragments (72 VGPRs) in registers for the whole rollout -- no weight traffic per step;
//   * activations travel through LDS (bf16), 5 workgroup barriers per step; tanh = 1 - 2 / (exp(2x) + 1) on v_exp / v_rcp;
//   * action noise: Philox4x32-10 keyed (seed, global env id, draw index, block) as amenv_gaussian_act, Box-Muller on the fast
//     intrinsics (v_log / v_sin / v_cos): the same distribution, not the same bits as the one-launch-per-op path.
// Above the lane-team kernel's batch range the same loop runs with one lane per env for the environment: amenv_lane_policy.hpp.
// bf16 rounding of observations / activations / weights perturbs the action means by ~1e-2 of their scale: an opt-in ROLLOUT mode
// (ppo.py fused_rollout=True); evaluate_policy / parity paths keep the fp32 kernels.
#pragma once
#include "amenv_team.hpp"

namespace amenv_dev {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPolFrags = 18, kPolBias = 5;                  // per wavefront: weight fragments (uint4 per lane), bias quadruples (float4 per lane)
constexpr int kPolLoBase = (4 * (kPolFrags + kPolBias) + 1) * 64;          // uint4 index of the first-layer residual fragments (rigid-vehicle rollout)
constexpr int kPolPackWords = (kPolLoBase + 4 * 4 * 64) * 4;                // dwords of the packed policy: 4 wavefronts + one block of action constants + W1 residuals
constexpr int kXS = 40, kH1S = 136, kH2S = 72;               // LDS row strides (bf16 elements): 32 / 128 / 64 + 8 of padding

// Flat parameter buffer (SB3 state-dict order, ppo.py ActorCritic.flatten_):
//   log_std[A] | pi.0 W[128,D] b | pi.2 W[64,128] b | pi.4 W[64,64] b | vf.0 .. vf.4 (same shapes) | action W[A,64] b | value W[1,64] b
struct PolLayout {
  int D, A, trunk, o_pi, o_vf, o_actw, o_actb, o_valw, o_valb;
  __host__ __device__ PolLayout(int d, int a) : D(d), A(a) {
    trunk = 128 * D + 128 + 64 * 128 + 64 + 64 * 64 + 64;
    o_pi = A; o_vf = o_pi + trunk; o_actw = o_vf + trunk; o_actb = o_actw + A * 64; o_valw = o_actb + A; o_valb = o_valw + 64;
  }
};

__device__ __forceinline__ uint32_t bf16_bits(float x) { return uint32_t(__builtin_bit_cast(unsigned short, (__bf16)x)); }

// fp32 parameters -> MFMA A-operand fragments (bf16) + bias quadruples, in the order the rollout kernel's wavefronts consume them.
// One thread per (wavefront, item, lane); item < 18: fragment, else bias.
__global__ void policy_pack_kernel(const float* __restrict__ Pm, int D, int A, uint32_t* __restrict__ out) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int per_wave = (kPolFrags + kPolBias) * 64;
  if (tid >= 4 * per_wave + 64 + 4 * 4 * 64) return;
  if (tid >= 4 * per_wave + 64) {   // second bf16 part of the first layer's weights, W1 - bf16(W1) (and of its bias column): fragment (wavefront w, tile j) at kPolLoBase + (4 w + j) * 64
    const int r = tid - (4 * per_wave + 64), w = r / 256, item = (r % 256) / 64, l = r & 63, row = l & 15, kq = l >> 4;
    const PolLayout Lo(D, A);
    const int T = 4 * w + item, net = T >> 3, neuron = 16 * (T & 7) + row;
    const float* W1 = Pm + (net == 0 ? Lo.o_pi : Lo.o_vf); const float* b1 = W1 + 128 * D;
    uint32_t o[4];
    for (int q = 0; q < 4; q++) {
      uint32_t two[2];
      for (int h = 0; h < 2; h++) {
        const int k = 8 * kq + 2 * q + h;
        const float v = k < D ? W1[neuron * D + k] : (k == D ? b1[neuron] : 0.0f);
        two[h] = bf16_bits(v - float((__bf16)v));
      }
      o[q] = two[0] | (two[1] << 16);
    }
    uint32_t* dst = out + (size_t(kPolLoBase) + size_t(4 * w + item) * 64 + l) * 4;
    dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
    return;
  }
  if (tid >= 4 * per_wave) {   // per-lane action constants: {std, log_std} of the wrench entry of lane c and of joint min(c, 2)
    const int l = tid - 4 * per_wave, cc = l & 3, cj = cc < 3 ? cc : 2;
    uint32_t* dst = out + size_t(4 * per_wave + l) * 4;
    dst[0] = __float_as_uint(expf(Pm[cc])); dst[1] = __float_as_uint(Pm[cc]);
    dst[2] = A > 4 ? __float_as_uint(expf(Pm[4 + cj])) : 0u; dst[3] = A > 4 ? __float_as_uint(Pm[4 + cj]) : 0u;   // (a rigid vehicle has no joint actions)
    return;
  }
  const int w = tid / per_wave, item = (tid % per_wave) / 64, l = tid & 63;
  const PolLayout Lo(D, A);
  const int row = l & 15, kq = l >> 4;
  uint32_t o[4] = {0, 0, 0, 0};
  auto trunk_ptr = [&](int net) { return Pm + (net == 0 ? Lo.o_pi : Lo.o_vf); };
  if (item < kPolFrags) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = 0.0f;
    if (item < 4) {                                   // layer 1: combined tile 4w + item, K = D inputs + the bias column (obs column D = 1)
      const int T = 4 * w + item, net = T >> 3, neuron = 16 * (T & 7) + row;
      const float* W1 = trunk_ptr(net); const float* b1 = W1 + 128 * D;
      for (int j = 0; j < 8; j++) { const int k = 8 * kq + j; v[j] = k < D ? W1[neuron * D + k] : (k == D ? b1[neuron] : 0.0f); }
    } else if (item < 12) {                           // layer 2: tile 2w + j, k-step ks
      const int j2 = (item - 4) >> 2, ks = (item - 4) & 3, T = 2 * w + j2, net = T >> 2, neuron = 16 * (T & 3) + row;
      const float* W2 = trunk_ptr(net) + 128 * D + 128;
      for (int j = 0; j < 8; j++) v[j] = W2[neuron * 128 + 32 * ks + 8 * kq + j];
    } else if (item < 16) {                           // layer 3
      const int j3 = (item - 12) >> 1, ks = (item - 12) & 1, T = 2 * w + j3, net = T >> 2, neuron = 16 * (T & 3) + row;
      const float* W3 = trunk_ptr(net) + 128 * D + 128 + 64 * 128 + 64;
      for (int j = 0; j < 8; j++) v[j] = W3[neuron * 64 + 32 * ks + 8 * kq + j];
    } else {                                          // heads: wavefront 0 the action mean (A rows), wavefront 1 the value (1 row)
      const int ks = item - 16;
      for (int j = 0; j < 8; j++) {
        const int k = 32 * ks + 8 * kq + j;
        v[j] = w == 0 ? (row < A ? Pm[Lo.o_actw + row * 64 + k] : 0.0f) : (w == 1 ? (row == 0 ? Pm[Lo.o_valw + k] : 0.0f) : 0.0f);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) o[q] = bf16_bits(v[2 * q]) | (bf16_bits(v[2 * q + 1]) << 16);
  } else {                                            // biases of the D tile this lane accumulates: neurons 16 tile + 4 (l >> 4) + r
    const int bi = item - kPolFrags;
    float b[4] = {0, 0, 0, 0};
    for (int r = 0; r < 4; r++) {
      const int nn = 4 * kq + r;
      if (bi < 2) { const int T = 2 * w + bi, net = T >> 2; b[r] = (trunk_ptr(net) + 128 * D + 128 + 64 * 128)[16 * (T & 3) + nn]; }
      else if (bi < 4) { const int T = 2 * w + (bi - 2), net = T >> 2; b[r] = (trunk_ptr(net) + 128 * D + 128 + 64 * 128 + 64 + 64 * 64)[16 * (T & 3) + nn]; }
      else b[r] = w == 0 ? (nn < A ? Pm[Lo.o_actb + nn] : 0.0f) : (w == 1 ? (nn == 0 ? Pm[Lo.o_valb] : 0.0f) : 0.0f);
    }
    for (int q = 0; q < 4; q++) o[q] = __float_as_uint(b[q]);
  }
  uint32_t* dst = out + (size_t(w) * (kPolFrags + kPolBias) + item) * 64 * 4 + l * 4;
  dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
}

struct PolicyIO {
  const uint4* pack;        // policy_pack_kernel output
  uint32_t seed_lo, seed_hi, draw0;
  float* obs;               // [T + 1][N][29]: row 0 = observation of the state at entry, row t + 1 = after step t
  float* actions;           // [T][N][7]  raw (unclipped) samples, as SB3's rollout buffer stores them
  float* logp;              // [T][N]
  float* values;            // [T][N]
  float* rewards;           // [T][N]
  uint8_t* dones;           // [T][N]
  uint32_t* info;           // [T][N] or null
  float* terminal_obs;      // [T][N][29] or null: rows written only where the episode ended at step t
};

__device__ __forceinline__ float fast_tanh(float x) {   // 1 - 2 / (exp(2x) + 1): exact limits, ~1e-6 abs elsewhere
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// OCC = wavefronts per SIMD the register allocation leaves room for: 1 in the latency regime (one workgroup per CU: all 512 registers,
// no spills), 2 in the throughput regime (<= 256 registers, a few spills, twice the resident wavefronts per SIMD).
template <int NROT, int OCC>
__global__ __launch_bounds__(256, OCC) void rollout_policy_kernel_team(void* __restrict__ blob, uint32_t tile_bytes, int32_t n_envs, int n_steps, const PolicyIO io,
                                                                  unsigned long long* __restrict__ stats, const ColdParams C, const TeamParams P) {
  constexpr int OD = 29, AD = 7;
  __shared__ __attribute__((aligned(16))) __bf16 xin[16 * kXS];
  __shared__ __attribute__((aligned(16))) __bf16 h1[2 * 16 * kH1S];
  __shared__ __attribute__((aligned(16))) __bf16 h2[2 * 16 * kH2S];
  __shared__ __attribute__((aligned(16))) __bf16 h3[2 * 16 * kH2S];
  __shared__ __attribute__((aligned(16))) float meanb[16 * 8];
  __shared__ float valb[16];
  TeamLane L;
  L.init(P);
  const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  const int el = wave * 4 + (L.lane >> 4);                   // env within the workgroup
  const int i = int(blockIdx.x) * 16 + el;
  const bool active = i < n_envs;
  char* tile = static_cast<char*>(blob) + size_t(i >> 6) * tile_bytes;
  // this wavefront's weight fragments and biases: loop-invariant, in registers
  uint4 wf[kPolFrags];
  f32x4 bias[kPolBias];
  {
    const uint4* src = io.pack + size_t(wave) * (kPolFrags + kPolBias) * 64 + L.lane;
#pragma unroll
    for (int k = 0; k < kPolFrags; k++) wf[k] = src[k * 64];
#pragma unroll
    for (int k = 0; k < kPolBias; k++) { const uint4 b = src[(kPolFrags + k) * 64]; bias[k] = f32x4{__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)}; }
  }
  TeamEnv E;
  team_load(tile, i, L, E);
  // per-lane action constants: std / log_std of the wrench entry (lane c) and of the joint (lane c < 3), action box
  const int cj = L.cc < 3 ? L.cc : 2;
  const uint4 ac = io.pack[size_t(4) * (kPolFrags + kPolBias) * 64 + L.lane];
  const float std_w = __uint_as_float(ac.x), ls_w = __uint_as_float(ac.y), std_j = __uint_as_float(ac.z), ls_j = __uint_as_float(ac.w);
  const float lo_w = L.cc == 0 ? 0.0f : -1.0f, hi_w = L.cc == 0 ? 2.0f : 1.0f;
  const int64_t gid = C.gid0 + i;
  const uint32_t g_lo = uint32_t(uint64_t(gid)), g_hi = uint32_t(uint64_t(gid) >> 32);
  const size_t n = size_t(n_envs);
  const int nrow = L.lane & 15, kq = L.lane >> 4;              // MFMA roles of this lane: env column / k-quarter (operand B), neuron quarter (result D)
  // observation of the state at entry -> row 0 of the buffer and the MLP input tile
  float vA, vB, vC;
  team_obs_vals(P, L, E, team_tool_offset(P, L.c, E.y), vA, vB, vC);
  auto publish_obs = [&](float* grow) {
    if (active) {
      if (L.okA) grow[L.offA] = vA;
      if (L.okB) grow[L.offB] = vB;
      if (L.okC) grow[L.offC] = vC;
    }
    __bf16* xr = xin + el * kXS;
    if (L.okA) xr[L.offA] = (__bf16)vA;
    if (L.okB) xr[L.offB] = (__bf16)vB;
    if (L.okC) xr[L.offC] = (__bf16)vC;
  };
  if (L.lead) { __bf16* xr = xin + el * kXS; xr[OD] = (__bf16)1.0f; xr[OD + 1] = (__bf16)0.0f; xr[OD + 2] = (__bf16)0.0f; }   // bias column, K padding
  publish_obs(io.obs + size_t(i) * OD);
  auto load_b = [&](const __bf16* base, int stride, int ks) {   // operand B: 8 consecutive inputs of env `nrow`
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + nrow * stride + 32 * ks + 8 * kq));
  };
  auto store_d = [&](__bf16* base, int stride, int tile16, const f32x4& acc) {   // tanh, 4 consecutive neurons of env `nrow`
    const f32x4 t{fast_tanh(acc[0]), fast_tanh(acc[1]), fast_tanh(acc[2]), fast_tanh(acc[3])};
    *reinterpret_cast<uint2*>(base + nrow * stride + 16 * tile16 + 4 * kq) = __builtin_bit_cast(uint2, __builtin_convertvector(t, bf16x4));
  };
  for (int t = 0; t < n_steps; t++) {
    __syncthreads();                                            // the observation tile is complete
    {   // layer 1: 4 tiles of the combined [actor | critic] 256 neurons, K = 32 (29 inputs + bias column)
      const bf16x8 B = load_b(xin, kXS, 0);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int T = 4 * wave + j, net = T >> 3;
        f32x4 acc{0.0f, 0.0f, 0.0f, 0.0f};
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
    if (wave < 2) {   // heads: wavefront 0 the action mean (rows 0..6), wavefront 1 the value (row 0)
      f32x4 acc = bias[4];
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[16 + ks]), load_b(h3 + wave * 16 * kH2S, kH2S, ks), acc, 0, 0, 0);
      if (wave == 0) { if (kq < 2) *reinterpret_cast<float4*>(meanb + nrow * 8 + 4 * kq) = make_float4(acc[0], acc[1], acc[2], acc[3]); }
      else if (kq == 0) valb[nrow] = acc[0];
    }
    __syncthreads();
    // ---- sample: raw = mean + std z, logp, clip (DiagGaussianDistribution + collect_rollouts' clip).  Even quads draw Philox block 0
    // (wrench entries), odd quads block 1 (joints); the neighbour quad's draws arrive by one DPP row rotation.
    const float mean_w = meanb[el * 8 + L.cc], mean_j = meanb[el * 8 + 4 + cj], value = valb[el];
    float z_mine;
    {
      uint32_t w4[4];
      philox4x32_10(io.seed_lo ^ 0x5bd1e995u, io.seed_hi ^ 0x27d4eb2fu, g_lo, g_hi, io.draw0 + uint32_t(t), uint32_t(L.bb & 1), w4);
      const uint32_t wa = L.cc < 2 ? w4[0] : w4[2], wb = L.cc < 2 ? w4[1] : w4[3];
      const float u1 = float((wa >> 8) + 1u) * 5.9604644775390625e-08f, u2 = float(wb >> 8) * 5.9604644775390625e-08f;
      const float rad = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1));
      const float ang = 6.28318530717958647692f * u2;
      z_mine = rad * ((L.cc & 1) ? __sinf(ang) : __cosf(ang));
    }
    const float z_other = row_ror<4>(z_mine);
    const float zw = (L.bb & 1) ? z_other : z_mine, zj = (L.bb & 1) ? z_mine : z_other;
    const float raw_w = fma_(std_w, zw, mean_w), raw_j = fma_(std_j, zj, mean_j);
    const float lw = fma_(-0.5f * zw, zw, -ls_w) - 0.918938533204672742f;
    const float lj = L.cc < 3 ? fma_(-0.5f * zj, zj, -ls_j) - 0.918938533204672742f : 0.0f;
    const float logp = sum4(lw + lj);
    const float act = clamp_(raw_w, lo_w, hi_w), actj = clamp_(raw_j, -1.0f, 1.0f);
    const size_t tn = size_t(t) * n;
    if (active && L.q0) {
      float* ar = io.actions + (tn + i) * AD;
      ar[L.cc] = raw_w;
      if (L.cc < 3) ar[4 + L.cc] = raw_j;
      if (L.lead) { io.logp[tn + i] = logp; io.values[tn + i] = value; }
    }
    // ---- env step
    const TeamOut o = team_advance<NROT>(P, C, L, E, act, actj, i, active, io.terminal_obs ? io.terminal_obs + tn * OD : nullptr, nullptr, nullptr);
    accumulate_stats(stats, int(blockIdx.x) * 4 + wave, o.bits, active && o.ended && L.lead, o.ep_len, o.ep_ret);
    if (active && L.lead) {
      io.rewards[tn + i] = o.reward;
      io.dones[tn + i] = o.ended ? 1 : 0;
      if (io.info) io.info[tn + i] = o.bits;
    }
    vA = o.vA; vB = o.vB; vC = o.vC;
    publish_obs(io.obs + (tn + n + i) * OD);                   // row t + 1, and the next step's MLP input (xin was last read before the second barrier)
  }
  team_store(tile, i, L, E);
}

}  // namespace amenv_dev
