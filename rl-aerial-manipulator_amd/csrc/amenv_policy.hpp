// amenv_policy.hpp -- fused forward pass of the reference's policy network for policy-in-the-loop rollouts (rows f3 / a11).
//
// SB3 `MlpPolicy` with `net_arch=[128, 64, 64]`, tanh (v2/rl_train.py:27-30): two separate trunks obs -> 128 -> 64 -> 64, a linear
// action head (mean of the diagonal Gaussian) and a linear value head.  PyTorch runs that as 14 library kernels per evaluation
// (~5-15 us each at 4096..32768 rows): next to an 8.6 us environment step the policy, not the simulator, sets the pace of a
// closed-loop rollout.  Here it is ONE launch:
//   * grid = (tiles of 64 envs, 2 nets); a workgroup of NW waves evaluates one trunk for one tile, lane = env;
//   * the weights are wave-uniform, so they travel through SCALAR loads and enter `v_fma_f32` as SGPR operands -- no weight ever
//     touches a VGPR or LDS (fp32 MFMA issues at the VALU FMA rate on gfx950, so there is nothing to gain from it at fp32; from 8,192
//     rows on ppo.py takes amenv_mlp_train.hpp's mlp_forward_kernel instead: fp32 products from bf16 MFMAs on split operands);
//   * each wave computes a 1/NW slice of every layer's outputs for its 64 envs and hands the activations to the other waves through
//     LDS ([neuron][lane], conflict-free), three barriers per evaluation.
// Parameters are read from the policy's flat buffer in SB3's state-dict order (ppo.py ActorCritic.flatten_):
//   log_std[A] | pi.0 W[128,D] b | pi.2 W[64,128] b | pi.4 W[64,64] b | vf.0 .. vf.4 (same shapes) | action W[A,64] b | value W[1,64] b
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amenv_dev {

constexpr int kPolH1 = 128, kPolH2 = 64, kPolH3 = 64;
constexpr int kPolWaves = 8;   // waves per workgroup (per trunk and tile)

template <int NIN, int NOUT_PER_WAVE>
__device__ __forceinline__ void dense_tanh_slice(const float* __restrict__ w /* rows of this wave's outputs, [.., NIN] */,
                                                 const float* __restrict__ b, const float* x, float* __restrict__ out /* lds + j0*64 + lane */) {
#pragma unroll
  for (int jj = 0; jj < NOUT_PER_WAVE; jj += 2) {   // two independent accumulators: the FMA chains interleave
    float a0 = b[jj], a1 = b[jj + 1];
#pragma unroll
    for (int i = 0; i < NIN; i++) {
      a0 = __builtin_fmaf(w[jj * NIN + i], x[i], a0);
      a1 = __builtin_fmaf(w[(jj + 1) * NIN + i], x[i], a1);
    }
    out[jj * 64] = tanhf(a0);
    out[(jj + 1) * 64] = tanhf(a1);
  }
}

template <int D, int A>
__global__ __launch_bounds__(64 * kPolWaves) void policy_forward_kernel(const float* __restrict__ P, const float* __restrict__ obs, int64_t n,
                                                                        float* __restrict__ mean, float* __restrict__ value) {
  constexpr int H1 = kPolH1, H2 = kPolH2, H3 = kPolH3, NW = kPolWaves;
  constexpr int trunk = H1 * D + H1 + H2 * H1 + H2 + H3 * H2 + H3;   // floats of one trunk
  constexpr int o_pi = A, o_vf = o_pi + trunk, o_actw = o_vf + trunk, o_actb = o_actw + A * H3, o_valw = o_actb + A, o_valb = o_valw + H3;
  __shared__ float lds[(H1 + H2) * 64];                               // h1 [H1][64] | h2 [H2][64]; h3 reuses h1's area
  const int lane = threadIdx.x & 63;
  const int q = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);   // wave in the workgroup: its slice of every layer
  const int net = blockIdx.y;                                          // 0 actor trunk, 1 critic trunk
  if ((net == 0 && !mean) || (net == 1 && !value)) return;             // whole workgroup: uniform
  const int64_t env = int64_t(blockIdx.x) * 64 + lane;
  const int64_t e = env < n ? env : n - 1;
  const float* W1 = P + (net == 0 ? o_pi : o_vf);
  const float* B1 = W1 + H1 * D;
  const float* W2 = B1 + H1;
  const float* B2 = W2 + H2 * H1;
  const float* W3 = B2 + H2;
  const float* B3 = W3 + H3 * H2;
  float* h1 = lds;
  float* h2 = lds + H1 * 64;
  float* h3 = lds;
  {
    float x[D];
#pragma unroll
    for (int i = 0; i < D; i++) x[i] = obs[e * D + i];
    constexpr int J = H1 / NW;
    dense_tanh_slice<D, J>(W1 + q * J * D, B1 + q * J, x, h1 + q * J * 64 + lane);
  }
  __syncthreads();
  {
    float x[H1];
#pragma unroll
    for (int i = 0; i < H1; i++) x[i] = h1[i * 64 + lane];
    constexpr int J = H2 / NW;
    dense_tanh_slice<H1, J>(W2 + q * J * H1, B2 + q * J, x, h2 + q * J * 64 + lane);
  }
  __syncthreads();      // every wave has finished reading h1: its area may now receive h3
  {
    float x[H2];
#pragma unroll
    for (int i = 0; i < H2; i++) x[i] = h2[i * 64 + lane];
    constexpr int J = H3 / NW;
    dense_tanh_slice<H2, J>(W3 + q * J * H2, B3 + q * J, x, h3 + q * J * 64 + lane);
  }
  __syncthreads();
  if (net == 0) {       // action head: output k on wave k (A <= NW)
    if (q < A) {
      const float* w = P + o_actw + q * H3;
      float a = P[o_actb + q];
#pragma unroll
      for (int i = 0; i < H3; i++) a = __builtin_fmaf(w[i], h3[i * 64 + lane], a);
      if (env < n) mean[env * A + q] = a;
    }
  } else if (q == 0) {  // value head
    const float* w = P + o_valw;
    float a = P[o_valb];
#pragma unroll
    for (int i = 0; i < H3; i++) a = __builtin_fmaf(w[i], h3[i * 64 + lane], a);
    if (env < n) value[env] = a;
  }
}

}  // namespace amenv_dev
