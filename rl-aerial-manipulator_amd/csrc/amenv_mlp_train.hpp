// amenv_mlp_train.hpp -- one PPO minibatch step of the reference's policy (v2/rl_train.py:38-56: SB3 MlpPolicy, net_arch [128, 64, 64],
// tanh, separate actor / critic trunks) as ONE kernel: forward of both MLPs, SB3's loss and its gradient, backward through both MLPs and
// all weight / bias gradients.  PyTorch ran this as ~90 library launches per minibatch (~0.9 ms for 65,536 samples, ~15 TFLOP/s).
//
// Arithmetic: fp32 in, fp32 out on the matrix cores -- v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fp32 fma chain, so the result
// differs from torch's fp32 modules only by summation order (test: <= 2e-5 of the largest gradient entry).
//
// Data flow of one wavefront = one tile of 32 samples (blockIdx.y = net: 0 actor, 1 critic):
//   * orientation D = W . X^T: M = 32 output neurons, N = the tile's 32 samples, K = inputs.  The accumulator tile of layer l (rows =
//     neurons in the 16 registers, column = sample on the lane) IS the B operand of layer l + 1: k-step (tile t, register r) consumes,
//     in lane half h, the neuron 32 t + (r & 3) + 8 (r >> 2) + 4 h -- exactly the row that register holds.  Activations never leave
//     registers on the forward pass nor on the data-gradient pass (dH = W^T dZ has the same shape); tanh and tanh' are lane-local.
//   * weights are the A operands, one coalesced 256-byte global load per MFMA (L2-resident: 66 KB per net): the forward reads the
//     k-major copies W^T (made by mlp_transpose_kernel after every optimiser step), the data-gradient pass the row-major originals.
//   * weight gradients contract over SAMPLES: the four wavefronts of a workgroup pool their tiles -- dZ and the layer inputs go through
//     an LDS transpose ([neuron][128 samples + 1]: written lane = sample, read lane = neuron, both conflict-free) -- and every 32 x 32
//     tile of every dW is OWNED by one wavefront, which accumulates it in registers over all samples the workgroup sees (64 k-steps per
//     iteration).  No atomics: a first version added per-wavefront tiles into an LDS accumulator with ds_add_f32 and spent 76 % of its
//     time there.  Bias gradients ride along as the running sum of the A fragments.  At the end a workgroup writes its tiles to a partial
//     slab; mlp_grad_reduce_kernel sums the slabs in a fixed order into the flat gradient: the step is deterministic.
//   * the loss part is SB3's (amenv_train.hpp ppo_loss_grad, same expressions), evaluated on the accumulator tile of the head: the 7
//     action means of a sample sit in registers 0..3 of the two lane halves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "amenv_train.hpp"

namespace amenv_dev {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kH1 = 128, kH2 = 64, kH3 = 64;
// per net: W1^T [32][128] | W2^T [128][64] | W3^T [64][64] | W4^T [64][32] | W4 padded to 32 rows [32][64]  (zeros where a row / column does not exist)
constexpr int kMlpWtPerNet = 32 * kH1 + kH1 * kH2 + kH2 * kH3 + kH3 * 32 + 32 * kH3;
// per-workgroup gradient accumulator (floats): dW1 [128][33] (col 32 = bias) | dW2 [64][129] | dW3 [64][65] | dW4 [32][65] | stats [16]
constexpr int kAccW1 = 0, kAccW2 = kAccW1 + kH1 * 33, kAccW3 = kAccW2 + kH2 * 129, kAccW4 = kAccW3 + kH3 * 65, kAccStats = kAccW4 + 32 * 65;
constexpr int kAccSize = kAccStats + 16;
constexpr int kXs = 129;                                                      // transpose row stride: 4 wavefronts x 32 samples + 1
constexpr int kXposeRows = 192;                                               // largest layer: dZ2 (64 rows) + H1 (128 rows)
constexpr int kMlpMaxBlocks = 256;

__host__ __device__ constexpr int mlp_rowmap(int r) { return (r & 3) + 8 * (r >> 2); }   // row of accumulator register r (+ 4 for the upper lane half)

__device__ __forceinline__ float tanh_acc(float x) {   // 1 - 2 / (exp(2x) + 1): exact limits, |err| ~1e-7 absolute (three VALU + two transcendental
  // instructions; a variant with a series around 0 for full RELATIVE accuracy cost 14 and bought nothing the gradient test can see)
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return fma_(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
}

// trunk / head parameter offsets inside the flat buffer (SB3 order, see amenv_team_policy.hpp PolLayout)
struct MlpNet {
  const float *W1, *b1, *W2, *b2, *W3, *b3, *W4, *b4;   // row-major originals
  int n_out;                                             // head rows: A (actor) or 1 (critic)
};
__device__ __forceinline__ MlpNet mlp_net(const float* Pm, int D, int A, int net) {
  const int trunk = kH1 * D + kH1 + kH2 * kH1 + kH2 + kH3 * kH2 + kH3;
  const float* t = Pm + A + net * trunk;
  MlpNet N;
  N.W1 = t; N.b1 = t + kH1 * D; N.W2 = N.b1 + kH1; N.b2 = N.W2 + kH2 * kH1; N.W3 = N.b2 + kH2; N.b3 = N.W3 + kH3 * kH2;
  const float* heads = Pm + A + 2 * trunk;
  N.W4 = net == 0 ? heads : heads + A * kH3 + A; N.b4 = N.W4 + (net == 0 ? A : 1) * kH3;
  N.n_out = net == 0 ? A : 1;
  return N;
}

// k-major copies of the four weight matrices of both nets (forward-pass A operands): one thread per element
__device__ __forceinline__ void mlp_transpose_element(const float* __restrict__ Pm, int D, int A, float* __restrict__ WT, int tid) {
  if (tid >= 2 * kMlpWtPerNet) return;
  const int net = tid / kMlpWtPerNet, e = tid % kMlpWtPerNet;
  const MlpNet N = mlp_net(Pm, D, A, net);
  float v;
  if (e < 32 * kH1) { const int k = e / kH1, n = e % kH1; v = k < D ? N.W1[n * D + k] : 0.0f; }
  else if (e < 32 * kH1 + kH1 * kH2) { const int q = e - 32 * kH1, k = q / kH2, n = q % kH2; v = N.W2[n * kH1 + k]; }
  else if (e < 32 * kH1 + kH1 * kH2 + kH2 * kH3) { const int q = e - 32 * kH1 - kH1 * kH2, k = q / kH3, n = q % kH3; v = N.W3[n * kH2 + k]; }
  else if (e < 32 * kH1 + kH1 * kH2 + kH2 * kH3 + kH3 * 32) { const int q = e - 32 * kH1 - kH1 * kH2 - kH2 * kH3, k = q / 32, n = q % 32; v = n < N.n_out ? N.W4[n * kH3 + k] : 0.0f; }
  else { const int q = e - 32 * kH1 - kH1 * kH2 - kH2 * kH3 - kH3 * 32, o = q / kH3, k = q % kH3; v = o < N.n_out ? N.W4[o * kH3 + k] : 0.0f; }
  WT[tid] = v;
}
__global__ void mlp_transpose_kernel(const float* __restrict__ Pm, int D, int A, float* __restrict__ WT) {
  mlp_transpose_element(Pm, D, A, WT, int(blockIdx.x * blockDim.x + threadIdx.x));
}
// Everything the fused kernel needs beforehand in ONE launch (each launch in this chain costs ~4.7 us whatever it does): workgroups
// [0, adv_blocks) sum the minibatch's advantages (ppo_adv_partials), the rest write the k-major weight copies.  256 threads.
__global__ __launch_bounds__(kPpoBlock) void ppo_mlp_prologue_kernel(const float* __restrict__ adv, int64_t n, double* __restrict__ adv_part,
                                                                     const int64_t* __restrict__ index, int adv_blocks, const float* __restrict__ Pm, int D, int A,
                                                                     float* __restrict__ WT) {
  if (int(blockIdx.x) < adv_blocks) ppo_adv_partials_block(adv, n, adv_part, index, int(blockIdx.x), adv_blocks);   // uniform per workgroup
  else mlp_transpose_element(Pm, D, A, WT, (int(blockIdx.x) - adv_blocks) * kPpoBlock + int(threadIdx.x));
}

// out[tile] = sum_k Wk[k][32 tile + n] * in[k]  (+ bias), k = 32 t + rowmap(r) + 4 h over the KT input tiles.  Wk is k-major ([K][ld]),
// zero-padded where a k or an output does not exist: the forward pass gives it W^T, the data-gradient pass the row-major W itself
// (out index = input neuron).  Every A operand is ONE load "uniform base + per-lane 32-bit offset": `voff` = (4 h ld + n) * 4 bytes is
// the only address register (passed through an empty asm by the caller once per sample tile, or hipcc hoists all ~600 loop-invariant
// 64-bit addresses of the kernel out of the tile loop: 400 spilled registers).
// Order: OUTPUT-tile major, one chunk = the k-steps of one input tile (16 MFMAs, 1024 clocks of matrix-core time -- above the L2 latency
// of the next chunk's loads, which are issued first).  A finished tile's epilogue (tanh, or the tanh derivative of the backward pass) is
// spread, element by element, between the MFMAs of the NEXT tile: the vector ALU works in the shadow of the matrix pipe instead of after
// it (a k-major version with the activations as separate loops spent half of the forward pass outside the MFMAs).
enum { kEpiNone = 0, kEpiTanh = 1, kEpiDtanh = 2 };
template <int KT, int NT, int RSTEPS, int LD, int EPI>
__device__ __forceinline__ void mlp_layer(const float* __restrict__ Wk, const float* __restrict__ bias, int n_valid, const f32x16* in, f32x16* out, uint32_t voff, int h,
                                          const f32x16* act = nullptr) {
  constexpr int CH = RSTEPS, NCH = NT * KT, SLOTS = KT * CH;      // chunk stream: (output tile, input tile); MFMA slots per output tile
  constexpr int EPS = (16 + SLOTS - 1) / SLOTS;                    // epilogue elements placed after each MFMA (1 when SLOTS >= 16)
#pragma unroll
  for (int tile = 0; tile < NT; tile++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = 32 * tile + mlp_rowmap(r) + 4 * h;
      out[tile][r] = (bias && row < n_valid) ? bias[row] : 0.0f;
    }
  auto epi = [&](int tile, int r) {
    if (EPI == kEpiTanh) out[tile][r] = tanh_acc(out[tile][r]);
    else if (EPI == kEpiDtanh) out[tile][r] *= fma_(-act[tile][r], act[tile][r], 1.0f);
  };
  float a[2][CH];
  auto load = [&](int c, int buf) {
#pragma unroll
    for (int j = 0; j < CH; j++)
      a[buf][j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Wk + (32 * (c % KT) + mlp_rowmap(j)) * LD + 32 * (c / KT)) + voff);   // + 4 h: in voff
  };
  load(0, 0);
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    if (c + 1 < NCH) load(c + 1, (c + 1) & 1);
    const int tile = c / KT, kt = c % KT;
#pragma unroll
    for (int j = 0; j < CH; j++) {
      out[tile] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][j], in[kt][j], out[tile], 0, 0, 0);
      if (EPI != kEpiNone && tile > 0) {
        const int slot = kt * CH + j;
        if (SLOTS >= 16) { if (slot % (SLOTS / 16) == 0) epi(tile - 1, slot / (SLOTS / 16)); }
        else {
#pragma unroll
          for (int e = 0; e < EPS; e++) if (slot * EPS + e < 16) epi(tile - 1, slot * EPS + e);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (EPI != kEpiNone) {
#pragma unroll
    for (int r = 0; r < 16; r++) epi(NT - 1, r);
  }
}

// accumulator tiles (lane = sample, registers = rows) -> LDS [row][kXs], this wavefront's 32 sample columns (lane = sample: conflict-free)
template <int NTILES>
__device__ __forceinline__ void xpose_store(float* buf, int row0, const f32x16* v, int col, int h) {
#pragma unroll
  for (int t = 0; t < NTILES; t++)
#pragma unroll
    for (int r = 0; r < 16; r++) buf[(row0 + 32 * t + mlp_rowmap(r) + 4 * h) * kXs + col] = v[t][r];
}

// d[i] += sum over the workgroup's 128 samples of dZ[o-tile][s] H[i-tile][s] for NI input tiles (row = 32-row tile index in the LDS
// transposes, lane n = neuron within the tile); bsum += this lane half's share of sum_s dZ[row n][s].
template <int NI>
__device__ __forceinline__ void mlp_dw(f32x16* d, float& bsum, const float* bufZ, int zrow, const float* bufH, int hrow, int n, int h) {
#ifndef MLP_DW_CHUNK
#define MLP_DW_CHUNK 8
#endif
#ifdef MLP_DW_OLD
#pragma unroll 4
  for (int t = 0; t < 64; t++) {
    const float a = bufZ[(zrow + n) * kXs + 2 * t + h];
    bsum += a;
#pragma unroll
    for (int i = 0; i < NI; i++) d[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bufH[(hrow + 32 * i + n) * kXs + 2 * t + h], d[i], 0, 0, 0);
  }
  return;
#endif
  constexpr int CH = MLP_DW_CHUNK / NI, NCH = 64 / CH;                 // LDS reads one chunk ahead of the MFMAs that use them
  const float* pz = bufZ + (zrow + n) * kXs + h;
  const float* ph = bufH + (hrow + n) * kXs + h;
  float a[2][CH], b[2][CH][NI];
  auto load = [&](int buf) {
#pragma unroll
    for (int j = 0; j < CH; j++) {
      a[buf][j] = pz[2 * j];
#pragma unroll
      for (int i = 0; i < NI; i++) b[buf][j][i] = ph[32 * i * kXs + 2 * j];
    }
    pz += 2 * CH; ph += 2 * CH;
  };
  auto mac = [&](int buf) {
#pragma unroll
    for (int j = 0; j < CH; j++) {
      bsum += a[buf][j];
#pragma unroll
      for (int i = 0; i < NI; i++) d[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][j], b[buf][j][i], d[i], 0, 0, 0);
    }
  };
  load(0);
#pragma unroll 1
  for (int c = 0; c < NCH; c += 2) {           // a rolled loop (two chunks per trip): unrolled, the register allocator spills
    load(1);
    mac(0);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 2 == NCH) { pz -= 2 * CH; ph -= 2 * CH; }   // last trip: re-read the last chunk instead of running past the row
    load(0);
    mac(1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// write one owned 32 x 32 tile (rows = output neurons in registers, column = input neuron on the lane) and its bias column to the slab
__device__ __forceinline__ void mlp_tile_out(float* slab, int stride, int row0, int col0, const f32x16& d, int n, int h) {
#pragma unroll
  for (int r = 0; r < 16; r++) slab[(row0 + mlp_rowmap(r) + 4 * h) * stride + col0 + n] = d[r];
}

struct MlpLoss { float clip, vf_coef, inv_n, mu, inv_sd; };

#ifdef AMENV_MLP_STAMPS   // diagnostic build: clocks per phase (forward, loss, weight gradients, data gradients), summed into stats slots 11..14
#define MLP_T0() unsigned long long t_ = __builtin_readcyclecounter()
#define MLP_TK(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_readcyclecounter(); tph[k] += float(n_ - t_); t_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MLP_T0()
#define MLP_TK(k)
#endif

// One PPO minibatch: forward + loss + backward + weight gradients of both nets.  grid = (blocks, 2), 256 threads.
template <int D, int A>
__global__ __launch_bounds__(256) void ppo_mlp_fused_kernel(const float* __restrict__ Pm, const float* __restrict__ WT, const float* __restrict__ obs,
                                                            const float* __restrict__ actions, const float* __restrict__ old_logp, const float* __restrict__ adv,
                                                            const float* __restrict__ ret, const int64_t* __restrict__ index, int64_t n, float clip,
                                                            float vf_coef, int normalize, const double* __restrict__ adv_part, int adv_blocks,
                                                            float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // transposes [kXposeRows][kXs] | stats [4 wavefronts][16]
  float* xb = lds;
  float* lstats = lds + kXposeRows * kXs;
  const int wave = int(threadIdx.x) >> 6, lane = int(threadIdx.x) & 63, nn = lane & 31, h = lane >> 5;
  const int col = wave * 32 + nn;                          // this lane's sample column in the transposes
  const int net = blockIdx.y;
  if (threadIdx.x < 64) lstats[threadIdx.x] = 0.0f;
  // advantage statistics: every block sums the partials in the same fixed order (as ppo_loss_grad)
  MlpLoss Lp{clip, vf_coef, 1.0f / float(n), 0.0f, 1.0f};
  if (normalize && n > 1) {
    double S = 0.0, Q = 0.0;
    for (int k = 0; k < adv_blocks; k++) { S += adv_part[2 * k]; Q += adv_part[2 * k + 1]; }
    const double m = S / double(n);
    const double var = fmax((Q - S * m) / double(n - 1), 0.0);
    Lp.mu = float(m);
    Lp.inv_sd = 1.0f / (float(sqrt(var)) + 1e-8f);
  }
  const MlpNet N = mlp_net(Pm, D, A, net);
  const float* wt = WT + net * kMlpWtPerNet;
  const float *W1T = wt, *W2T = wt + 32 * kH1, *W3T = W2T + kH1 * kH2, *W4T = W3T + kH2 * kH3, *W4P = W4T + kH3 * 32;
  // per-lane byte offsets of the A-operand loads, one per leading dimension: (4 h ld + n) * 4
  uint32_t vo128 = uint32_t(4 * h * 128 + nn) * 4u, vo64 = uint32_t(4 * h * 64 + nn) * 4u, vo32 = uint32_t(4 * h * 32 + nn) * 4u;
  float els[4], isd[4];                                   // log_std / 1 / std of this lane's four head rows (actor)
#pragma unroll
  for (int r = 0; r < 4; r++) { const int k = r + 4 * h; els[r] = k < A ? Pm[k] : 0.0f; isd[r] = __expf(-els[r]); }
  float s_dls[4] = {0, 0, 0, 0}, s_pol = 0.0f, s_val = 0.0f, s_clipn = 0.0f;
#ifdef AMENV_MLP_STAMPS
  float tph[4] = {0, 0, 0, 0};
#endif
  // weight-gradient tiles this wavefront owns, accumulated over every sample the workgroup sees:
  //   dW1 [128 x 32]: o-tile = wave            dW2 [64 x 128]: o-tile = wave >> 1, i-tiles 2 (wave & 1) + {0, 1}
  //   dW3 [64 x 64] : (wave >> 1, wave & 1)    dW4 [32 x 64] : i-tile = wave (wavefronts 0, 1)
  f32x16 gW1[1], gW2[2], gW3[1], gW4[1];
  float gb1 = 0.0f, gb2 = 0.0f, gb3 = 0.0f, gb4 = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; r++) { gW1[0][r] = 0.0f; gW2[0][r] = 0.0f; gW2[1][r] = 0.0f; gW3[0][r] = 0.0f; gW4[0][r] = 0.0f; }
  const int64_t ntiles = (n + 31) / 32;
  const int64_t iters = (ntiles + int64_t(gridDim.x) * 4 - 1) / (int64_t(gridDim.x) * 4);   // the same for every wavefront: barriers inside
  for (int64_t it = 0; it < iters; it++) {
    const int64_t tile = (int64_t(blockIdx.x) * iters + it) * 4 + wave;
    asm volatile("" : "+v"(vo128), "+v"(vo64), "+v"(vo32));   // keep the A-operand addresses inside the loop (see mlp_layer)
    const int64_t s = tile * 32 + nn;
    const bool sv = s < n;
    const int64_t sl = sv ? s : n - 1;
#ifdef MLP_NO_INDEX
    const int64_t sc = sl;
#else
    const int64_t sc = index ? index[sl] : sl;            // row of this sample in the rollout tensors
#endif
    // ---- forward
    MLP_T0();
    f32x16 X[1], H1[4], H2[2], H3[2], Y[1];
#pragma unroll
    for (int r = 0; r < 16; r++) { const int k = mlp_rowmap(r) + 4 * h; X[0][r] = (sv && k < D) ? obs[sc * D + k] : 0.0f; }
    mlp_layer<1, 4, 16, kH1, kEpiTanh>(W1T, N.b1, kH1, X, H1, vo128, h);
    mlp_layer<4, 2, 16, kH2, kEpiTanh>(W2T, N.b2, kH2, H1, H2, vo64, h);
    mlp_layer<2, 2, 16, kH3, kEpiTanh>(W3T, N.b3, kH3, H2, H3, vo64, h);
    mlp_layer<2, 1, 16, 32, kEpiNone>(W4T, N.b4, N.n_out, H3, Y, vo32, h);
    MLP_TK(0);
    // ---- loss gradient with respect to the head outputs (rows 0..3 in lane half 0, 4..7 in half 1)
    f32x16 dY[1];
#pragma unroll
    for (int r = 0; r < 16; r++) dY[0][r] = 0.0f;
    if (net == 0) {
      float z[4], lp = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int k = r + 4 * h;
        z[r] = k < A ? (actions[sc * A + k] - Y[0][r]) * isd[r] : 0.0f;
        lp += k < A ? fma_(-0.5f * z[r], z[r], -els[r]) - 0.918938533204672742f : 0.0f;
      }
      lp += __shfl_xor(lp, 32);
      const float ratio = __expf(lp - old_logp[sc]);
      const float a = (adv[sc] - Lp.mu) * Lp.inv_sd;
      const float s1 = a * ratio, s2 = a * fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
      const bool inside = ratio >= 1.0f - clip && ratio <= 1.0f + clip;
      const float g_lp = (sv && (inside || s1 < s2)) ? -a * ratio * Lp.inv_n : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        dY[0][r] = g_lp * z[r] * isd[r];
        s_dls[r] += (r + 4 * h < A) ? g_lp * fma_(z[r], z[r], -1.0f) : 0.0f;
      }
      if (sv && h == 0) { s_pol += -fminf(s1, s2); s_clipn += fabsf(ratio - 1.0f) > clip ? 1.0f : 0.0f; }
    } else if (h == 0) {
      const float dv = ret[sc] - Y[0][0];
      dY[0][0] = sv ? -2.0f * vf_coef * dv * Lp.inv_n : 0.0f;
      if (sv) s_val += dv * dv;
    }
    MLP_TK(1);
    // ---- backward.  Per layer: every wavefront publishes its samples' dZ and layer inputs, barrier, the owned dW tiles accumulate over the
    // workgroup's 128 samples while the data gradient of the next layer (registers only) is formed, barrier before the area is rewritten.
    float* xz = xb;                 // dZ rows
    // head: dZ4 = dY (32 rows, 8 used), inputs H3 (64 rows)
    __syncthreads();                // (previous iteration's readers are done)
    xpose_store<1>(xz, 0, dY, col, h);
    xpose_store<2>(xz, 32, H3, col, h);
    __syncthreads();
    if (wave < 2) mlp_dw<1>(gW4, gb4, xz, 0, xz, 32 + 32 * wave, nn, h);
    MLP_TK(2);
    f32x16 dZ3[2];
    mlp_layer<1, 2, 4, kH3, kEpiDtanh>(W4P, nullptr, 0, dY, dZ3, vo64, h, H3);     // dZ3 = (W4^T dY) (1 - H3^2)  (rows 0..7 of dY only)
    MLP_TK(3);
    // layer 3: dZ3 (64 rows), inputs H2 (64 rows)
    __syncthreads();
    xpose_store<2>(xz, 0, dZ3, col, h);
    xpose_store<2>(xz, 64, H2, col, h);
    __syncthreads();
    mlp_dw<1>(gW3, gb3, xz, 32 * (wave >> 1), xz, 64 + 32 * (wave & 1), nn, h);
    MLP_TK(2);
    f32x16 dZ2[2];
    mlp_layer<2, 2, 16, kH2, kEpiDtanh>(N.W3, nullptr, 0, dZ3, dZ2, vo64, h, H2);  // dZ2 = (W3^T dZ3) (1 - H2^2)
    MLP_TK(3);
    // layer 2: dZ2 (64 rows), inputs H1 (128 rows)
    __syncthreads();
    xpose_store<2>(xz, 0, dZ2, col, h);
    xpose_store<4>(xz, 64, H1, col, h);
    __syncthreads();
    mlp_dw<2>(gW2, gb2, xz, 32 * (wave >> 1), xz, 64 + 64 * (wave & 1), nn, h);
    MLP_TK(2);
    f32x16 dZ1[4];
    mlp_layer<2, 4, 16, kH1, kEpiDtanh>(N.W2, nullptr, 0, dZ2, dZ1, vo128, h, H1); // dZ1 = (W2^T dZ2) (1 - H1^2)
    MLP_TK(3);
    // layer 1: dZ1 (128 rows), inputs = the observation tile (32 rows)
    __syncthreads();
    xpose_store<4>(xz, 0, dZ1, col, h);
    xpose_store<1>(xz, 128, X, col, h);
    __syncthreads();
    mlp_dw<1>(gW1, gb1, xz, 32 * wave, xz, 128, nn, h);
    MLP_TK(2);
  }
  // ---- this workgroup's slab: owned tiles, bias columns (the two lane halves each summed half of the samples), loss sums
  float* slab = part + (size_t(blockIdx.y) * gridDim.x + blockIdx.x) * kAccSize;
  mlp_tile_out(slab + kAccW1, 33, 32 * wave, 0, gW1[0], nn, h);
  mlp_tile_out(slab + kAccW2, 129, 32 * (wave >> 1), 64 * (wave & 1), gW2[0], nn, h);
  mlp_tile_out(slab + kAccW2, 129, 32 * (wave >> 1), 64 * (wave & 1) + 32, gW2[1], nn, h);
  mlp_tile_out(slab + kAccW3, 65, 32 * (wave >> 1), 32 * (wave & 1), gW3[0], nn, h);
  if (wave < 2) mlp_tile_out(slab + kAccW4, 65, 0, 32 * wave, gW4[0], nn, h);
  gb1 += __shfl_xor(gb1, 32); gb2 += __shfl_xor(gb2, 32); gb3 += __shfl_xor(gb3, 32); gb4 += __shfl_xor(gb4, 32);
  if (h == 0) {
    slab[kAccW1 + (32 * wave + nn) * 33 + 32] = gb1;
    if ((wave & 1) == 0) { slab[kAccW2 + (32 * (wave >> 1) + nn) * 129 + 128] = gb2; slab[kAccW3 + (32 * (wave >> 1) + nn) * 65 + 64] = gb3; }
    if (wave == 0) slab[kAccW4 + nn * 65 + 64] = gb4;
  }
  // per-lane loss sums -> the stats slots (d log_std[0..A), policy, value, clip count): a butterfly over the 32 lanes of each half, one
  // LDS slot per wavefront, then a fixed-order sum over the wavefronts -- deterministic like the rest
  {
    float vals[7] = {s_dls[0], s_dls[1], s_dls[2], s_dls[3], s_pol, s_val, s_clipn};
#pragma unroll
    for (int k = 0; k < 7; k++)
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) vals[k] += __shfl_xor(vals[k], o);
    if (nn == 0) {
      float* ws_ = lstats + 16 * wave;
      if (net == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) ws_[r + 4 * h] = vals[r];
        if (h == 0) { ws_[8] = vals[4]; ws_[10] = vals[6]; }
      } else if (h == 0) ws_[9] = vals[5];
#ifdef AMENV_MLP_STAMPS
      if (h == 0) for (int k = 0; k < 4; k++) ws_[11 + k] = tph[k];
#endif
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) slab[kAccStats + threadIdx.x] = ((lstats[threadIdx.x] + lstats[16 + threadIdx.x]) + lstats[32 + threadIdx.x]) + lstats[48 + threadIdx.x];
}

// Fixed-order sum of the per-workgroup slabs into the flat gradient (SB3 parameter order) + d log_std + the four reported scalars.
// 512 threads = 64 outputs x 8 slab groups: a wavefront reads 64 consecutive slots of one slab (coalesced), the (up to 16) loads of its
// group in flight together; the group sums meet in LDS and are added in group order.
constexpr int kRedGroups = 8;
__global__ __launch_bounds__(64 * kRedGroups) void mlp_grad_reduce_kernel(const float* __restrict__ part, int blocks, int D, int A, int64_t n, const float* __restrict__ Pm,
                                                              float ent_coef, float* __restrict__ grad, float* __restrict__ stats) {
  __shared__ float sh[kRedGroups][64];
  const int trunk = kH1 * D + kH1 + kH2 * kH1 + kH2 + kH3 * kH2 + kH3;
  const int total = A + 2 * trunk + A * kH3 + A + kH3 + 1;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int tid = blockIdx.x * 64 + lane;
  int net = 0, slot = -1;                                    // slot < 0: nothing to sum (entropy, out of range)
  if (tid < A) slot = kAccStats + tid;
  else if (tid < total) {
    int e = tid - A;
    if (e < 2 * trunk) {
      net = e / trunk; e %= trunk;
      if (e < kH1 * D) slot = kAccW1 + (e / D) * 33 + e % D;
      else if ((e -= kH1 * D) < kH1) slot = kAccW1 + e * 33 + 32;
      else if ((e -= kH1) < kH2 * kH1) slot = kAccW2 + (e / kH1) * 129 + e % kH1;
      else if ((e -= kH2 * kH1) < kH2) slot = kAccW2 + e * 129 + 128;
      else if ((e -= kH2) < kH3 * kH2) slot = kAccW3 + (e / kH2) * 65 + e % kH2;
      else { e -= kH3 * kH2; slot = kAccW3 + e * 65 + 64; }
    } else {
      e -= 2 * trunk;
      if (e < A * kH3) { net = 0; slot = kAccW4 + (e / kH3) * 65 + e % kH3; }
      else if ((e -= A * kH3) < A) { net = 0; slot = kAccW4 + e * 65 + 64; }
      else if ((e -= A) < kH3) { net = 1; slot = kAccW4 + e; }
      else { net = 1; slot = kAccW4 + 64; }
    }
  } else if (tid < total + 4) {
    const int k = tid - total;                               // policy loss, value loss, entropy loss, clip fraction
    if (k == 0) slot = kAccStats + 8;
    else if (k == 1) { net = 1; slot = kAccStats + 9; }
    else if (k == 3) slot = kAccStats + 10;
  }
  float t = 0.0f;
  if (slot >= 0) {
    const int per = (blocks + kRedGroups - 1) / kRedGroups, b0 = grp * per, b1 = min(blocks, b0 + per);
    const float* p = part + size_t(net) * blocks * kAccSize + slot;
    int b = b0;
    for (; b + 16 <= b1; b += 16) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = p[size_t(b + j) * kAccSize];
#pragma unroll
      for (int j = 0; j < 16; j++) t += v[j];
    }
    for (; b < b1; b++) t += p[size_t(b) * kAccSize];
  }
  sh[grp][lane] = t;
  __syncthreads();
  if (grp != 0 || tid >= total + 4) return;
  t = 0.0f;
#pragma unroll
  for (int g = 0; g < kRedGroups; g++) t += sh[g][lane];
  if (tid >= total) {
    const int k = tid - total;
    if (k == 2) { float e = 0.0f; for (int j = 0; j < A; j++) e += 1.418938533204672742f + Pm[j]; stats[2] = -e; }
    else stats[k] = t / float(n);
  } else grad[tid] = tid < A ? t - ent_coef : t;
}

// Gradient-norm clip + Adam on the flat parameter buffer in one launch (torch.nn.utils.clip_grad_norm_ + torch.optim.Adam, the step SB3's
// PPO.train() takes).  Every workgroup sums |scale * g|^2 over the WHOLE buffer in the same order (so all agree on the norm to the last
// bit), then updates its slice.  The gradient buffer is READ-ONLY here: a first version wrote the clipped gradient back into it
// (clip_grad_norm_'s in-place semantics), so a workgroup still summing could read slices another one had already clipped -- different
// clip factors inside one step, run-to-run differences in the last bits (rare, timing-dependent; found by a bit-exact test failing once).
// Nothing downstream reads the clipped gradient, so it is no longer stored; a one-workgroup variant with a barrier between the phases was
// correct too but 20 us instead of 3.  `step` is torch's capturable-Adam step counter (a float on the device): read by every workgroup at
// entry, incremented by the last one to finish (ticket), after every workgroup has read it.
// hyper: [0] lr  [1] beta1  [2] beta2  [3] eps  [4] max_grad_norm (<= 0: no clipping)  [5] grad_scale (1 / world size after a sum all-reduce)
constexpr int kAdamBlock = 1024, kAdamMaxBlocks = 64;
__global__ __launch_bounds__(kAdamBlock) void adam_clip_kernel(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                                                               float* __restrict__ step, int64_t n, const float* __restrict__ hyper, float* __restrict__ grad_norm,
                                                               unsigned int* __restrict__ ticket) {
  __shared__ float sh[kAdamBlock / 64];
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], max_norm = hyper[4], scale = hyper[5];
  const float t = step[0] + 1.0f;
  float ss = 0.0f;
  {   // 16-byte loads, eight in flight per thread (the buffer is 16-byte aligned: checked by the caller); the sum order is fixed by (thread, i)
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(grad);
    int64_t i = threadIdx.x;
    for (; i + 7 * kAdamBlock < n4; i += 8 * kAdamBlock) {
      float4 g[8];
#pragma unroll
      for (int j = 0; j < 8; j++) g[j] = g4[i + j * kAdamBlock];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float a = g[j].x * scale, b = g[j].y * scale, c = g[j].z * scale, d = g[j].w * scale;
        ss = fma_(a, a, ss); ss = fma_(b, b, ss); ss = fma_(c, c, ss); ss = fma_(d, d, ss);
      }
    }
    for (; i < n4; i += kAdamBlock) {
      const float4 q = g4[i];
      const float a = q.x * scale, b = q.y * scale, c = q.z * scale, d = q.w * scale;
      ss = fma_(a, a, ss); ss = fma_(b, b, ss); ss = fma_(c, c, ss); ss = fma_(d, d, ss);
    }
    for (int64_t k = (n4 << 2) + threadIdx.x; k < n; k += kAdamBlock) { const float g = grad[k] * scale; ss = fma_(g, g, ss); }
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int k = 0; k < kAdamBlock / 64; k++) tot += sh[k];
  const float gn = sqrtf(tot);
  const float coef = max_norm > 0.0f ? fminf(max_norm / (gn + 1e-6f), 1.0f) * scale : scale;
  const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  const int64_t per = (n + gridDim.x - 1) / gridDim.x, i0 = int64_t(blockIdx.x) * per, i1 = i0 + per < n ? i0 + per : n;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += kAdamBlock) {
    const float g = grad[i] * coef;
    const float mi = fma_(g - m[i], 1.0f - b1, m[i]);                 // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = fma_(g * g, 1.0f - b2, v[i] * b2);               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    m[i] = mi; v[i] = vi;
    param[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && grad_norm) grad_norm[0] = gn;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) { step[0] = t; *ticket = 0u; }
  }
}

}  // namespace amenv_dev
