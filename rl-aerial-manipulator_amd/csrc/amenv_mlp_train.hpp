// amenv_mlp_train.hpp -- one PPO minibatch step of the reference's policy (v2/rl_train.py:38-56: SB3 MlpPolicy, net_arch [128, 64, 64],
// tanh, separate actor / critic trunks) as ONE kernel: forward of both MLPs, SB3's loss and its gradient, backward through both MLPs and
// all weight / bias gradients.  PyTorch ran this as ~90 library launches per minibatch (~0.9 ms for 65,536 samples, ~15 TFLOP/s).
//
// Arithmetic: fp32 in, fp32 out, on the bf16 matrix pipe.  Every fp32 operand is split EXACTLY into three bf16 parts (x = hi + mid + lo,
// 8 significant bits each, each rounded to nearest) and a product a b is formed from the six partial products of weight >= 2^-16
// (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid) by v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the three dropped ones are below
// 2^-23 |a b| -- the size of one fp32 rounding.  Round 2 ran the same data flow on v_mfma_f32_32x32x2_f32: on gfx950 that instruction
// issues at the vector FMA rate (157 TFLOP/s) and shares its issue slots with the wavefront's other VALU work (tools/micro/mfma_rate.hip);
// six bf16 MFMAs cover K = 16 in 192 clocks where eight fp32 MFMAs took 512, and VALU instructions cost ~2 clocks each in their shadow
// (tools/micro/mfma_bf16_rate.hip).  Against torch's fp32 modules the gradient differs by summation order only (test: <= 2e-5 of the
// largest gradient entry, unchanged from the fp32-MFMA version).
//
// Data flow of one wavefront = one tile of 32 samples (blockIdx.y = net: 0 actor, 1 critic):
//   * orientation D = W . X^T: M = 32 output neurons, N = the tile's 32 samples, K = inputs.  The accumulator tile of layer l (rows =
//     neurons in the 16 registers, column = sample on the lane) IS the B operand of layer l + 1: the MFMA pairs element j of lane half g
//     of A with element j of lane half g of B whatever k they stand for, so k-group q of input tile t is DEFINED as the rows registers
//     8 q .. 8 q + 7 hold: k = 32 t + 16 q + (j & 3) + 8 (j >> 2) + 4 g.  Activations never leave registers on the forward pass nor on
//     the data-gradient pass (dH = W^T dZ has the same shape); tanh, tanh' and the three-way split are lane-local.
//   * weights are the A operands: mlp_pack_element (prologue launch, after every optimiser step) writes, per net, the split weights of
//     the four forward and three data-gradient passes as a stream of 1-KB fragments in exactly the order the kernel consumes them
//     (chunk = (output tile, input tile) x k-group x part; lane x 8 bf16 = one 16-byte load per lane), L2-resident (186 KB per net),
//     fetched two chunks ahead of the MFMAs that use them through a three-chunk register ring.
//   * weight gradients contract over SAMPLES: the four wavefronts of a workgroup pool their tiles -- dZ and the layer inputs go through
//     an fp32 LDS transpose ([neuron][128 samples + 4]: written lane = sample, read 8 consecutive samples per lane as two ds_read_b128)
//     -- and every 32 x 32 tile of every dW is OWNED by one wavefront, which splits what it reads and accumulates the tile in registers
//     over all samples the workgroup sees.  No atomics.  Bias gradients ride along as the running sum of the A operands.  At the end a
//     workgroup writes its tiles to a partial slab; mlp_grad_reduce_kernel sums the slabs in a fixed order into the flat gradient: the
//     step is deterministic.
//   * the forward passes run as one chunk stream (each requests the next one's first weight chunks and leaves its last tile's tanh +
//     split to the shadow of the next one's first MFMAs); the data-gradient passes request theirs before the publish barriers; the
//     biases sit in LDS.  One wavefront per SIMD at ~480 registers: every latency that is not covered this way is paid in full.
//   * the loss part is SB3's (amenv_train.hpp ppo_loss_grad, same expressions), evaluated on the accumulator tile of the head: the 7
//     action means of a sample sit in registers 0..3 of the two lane halves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "amenv_train.hpp"

namespace amenv_dev {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kH1 = 128, kH2 = 64, kH3 = 64;
// per-workgroup gradient accumulator (floats): dW1 [128][33] (col 32 = bias) | dW2 [64][129] | dW3 [64][65] | dW4 [32][65] | stats [16]
constexpr int kAccW1 = 0, kAccW2 = kAccW1 + kH1 * 33, kAccW3 = kAccW2 + kH2 * 129, kAccW4 = kAccW3 + kH3 * 65, kAccStats = kAccW4 + 32 * 65;
constexpr int kAccSize = kAccStats + 16;
constexpr int kXs = 132;                                                      // transpose row stride (floats): 4 wavefronts x 32 samples + 4 -- 16-byte rows for
                                                                              // the ds_read_b128 of mlp_dw, 16 consecutive rows on 16 different bank quads
constexpr int kXposeRows = 192;                                               // largest layer: dZ2 (64 rows) + H1 (128 rows)
constexpr int kMlpLdsBytes = (kXposeRows * kXs + 64 + kH1 + kH2 + kH3 + 32) * 4;  // + stats [4 wavefronts][16] + the net's biases
constexpr int kMlpMaxBlocks = 256;

// The split-weight stream of one net: passes in the order the kernel runs them; a pass = NT output tiles x KT input tiles chunks, a chunk =
// NQ k-groups x 3 parts fragments of 64 lanes x 8 bf16 (1 KB).
struct MlpPass { int nt, kt, nq; };
__host__ __device__ constexpr MlpPass mlp_pass(int p) {
  return p == 0 ? MlpPass{4, 1, 2} : p == 1 ? MlpPass{2, 4, 2} : p == 2 ? MlpPass{2, 2, 2} : p == 3 ? MlpPass{1, 2, 2}   // forward: layer 1 (32 -> 128), 2 (128 -> 64), 3 (64 -> 64), head (64 -> 32)
       : p == 4 ? MlpPass{2, 1, 1} : p == 5 ? MlpPass{2, 2, 2} : MlpPass{4, 2, 2};                                       // data gradients: head (rows 0..15 of dY -> 64), layer 3 (64 -> 64), layer 2 (64 -> 128)
}
__host__ __device__ constexpr int mlp_pass_units(int p) { return mlp_pass(p).nt * mlp_pass(p).kt * mlp_pass(p).nq; }   // (chunk, k-group) units
template <int P> struct MlpUnitBase { static constexpr int value = MlpUnitBase<P - 1>::value + mlp_pass_units(P - 1); };   // first unit of pass P
template <> struct MlpUnitBase<0> { static constexpr int value = 0; };
constexpr int kMlpUnitsPerNet = MlpUnitBase<6>::value + mlp_pass_units(6);   // 62
constexpr int kMlpFragsPerNet = 3 * kMlpUnitsPerNet;                          // 186 fragments of 1 KB
constexpr int kMlpPackThreadsPerNet = kMlpUnitsPerNet * 512;                  // one thread per (unit, lane, j): writes the three parts

__host__ __device__ constexpr int mlp_rowmap(int r) { return (r & 3) + 8 * (r >> 2); }   // row of accumulator register r (+ 4 for the upper lane half)

__device__ __forceinline__ float tanh_acc(float x) {   // 1 - 2 / (exp(2x) + 1): exact limits, |err| ~1e-7 absolute (three VALU + two transcendental
  // instructions; a variant with a series around 0 for full RELATIVE accuracy cost 14 and bought nothing the gradient test can see)
  const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return fma_(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
}

// x = hi + mid + lo EXACTLY, each part a bf16 value (8 significant bits), by rounding to nearest: hi = bf16(x), mid = bf16(x - hi), lo = x - hi -
// mid (x has 24 bits: what is left after two 8-bit parts fits the third).  |mid| <= 2^-8 |hi|, |lo| <= 2^-8 |mid|, so the three products a
// six-product sum drops (mid lo, lo mid, lo lo) are below 2^-23 |a b|; truncation instead of rounding costs the same instructions and drops
// up to 2^-21 (tests/test_ppo_cpu.py restates both).  A pair at a time: v_cvt_pk_bf16_f32 rounds and packs two values in one instruction.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float v0, float v1) {   // bf16(v0) in the low half, bf16(v1) in the high half
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v0, v1}, bf16x2));
}
__device__ __forceinline__ void split_pair_packed(float v0, float v1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = cvt_pk_bf16(v0, v1);
  const float r0 = v0 - __uint_as_float(hi << 16), r1 = v1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(r0, r1);
  lo = cvt_pk_bf16(r0 - __uint_as_float(mid << 16), r1 - __uint_as_float(mid & 0xffff0000u));
}

// the three bf16 parts of the 8 values of one k-group, packed as MFMA operands: p[part][d] = elements 2 d (low half), 2 d + 1 (high half)
struct Bf3 { u32x4 p[3]; };
__device__ __forceinline__ void split_pair(float v0, float v1, Bf3& s, int d) {
  uint32_t h, m, l;
  split_pair_packed(v0, v1, h, m, l);
  s.p[0][d] = h; s.p[1][d] = m; s.p[2][d] = l;
}
template <int NQ = 2>
__device__ __forceinline__ void split_tile(const f32x16& v, Bf3* s) {   // s[q]: registers 8 q .. 8 q + 7
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int d = 0; d < 4; d++) split_pair(v[8 * q + 2 * d], v[8 * q + 2 * d + 1], s[q], d);
}

// acc += A . B from the six partial products of weight >= 2^-16, smallest first
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__host__ __device__ constexpr int prod_a(int i) { return (0x001102 >> (4 * i)) & 15; }   // A parts 2 0 1 1 0 0
__host__ __device__ constexpr int prod_b(int i) { return (0x010120 >> (4 * i)) & 15; }   // B parts 0 2 1 0 1 0

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

// One element of the split-weight stream: thread = (net, unit = (chunk, k-group) of a pass, lane, j).  Lane (m, g) of the A operand of
// output tile `tile`, input tile `kt`, k-group q holds, in element j, the weight that multiplies input k = 32 kt + 16 q + (j & 3) +
// 8 (j >> 2) + 4 g into output 32 tile + m: W[out][k] on the forward passes, W[k][out] (out = input neuron) on the data-gradient passes;
// zero where the row / column does not exist.  The three parts go to fragments 3 unit + {0, 1, 2}.
__device__ __forceinline__ void mlp_pack_element(const float* __restrict__ Pm, int D, int A, uint16_t* __restrict__ WS, int tid) {
  if (tid >= 2 * kMlpPackThreadsPerNet) return;
  const int net = tid / kMlpPackThreadsPerNet, e = tid % kMlpPackThreadsPerNet;
  const int unit = e >> 9, lane = (e >> 3) & 63, j = e & 7, m = lane & 31, g = lane >> 5;
  int pass = 0, ubase = 0;
#define MLP_PASS_OF(P) if (unit >= MlpUnitBase<P>::value) { pass = P; ubase = MlpUnitBase<P>::value; }
  MLP_PASS_OF(1) MLP_PASS_OF(2) MLP_PASS_OF(3) MLP_PASS_OF(4) MLP_PASS_OF(5) MLP_PASS_OF(6)
#undef MLP_PASS_OF
  const int u = unit - ubase, nq = (pass == 4) ? 1 : 2, kt_n = (pass == 1) ? 4 : (pass == 0 || pass == 4) ? 1 : 2;
  static_assert(mlp_pass(4).nq == 1 && mlp_pass(1).kt == 4 && mlp_pass(0).kt == 1 && mlp_pass(4).kt == 1 && mlp_pass(2).kt == 2 && mlp_pass(3).kt == 2 &&
                mlp_pass(5).kt == 2 && mlp_pass(6).kt == 2, "the pass table and its run-time copy disagree");
  const int c = u / nq, q = u % nq, tile = c / kt_n, kt = c % kt_n;
  const int out = 32 * tile + m, k = 32 * kt + 16 * q + (j & 3) + 8 * (j >> 2) + 4 * g;
  const MlpNet N = mlp_net(Pm, D, A, net);
  float v = 0.0f;
  switch (pass) {
    case 0: v = k < D ? N.W1[out * D + k] : 0.0f; break;
    case 1: v = N.W2[out * kH1 + k]; break;
    case 2: v = N.W3[out * kH2 + k]; break;
    case 3: v = out < N.n_out ? N.W4[out * kH3 + k] : 0.0f; break;
    case 4: v = k < N.n_out ? N.W4[k * kH3 + out] : 0.0f; break;
    case 5: v = N.W3[k * kH2 + out]; break;
    default: v = N.W2[k * kH1 + out]; break;
  }
  uint32_t part[3];
  split_pair_packed(v, 0.0f, part[0], part[1], part[2]);   // (the low halves: this element's parts)
  uint16_t* dst = WS + (size_t(net) * kMlpFragsPerNet + 3 * unit) * 512 + lane * 8 + j;
#pragma unroll
  for (int p = 0; p < 3; p++) dst[p * 512] = uint16_t(part[p]);
}
// Everything the fused kernel needs beforehand in ONE launch (each launch in this chain costs ~4.7 us whatever it does): workgroups
// [0, adv_blocks) sum the minibatch's advantages (ppo_adv_partials), the rest write the split-weight streams.  256 threads.
__global__ __launch_bounds__(kPpoBlock) void ppo_mlp_prologue_kernel(const float* __restrict__ adv, int64_t n, double* __restrict__ adv_part,
                                                                     const int64_t* __restrict__ index, int adv_blocks, const float* __restrict__ Pm, int D, int A,
                                                                     uint16_t* __restrict__ WS) {
  if (int(blockIdx.x) < adv_blocks) ppo_adv_partials_block(adv, n, adv_part, index, int(blockIdx.x), adv_blocks);   // uniform per workgroup
  else mlp_pack_element(Pm, D, A, WS, (int(blockIdx.x) - adv_blocks) * kPpoBlock + int(threadIdx.x));
}

// out[tile] = W in (+ bias) for pass PASS of the stream: NT output tiles, KT input tiles given as split k-groups in[kt][q].
// Order: OUTPUT-tile major, one chunk = the k-groups of one input tile (NQ x 6 MFMAs).  Every A operand is ONE 16-byte load "uniform
// base + per-lane offset" (`voff` = 16 lane: the only address register; passed through an empty asm by the caller once per sample tile,
// or hipcc hoists the loop-invariant 64-bit addresses of the whole kernel out of the tile loop and spills them), issued two chunks ahead.
// A finished tile's epilogue (tanh, or the tanh derivative of the backward pass, and the three-way split that makes it the next pass's
// B operand) is spread, register pair by register pair, between the MFMAs of the NEXT tile: the vector ALU works in the shadow of the
// matrix pipe instead of after it.
enum { kEpiNone = 0, kEpiTanh = 1, kEpiDtanh = 2 };
// epilogue of registers 2 e, 2 e + 1 of one finished tile: activation (or its derivative against the forward activation `act`), split
template <int EPI, bool SPLIT>
__device__ __forceinline__ void mlp_epi_pair(f32x16& out, const f32x16* act, Bf3* outS, int e) {
  float v0 = out[2 * e], v1 = out[2 * e + 1];
  if (EPI == kEpiTanh) { v0 = tanh_acc(v0); v1 = tanh_acc(v1); }
  else if (EPI == kEpiDtanh) { v0 *= fma_(-(*act)[2 * e], (*act)[2 * e], 1.0f); v1 *= fma_(-(*act)[2 * e + 1], (*act)[2 * e + 1], 1.0f); }
  out[2 * e] = v0; out[2 * e + 1] = v1;
  if (SPLIT) split_pair(v0, v1, outS[e >> 2], e & 3);
}
// accumulators of a forward pass = its bias, from the copy the workgroup keeps in LDS ([b1 128 | b2 64 | b3 64 | b4 padded to 32]: ~120
// clocks before the pass's first MFMA instead of an L2 round trip; all lanes of a half read the same 16 bytes)
constexpr int kBiasFloats = kH1 + kH2 + kH3 + 32;
template <int PASS>
__device__ __forceinline__ void mlp_bias_init(const float* lbias, f32x16* out, int h) {
  constexpr int NT = mlp_pass(PASS).nt, OFF = PASS == 0 ? 0 : PASS == 1 ? kH1 : PASS == 2 ? kH1 + kH2 : kH1 + kH2 + kH3;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int tile = 0; tile < NT; tile++)
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + OFF + 32 * tile + 8 * g4 + 4 * h);
#pragma unroll
      for (int k = 0; k < 4; k++) out[tile][4 * g4 + k] = b[k];
    }
}
struct MlpNoPrev { __device__ __forceinline__ void operator()(int) const {} };
// The A-operand ring: three chunks of up to six fragments.  The forward passes run as ONE chunk stream through it (SLOT0 = first slot of a
// pass, PRELOADED = its chunks 0 and 1 were fetched by the pass before, NEXT = the pass whose chunks 0 and 1 this one fetches during its
// last two chunks): a pass that starts cold waits an L2 round trip (~700 clocks) before its first MFMA.
struct MlpRing { u32x4 a[3][6]; };
template <int PASS>
__device__ __forceinline__ void mlp_ring_load(MlpRing& R, const u32x4* __restrict__ WSnet, uint32_t voff, int c, int slot) {
  constexpr int FR = 3 * mlp_pass(PASS).nq, FBASE = 3 * MlpUnitBase<PASS>::value;
#pragma unroll
  for (int f = 0; f < FR; f++) {   // address = uniform base + (per-lane offset + 4-KB group: one VALU add per group) + immediate < 4 KB
    const uint32_t at = uint32_t(FBASE + c * FR + f) * 1024u;
    R.a[slot][f] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(WSnet) + size_t(voff + (at & ~4095u)) + (at & 4095u));
  }
}
template <int PASS>
__device__ __forceinline__ void mlp_ring_prefetch(MlpRing& R, const u32x4* __restrict__ WSnet, uint32_t voff) {   // chunks 0, 1 -> slots 0, 1
  mlp_ring_load<PASS>(R, WSnet, voff, 0, 0);
  if (mlp_pass(PASS).nt * mlp_pass(PASS).kt > 1) mlp_ring_load<PASS>(R, WSnet, voff, 1, 1);
}
// `prev`: the epilogue pairs of the LAST tile of the pass before (which made this pass's last input tile), run between the MFMAs of this
// pass's first chunks -- those read the earlier input tiles only -- when PREV is set (needs KT >= 2); a pass whose successor does that
// is called with DEFER and leaves its last tile unfinished.
template <int PASS, int EPI, bool SPLIT, int SLOT0 = 0, bool PRELOADED = false, int NEXT = -1, bool DEFER = false, bool PREV = false, class Prev = MlpNoPrev>
__device__ __forceinline__ void mlp_layer(MlpRing& R, const u32x4* __restrict__ WSnet, const float* lbias, const Bf3 (*in)[2], f32x16* out,
                                          Bf3 (*outS)[2], uint32_t voff, int h, const f32x16* act = nullptr, Prev prev = Prev()) {
  constexpr int NT = mlp_pass(PASS).nt, KT = mlp_pass(PASS).kt, NQ = mlp_pass(PASS).nq;
  constexpr int NCH = NT * KT, SLOTS = KT * NQ * 6;                // chunks; MFMA slots per output tile
  constexpr int STRIDE = SLOTS / 8 > 0 ? SLOTS / 8 : 1, PER = (8 + SLOTS - 1) / SLOTS;   // epilogue pairs: one every STRIDE slots, or PER per slot
  constexpr int PSLOTS = (KT - 1) * NQ * 6, PSTRIDE = PSLOTS / 8 > 0 ? PSLOTS / 8 : 1;   // slots and spacing for the pass before's pairs
  static_assert(!PREV || PSLOTS >= 8, "the deferred epilogue needs eight MFMA slots before the last input tile is read");
  if (PASS < 4) mlp_bias_init<PASS>(lbias, out, h);
  else {
#pragma unroll
    for (int tile = 0; tile < NT; tile++)
#pragma unroll
      for (int r = 0; r < 16; r++) out[tile][r] = 0.0f;
  }
  auto epi = [&](int tile, int e) { mlp_epi_pair<EPI, SPLIT>(out[tile], act ? act + tile : nullptr, SPLIT ? outS[tile] : nullptr, e); };
  if (!PRELOADED) {
    mlp_ring_load<PASS>(R, WSnet, voff, 0, SLOT0 % 3);
    if (NCH > 1) mlp_ring_load<PASS>(R, WSnet, voff, 1, (SLOT0 + 1) % 3);
  }
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    if (c + 2 < NCH) mlp_ring_load<PASS>(R, WSnet, voff, c + 2, (SLOT0 + c + 2) % 3);
    else if (NEXT >= 0) mlp_ring_load<(NEXT >= 0 ? NEXT : 0)>(R, WSnet, voff, c + 2 - NCH, (SLOT0 + c + 2) % 3);
    const int tile = c / KT, kt = c % KT;
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
      for (int i = 0; i < 6; i++) {
        out[tile] = mfma_bf16(R.a[(SLOT0 + c) % 3][3 * q + prod_a(i)], in[kt][q].p[prod_b(i)], out[tile]);
        const int slot = (kt * NQ + q) * 6 + i;
        if (PREV && tile == 0 && kt < KT - 1) {
          if (slot % PSTRIDE == 0 && slot / PSTRIDE < 8) prev(slot / PSTRIDE);
          __builtin_amdgcn_sched_barrier(0);
        }
        if ((EPI != kEpiNone || SPLIT) && tile > 0) {
          if (SLOTS >= 8) { if (slot % STRIDE == 0 && slot / STRIDE < 8) epi(tile - 1, slot / STRIDE); }
          else {
#pragma unroll
            for (int e = 0; e < PER; e++) if (slot * PER + e < 8) epi(tile - 1, slot * PER + e);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  if ((EPI != kEpiNone || SPLIT) && !DEFER) {
#pragma unroll
    for (int e = 0; e < 8; e++) epi(NT - 1, e);
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
// transposes, lane (n, g) = neuron n of the tile, samples 16 kq + 8 g .. + 7 of every k-group kq: two ds_read_b128 per operand);
// bsum += this lane's share of sum_s dZ[row n][s].  The fp32 values are split three ways HERE, by the wavefront that consumes them
// (11 VALU instructions per pair of values, in the shadow of the previous k-group's MFMAs): 6 bf16 MFMAs per 16 samples instead of
// 8 fp32 ones at twice the clocks each.  (Publishing the split parts instead -- three bf16 planes [sample][row], written as 8-byte
// chunks and read back with ds_read_b64_tr_b16 -- was built and is correct, but its extra splitting and LDS stores on the publishing side
// cost more than the reads here save: 167 us per 65,536 samples against 149 with fp32 MFMAs in this function.)
template <int NI, int NKQ = 8>
__device__ __forceinline__ void mlp_dw(f32x16* d, float& bsum, const float* bufZ, int zrow, const float* bufH, int hrow, int n, int g, int kq0 = 0) {
  const float* pz = bufZ + (zrow + n) * kXs + 8 * g + 16 * kq0;   // k-groups kq0 .. kq0 + NKQ - 1
  const float* ph = bufH + (hrow + n) * kXs + 8 * g + 16 * kq0;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 a[2][2], b[2][NI][2];
  auto load = [&](int buf) {   // LDS reads one k-group ahead of the MFMAs that use them
#pragma unroll
    for (int k = 0; k < 2; k++) {
      a[buf][k] = *reinterpret_cast<const f32x4*>(pz + 4 * k);
#pragma unroll
      for (int i = 0; i < NI; i++) b[buf][i][k] = *reinterpret_cast<const f32x4*>(ph + 32 * i * kXs + 4 * k);
    }
    pz += 16; ph += 16;
  };
  auto split8 = [&](const f32x4* v, Bf3& s) {
#pragma unroll
    for (int dd = 0; dd < 4; dd++) split_pair(v[dd >> 1][2 * (dd & 1)], v[dd >> 1][2 * (dd & 1) + 1], s, dd);
  };
  auto mac = [&](int buf) {
#pragma unroll
    for (int k = 0; k < 2; k++) bsum += (a[buf][k][0] + a[buf][k][1]) + (a[buf][k][2] + a[buf][k][3]);
    Bf3 as, bs;
    split8(a[buf], as);
#pragma unroll
    for (int i = 0; i < NI; i++) {
      split8(b[buf][i], bs);
#pragma unroll
      for (int k = 0; k < 6; k++) d[i] = mfma_bf16(as.p[prod_a(k)], bs.p[prod_b(k)], d[i]);
    }
  };
  load(0);
#pragma unroll 1
  for (int kq = 0; kq < NKQ; kq += 2) {        // a rolled loop (two k-groups per trip): unrolled, the register allocator spills
    load(1);
    mac(0);
    __builtin_amdgcn_sched_barrier(0);
    if (kq + 2 == NKQ) { pz -= 16; ph -= 16; } // last trip: re-read the last k-group instead of running past the row
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

#ifdef AMENV_MLP_STAMPS   // diagnostic build: clocks per phase (forward, loss, weight gradients, data gradients, LDS publish incl. barriers), summed into stats slots 11..15
#define MLP_T0() unsigned long long t_ = __builtin_readcyclecounter()
#define MLP_TK(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_readcyclecounter(); tph[k] += float(n_ - t_); t_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MLP_T0()
#define MLP_TK(k)
#endif

// One PPO minibatch: forward + loss + backward + weight gradients of both nets.  grid = (blocks, 2), 256 threads.
template <int D, int A>
__global__ __launch_bounds__(256) void ppo_mlp_fused_kernel(const float* __restrict__ Pm, const u32x4* __restrict__ WS, const float* __restrict__ obs,
                                                            const float* __restrict__ actions, const float* __restrict__ old_logp, const float* __restrict__ adv,
                                                            const float* __restrict__ ret, const int64_t* __restrict__ index, int64_t n, float clip,
                                                            float vf_coef, int normalize, const double* __restrict__ adv_part, int adv_blocks,
                                                            float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float mlp_lds[];   // transposes [kXposeRows][kXs] | stats [4 wavefronts][16]
  float* xz = mlp_lds;
  float* lstats = mlp_lds + kXposeRows * kXs;
  const int wave = int(threadIdx.x) >> 6, lane = int(threadIdx.x) & 63, nn = lane & 31, h = lane >> 5;
  const int col = wave * 32 + nn;                          // this lane's sample column in the transposes
  const int net = blockIdx.y;
  float* lbias = lstats + 64;                             // [b1 | b2 | b3 | b4 padded to 32 rows]
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
  for (int k = threadIdx.x; k < kBiasFloats; k += 256) {
    const int k4 = k - (kH1 + kH2 + kH3);
    lbias[k] = k < kH1 ? N.b1[k] : k < kH1 + kH2 ? N.b2[k - kH1] : k4 < 0 ? N.b3[k - kH1 - kH2] : k4 < N.n_out ? N.b4[k4] : 0.0f;
  }
  __syncthreads();
  const u32x4* ws = WS + size_t(net) * kMlpFragsPerNet * 64;
  uint32_t voff = uint32_t(lane) * 16u;                    // per-lane byte offset of every A-operand load
  float els[4], isd[4];                                   // log_std / 1 / std of this lane's four head rows (actor)
#pragma unroll
  for (int r = 0; r < 4; r++) { const int k = r + 4 * h; els[r] = k < A ? Pm[k] : 0.0f; isd[r] = __expf(-els[r]); }
  float s_dls[4] = {0, 0, 0, 0}, s_pol = 0.0f, s_val = 0.0f, s_clipn = 0.0f;
#ifdef AMENV_MLP_STAMPS
  float tph[5] = {0, 0, 0, 0, 0};
#endif
  // weight-gradient tiles this wavefront owns, accumulated over every sample the workgroup sees:
  //   dW1 [128 x 32]: o-tile = wave            dW2 [64 x 128]: o-tile = wave >> 1, i-tiles 2 (wave & 1) + {0, 1}
  //   dW3 [64 x 64] : (wave >> 1, wave & 1)    dW4 [32 x 64] : i-tile = wave & 1, samples 64 (wave >> 1) .. + 63 of the workgroup's 128
  f32x16 gW1[1], gW2[2], gW3[1], gW4[1];
  float gb1 = 0.0f, gb2 = 0.0f, gb3 = 0.0f, gb4 = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; r++) { gW1[0][r] = 0.0f; gW2[0][r] = 0.0f; gW2[1][r] = 0.0f; gW3[0][r] = 0.0f; gW4[0][r] = 0.0f; }
  const int64_t ntiles = (n + 31) / 32;
  const int64_t iters = (ntiles + int64_t(gridDim.x) * 4 - 1) / (int64_t(gridDim.x) * 4);   // the same for every wavefront: barriers inside
  for (int64_t it = 0; it < iters; it++) {
    const int64_t tile = (int64_t(blockIdx.x) * iters + it) * 4 + wave;
    asm volatile("" : "+v"(voff));   // keep the A-operand addresses inside the loop (see mlp_layer)
    const int64_t s = tile * 32 + nn;
    const bool sv = s < n;
    const int64_t sl = sv ? s : n - 1;
    const int64_t sc = index ? index[sl] : sl;            // row of this sample in the rollout tensors
    // ---- forward
    MLP_T0();
    f32x16 X[1], H1[4], H2[2], H3[2], Y[1];
#pragma unroll
    for (int r = 0; r < 16; r++) { const int k = mlp_rowmap(r) + 4 * h; const float x = obs[sc * D + (k < D ? k : D - 1)]; X[0][r] = (sv && k < D) ? x : 0.0f; }
    // the loss inputs of this sample, fetched now (HBM latency: nothing in the forward pass depends on them)
    float act4[4], olp_s, adv_s;
#pragma unroll
    for (int r = 0; r < 4; r++) { const int k = r + 4 * h; act4[r] = net == 0 ? actions[sc * A + (k < A ? k : A - 1)] : 0.0f; }   // (rows >= A are masked where used)
    olp_s = net == 0 ? old_logp[sc] : 0.0f;
    adv_s = net == 0 ? adv[sc] : ret[sc];                  // critic: the return
    MlpRing ring;
    {
      Bf3 XS[1][2], H1S[4][2];
      split_tile(X[0], XS[0]);
      // one chunk stream through the ring: 4 + 8 + 4 + 2 chunks; each pass fetches the next one's first two chunks and leaves its last
      // tile's tanh + split to the first MFMAs of the next
      mlp_layer<0, kEpiTanh, true, 0, false, 1, true>(ring, ws, lbias, XS, H1, H1S, voff, h);
      Bf3 H2S[2][2];
      mlp_layer<1, kEpiTanh, true, 4 % 3, true, 2, true, true>(ring, ws, lbias, H1S, H2, H2S, voff, h, nullptr,
                                                                [&](int e) { mlp_epi_pair<kEpiTanh, true>(H1[3], nullptr, H1S[3], e); });
      Bf3 H3S[2][2];
      mlp_layer<2, kEpiTanh, true, 12 % 3, true, 3, true, true>(ring, ws, lbias, H2S, H3, H3S, voff, h, nullptr,
                                                                 [&](int e) { mlp_epi_pair<kEpiTanh, true>(H2[1], nullptr, H2S[1], e); });
      mlp_layer<3, kEpiNone, false, 16 % 3, true, -1, false, true>(ring, ws, lbias, H3S, Y, nullptr, voff, h, nullptr,
                                                                    [&](int e) { mlp_epi_pair<kEpiTanh, true>(H3[1], nullptr, H3S[1], e); });
    }
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
        z[r] = k < A ? (act4[r] - Y[0][r]) * isd[r] : 0.0f;
        lp += k < A ? fma_(-0.5f * z[r], z[r], -els[r]) - 0.918938533204672742f : 0.0f;
      }
      lp += __shfl_xor(lp, 32);
      const float ratio = __expf(lp - olp_s);
      const float a = (adv_s - Lp.mu) * Lp.inv_sd;
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
      const float dv = adv_s - Y[0][0];
      dY[0][0] = sv ? -2.0f * vf_coef * dv * Lp.inv_n : 0.0f;
      if (sv) s_val += dv * dv;
    }
    Bf3 dYS[1][2];
    split_tile<1>(dY[0], dYS[0]);                           // rows 0..15 are all the head pass reads
    MLP_TK(1);
    // ---- backward.  Per layer: every wavefront publishes its samples' dZ and layer inputs, barrier; the data gradient of the next layer
    // (registers only; its first weight chunks were requested before the barriers) is formed, then the owned dW tiles accumulate over the
    // workgroup's 128 samples; barrier before the area is rewritten.
    // head: dZ4 = dY (32 rows, 8 used), inputs H3 (64 rows)
    mlp_ring_prefetch<4>(ring, ws, voff);
    __syncthreads();                // (previous iteration's readers are done)
    xpose_store<1>(xz, 0, dY, col, h);
    xpose_store<2>(xz, 32, H3, col, h);
    __syncthreads();
    MLP_TK(4);
    f32x16 dZ3[2];
    Bf3 dZ3S[2][2];
    mlp_layer<4, kEpiDtanh, true, 0, true>(ring, ws, nullptr, dYS, dZ3, dZ3S, voff, h, H3);     // dZ3 = (W4^T dY) (1 - H3^2)  (rows 0..15 of dY only)
    MLP_TK(3);
    mlp_dw<1, 4>(gW4, gb4, xz, 0, xz, 32 + 32 * (wave & 1), nn, h, 4 * (wave >> 1));   // i-tile = wave & 1, half of the samples each: summed at the end
    MLP_TK(2);
    // layer 3: dZ3 (64 rows), inputs H2 (64 rows)
    mlp_ring_prefetch<5>(ring, ws, voff);
    __syncthreads();
    xpose_store<2>(xz, 0, dZ3, col, h);
    xpose_store<2>(xz, 64, H2, col, h);
    __syncthreads();
    MLP_TK(4);
    f32x16 dZ2[2];
    Bf3 dZ2S[2][2];
    mlp_layer<5, kEpiDtanh, true, 0, true>(ring, ws, nullptr, dZ3S, dZ2, dZ2S, voff, h, H2);    // dZ2 = (W3^T dZ3) (1 - H2^2)
    MLP_TK(3);
    mlp_dw<1>(gW3, gb3, xz, 32 * (wave >> 1), xz, 64 + 32 * (wave & 1), nn, h);
    MLP_TK(2);
    // layer 2: dZ2 (64 rows), inputs H1 (128 rows)
    mlp_ring_prefetch<6>(ring, ws, voff);
    __syncthreads();
    xpose_store<2>(xz, 0, dZ2, col, h);
    xpose_store<4>(xz, 64, H1, col, h);
    __syncthreads();
    MLP_TK(4);
    f32x16 dZ1[4];
    mlp_layer<6, kEpiDtanh, false, 0, true>(ring, ws, nullptr, dZ2S, dZ1, nullptr, voff, h, H1); // dZ1 = (W2^T dZ2) (1 - H1^2)
    MLP_TK(3);
    mlp_dw<2>(gW2, gb2, xz, 32 * (wave >> 1), xz, 64 + 64 * (wave & 1), nn, h);
    MLP_TK(2);
    // layer 1: dZ1 (128 rows), inputs = the observation tile (32 rows)
    __syncthreads();
    xpose_store<4>(xz, 0, dZ1, col, h);
    xpose_store<1>(xz, 128, X, col, h);
    __syncthreads();
    MLP_TK(4);
    mlp_dw<1>(gW1, gb1, xz, 32 * wave, xz, 128, nn, h);
    MLP_TK(2);
  }
  // ---- this workgroup's slab: owned tiles, bias columns (the two lane halves each summed half of the samples), loss sums
  float* slab = part + (size_t(blockIdx.y) * gridDim.x + blockIdx.x) * kAccSize;
  mlp_tile_out(slab + kAccW1, 33, 32 * wave, 0, gW1[0], nn, h);
  mlp_tile_out(slab + kAccW2, 129, 32 * (wave >> 1), 64 * (wave & 1), gW2[0], nn, h);
  mlp_tile_out(slab + kAccW2, 129, 32 * (wave >> 1), 64 * (wave & 1) + 32, gW2[1], nn, h);
  mlp_tile_out(slab + kAccW3, 65, 32 * (wave >> 1), 32 * (wave & 1), gW3[0], nn, h);
  {   // dW4: wavefronts 2, 3 hand their half-sample sums to 0, 1 through LDS (the transposes are free: every reader passed the last barrier)
    __syncthreads();
    float* hand = xz + (wave & 1) * 17 * 64;
    if (wave >= 2) {
#pragma unroll
      for (int r = 0; r < 16; r++) hand[r * 64 + lane] = gW4[0][r];
      hand[16 * 64 + lane] = gb4;
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
      for (int r = 0; r < 16; r++) gW4[0][r] += hand[r * 64 + lane];
      gb4 += hand[16 * 64 + lane];
      mlp_tile_out(slab + kAccW4, 65, 0, 32 * wave, gW4[0], nn, h);
    }
  }
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
      if (h == 0) for (int k = 0; k < 5; k++) ws_[11 + k] = tph[k];
#endif
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) slab[kAccStats + threadIdx.x] = ((lstats[threadIdx.x] + lstats[16 + threadIdx.x]) + lstats[32 + threadIdx.x]) + lstats[48 + threadIdx.x];
}

// The forward passes alone: mean [n, A] = action_net(pi trunk(obs)), value [n] = value_net(vf trunk(obs)) for large batches (the rollout
// buffer's log-probabilities / values, a vec-env's policy step), same arithmetic and weight stream as the training kernel.  grid =
// (blocks, 2 nets), 256 threads; a wavefront takes 32 rows per iteration.  amenv_policy.hpp's VALU kernel stays for small batches.
__global__ void mlp_pack_kernel(const float* __restrict__ Pm, int D, int A, uint16_t* __restrict__ WS) {
  mlp_pack_element(Pm, D, A, WS, int(blockIdx.x * blockDim.x + threadIdx.x));
}
template <int D, int A>
__global__ __launch_bounds__(256, 2) void mlp_forward_kernel(const float* __restrict__ Pm, const u32x4* __restrict__ WS, const float* __restrict__ obs, int64_t n,
                                                          float* __restrict__ mean, float* __restrict__ value) {
  __shared__ __attribute__((aligned(16))) float lbias[kBiasFloats];
  const int wave = int(threadIdx.x) >> 6, lane = int(threadIdx.x) & 63, nn = lane & 31, h = lane >> 5;
  const int net = blockIdx.y;
  if ((net == 0 && !mean) || (net == 1 && !value)) return;   // whole workgroup: uniform
  const MlpNet N = mlp_net(Pm, D, A, net);
  for (int k = threadIdx.x; k < kBiasFloats; k += 256) {
    const int k4 = k - (kH1 + kH2 + kH3);
    lbias[k] = k < kH1 ? N.b1[k] : k < kH1 + kH2 ? N.b2[k - kH1] : k4 < 0 ? N.b3[k - kH1 - kH2] : k4 < N.n_out ? N.b4[k4] : 0.0f;
  }
  __syncthreads();
  const u32x4* ws = WS + size_t(net) * kMlpFragsPerNet * 64;
  uint32_t voff = uint32_t(lane) * 16u;
  const int64_t ntiles = (n + 31) / 32;
  for (int64_t tile = int64_t(blockIdx.x) * 4 + wave; tile < ntiles; tile += int64_t(gridDim.x) * 4) {   // (no barrier inside: a ragged tail is fine)
    asm volatile("" : "+v"(voff));
    const int64_t s = tile * 32 + nn;
    const bool sv = s < n;
    const int64_t sc = sv ? s : n - 1;
    f32x16 X, H1[4], H2[2], H3[2], Y[1];
#pragma unroll
    for (int r = 0; r < 16; r++) { const int k = mlp_rowmap(r) + 4 * h; const float x = obs[sc * D + (k < D ? k : D - 1)]; X[r] = k < D ? x : 0.0f; }
    MlpRing ring;
    Bf3 XS[1][2], H1S[4][2];
    split_tile(X, XS[0]);
    mlp_layer<0, kEpiTanh, true, 0, false, 1, true>(ring, ws, lbias, XS, H1, H1S, voff, h);
    Bf3 H2S[2][2];
    mlp_layer<1, kEpiTanh, true, 4 % 3, true, 2, true, true>(ring, ws, lbias, H1S, H2, H2S, voff, h, nullptr,
                                                              [&](int e) { mlp_epi_pair<kEpiTanh, true>(H1[3], nullptr, H1S[3], e); });
    Bf3 H3S[2][2];
    mlp_layer<2, kEpiTanh, true, 12 % 3, true, 3, true, true>(ring, ws, lbias, H2S, H3, H3S, voff, h, nullptr,
                                                               [&](int e) { mlp_epi_pair<kEpiTanh, true>(H2[1], nullptr, H2S[1], e); });
    mlp_layer<3, kEpiNone, false, 16 % 3, true, -1, false, true>(ring, ws, lbias, H3S, Y, nullptr, voff, h, nullptr,
                                                                  [&](int e) { mlp_epi_pair<kEpiTanh, true>(H3[1], nullptr, H3S[1], e); });
    if (sv) {   // head rows 0..3 in lane half 0, 4..7 in half 1
      if (net == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) if (r + 4 * h < A) mean[s * A + r + 4 * h] = Y[0][r];
      } else if (h == 0) value[s] = Y[0][0];
    }
  }
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
