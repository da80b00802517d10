// Micro-benchmark: issue cost of v_mfma_f32_32x32x2_f32 on gfx950 in the patterns the fused PPO kernel uses.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Prints wall clocks (s_memtime) per MFMA for: one dependent accumulator chain, two / four independent chains, a chain with N vector-ALU
// instructions between MFMAs, a chain with one L2-resident global load per MFMA, and 1 / 2 / 4 wavefronts per SIMD... (one workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS, int VALU, int LOADS>
__global__ __launch_bounds__(512) void k(const float* __restrict__ w, float* out, unsigned long long* clk, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.0f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f, x = a;
  const float* p = w + (threadIdx.x & 63);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        float av = a;
        if (LOADS) av = p[((i * 16 + j) * CHAINS + c) % 512 * 64];
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[c], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < VALU; v++) x = __builtin_fmaf(x, 1.0001f, 0.5f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = x;
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int CHAINS, int VALU, int LOADS>
void run(const char* name, int threads, const float* w, float* out, unsigned long long* clk) {
  const int iters = 200, blocks = 256;
  hipLaunchKernelGGL((k<CHAINS, VALU, LOADS>), dim3(blocks), dim3(threads), 0, 0, w, out, clk, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, VALU, LOADS>), dim3(blocks), dim3(threads), 0, 0, w, out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += double(v); avg /= blocks;
  const double n = double(iters) * 16 * CHAINS;
  printf("%-44s waves/CU %d: %7.1f clocks per MFMA per wave, kernel %.1f us -> %.2f GHz counter, %.1f TFLOP/s\n", name, threads / 64, avg / n, ms * 1e3,
         avg / (ms * 1e6), n * (threads / 64) * blocks * 4096.0 / (ms * 1e-3) * 1e-12);
}

int main() {
  float *w, *out; unsigned long long* clk;
  hipMalloc(&w, 512 * 64 * 4 + 1024); hipMemset(w, 0, 512 * 64 * 4 + 1024); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 256 * 8);
  for (int threads : {256, 512}) {
    run<1, 0, 0>("1 chain", threads, w, out, clk);
    run<2, 0, 0>("2 chains", threads, w, out, clk);
    run<4, 0, 0>("4 chains", threads, w, out, clk);
    run<1, 8, 0>("1 chain + 8 VALU per MFMA", threads, w, out, clk);
    run<1, 14, 0>("1 chain + 14 VALU per MFMA", threads, w, out, clk);
    run<2, 14, 0>("2 chains + 14 VALU per MFMA", threads, w, out, clk);
    run<1, 0, 1>("1 chain + 1 global load per MFMA", threads, w, out, clk);
    run<2, 0, 1>("2 chains + 1 global load per MFMA", threads, w, out, clk);
    run<2, 8, 1>("2 chains + 8 VALU + 1 load per MFMA", threads, w, out, clk);
  }
  return 0;
}
