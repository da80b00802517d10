// Micro-benchmark: v_mfma_f32_32x32x16_bf16 on gfx950 -- issue cost of a dependent chain, and whether vector-ALU work of the same wavefront
// (or of a second wavefront on the SIMD) runs in the shadow of the matrix pipe.  (fp32 MFMA does not: mfma_rate.hip.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_bf16_rate.hip -o /tmp/mfma_bf16_rate && /tmp/mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, int VALU, int DEP>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* clk, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.0f;
  bf16x8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = __bf16(threadIdx.x * 1e-3f + j); b[j] = __bf16(1.0f + j * 1e-2f); }
  float x[4] = {threadIdx.x * 1e-3f, 1.0f, 2.0f, 3.0f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < VALU; v++) { float& y = x[DEP ? 0 : v & 3]; y = __builtin_fmaf(y, 1.0001f, 0.5f); }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = x[0] + x[1] + x[2] + x[3];
  for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int CHAINS, int VALU, int DEP>
void run(const char* name, int threads, float* out, unsigned long long* clk) {
  const int iters = 200, blocks = 256;
  hipLaunchKernelGGL((k<CHAINS, VALU, DEP>), dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, VALU, DEP>), dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += double(v); avg /= blocks;
  const double n = double(iters) * 16 * CHAINS;
  printf("%-44s waves/CU %d: %7.1f counter ticks per MFMA per wave, kernel %.1f us, %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n", name, threads / 64, avg / n, ms * 1e3,
         ms * 1e6 / (n * (threads / 256)), n * (threads / 64) * blocks * 32768.0 / (ms * 1e-3) * 1e-12);
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 256 * 8);
  for (int threads : {256, 512}) {
    run<1, 0, 0>("1 chain", threads, out, clk);
    run<2, 0, 0>("2 chains", threads, out, clk);
    run<1, 4, 1>("1 chain + 4 dependent VALU per MFMA", threads, out, clk);
    run<1, 4, 0>("1 chain + 4 independent VALU per MFMA", threads, out, clk);
    run<1, 8, 0>("1 chain + 8 independent VALU per MFMA", threads, out, clk);
    run<2, 8, 0>("2 chains + 8 independent VALU per MFMA", threads, out, clk);
    run<1, 16, 0>("1 chain + 16 independent VALU per MFMA", threads, out, clk);
  }
  return 0;
}
