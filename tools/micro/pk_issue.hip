// Micro-benchmark: issue cost of v_fma_f32 vs v_pk_fma_f32 for ONE wave alone on a SIMD (the regime of the 4096-env launch).
// hipcc --offload-arch=gfx950 -O3 tools/micro/pk_issue.hip -o tools/micro/pk_issue && ./tools/micro/pk_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: 8 independent scalar FMA chains, 1: 4 independent packed chains (same number of FMAs), 2: 8 packed chains (2x FMAs)
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, int iters) {
  float a[8]; float2_ p[8];
  for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 1e-3f + i; p[i] = float2_{a[i], a[i] + 0.5f}; }
  const float b = 0.999f, c = 1e-3f; const float2_ b2{b, b}, c2{c, c};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = __builtin_fmaf(a[i], b, c);
      } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) p[i] = __builtin_elementwise_fma(p[i], b2, c2);
      } else {
#pragma unroll
        for (int i = 0; i < 8; i++) p[i] = __builtin_elementwise_fma(p[i], b2, c2);
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 64 * 64 * 4); hipMalloc(&cyc, 64 * 8);
  const int iters = 2000; unsigned long long h[64];
  for (int rep = 0; rep < 2; rep++) {
    for (int mode = 0; mode < 3; mode++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(64), dim3(64), 0, 0, out, cyc, iters);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(64), dim3(64), 0, 0, out, cyc, iters);
      else hipLaunchKernelGGL(k<2>, dim3(64), dim3(64), 0, 0, out, cyc, iters);
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      const double ninst = double(iters) * 16 * (mode == 1 ? 4 : 8);
      if (rep) printf("mode %d: %.1f us, %.2f ns per instruction, %.2f ns per FMA\n", mode, ms * 1e3, ms * 1e6 / ninst, ms * 1e6 / (ninst * (mode ? 2 : 1)));
    }
  }
  return 0;
}
