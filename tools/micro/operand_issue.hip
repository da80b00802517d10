// Micro-benchmark: cost per v_fma_f32 for ONE wave alone with 1, 2 or 3 VGPR source operands (4 and 8 independent chains).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NC, int NV>
__global__ __launch_bounds__(64) void k(float* out, const float* in, int iters) {
  float a[NC], x[NC], y[NC];
  for (int i = 0; i < NC; i++) { a[i] = threadIdx.x * 1e-3f + i; x[i] = in[threadIdx.x + i] ; y[i] = in[threadIdx.x + 64 + i]; }
  const float b = 0.999f, c = 1e-3f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 64 / NC; r++) {
#pragma unroll
      for (int i = 0; i < NC; i++) {
        if (NV == 1) a[i] = __builtin_fmaf(a[i], b, c);
        else if (NV == 2) a[i] = __builtin_fmaf(a[i], x[i], c);
        else a[i] = __builtin_fmaf(a[i], x[i], y[i]);
      }
    }
  }
  float s = 0; for (int i = 0; i < NC; i++) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NC, int NV> void run(float* out, float* in) {
  const int iters = 4000;
  for (int rep = 0; rep < 2; rep++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    hipLaunchKernelGGL((k<NC, NV>), dim3(64), dim3(64), 0, 0, out, (const float*)in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("chains %d, %d VGPR source(s): %.2f ns per FMA instruction\n", NC, NV, ms * 1e6 / (double(iters) * 64));
  }
}

int main() {
  float *out, *in; hipMalloc(&out, 64 * 64 * 4); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
  run<1, 1>(out, in); run<1, 3>(out, in);
  run<4, 1>(out, in); run<4, 2>(out, in); run<4, 3>(out, in);
  run<8, 1>(out, in); run<8, 2>(out, in); run<8, 3>(out, in);
  return 0;
}
