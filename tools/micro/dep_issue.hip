// Micro-benchmark: cost per v_fma_f32 for ONE wave alone as a function of the number of independent dependency chains it interleaves.
// hipcc --offload-arch=gfx950 -O3 tools/micro/dep_issue.hip -o tools/micro/dep_issue && ./tools/micro/dep_issue
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NC>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  float a[NC];
  for (int i = 0; i < NC; i++) a[i] = threadIdx.x * 1e-3f + i;
  const float b = 0.999f, c = 1e-3f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 64 / NC; r++) {
#pragma unroll
      for (int i = 0; i < NC; i++) a[i] = __builtin_fmaf(a[i], b, c);
    }
  }
  float s = 0; for (int i = 0; i < NC; i++) s += a[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NC> void run(float* out, int waves_per_block) {
  const int iters = 4000;
  for (int rep = 0; rep < 2; rep++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    hipLaunchKernelGGL(k<NC>, dim3(64), dim3(64 * waves_per_block), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("chains %d, %d wave(s)/workgroup: %.2f ns per FMA instruction per wave\n", NC, waves_per_block, ms * 1e6 / (double(iters) * 64));
  }
}

int main() {
  float* out; hipMalloc(&out, 64 * 512 * 4);
  run<1>(out, 1); run<2>(out, 1); run<4>(out, 1); run<8>(out, 1); run<16>(out, 1);
  run<8>(out, 2); run<8>(out, 4); run<8>(out, 8); run<1>(out, 2); run<1>(out, 4);
  return 0;
}
