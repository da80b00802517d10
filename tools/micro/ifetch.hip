// Micro-benchmark: does instruction fetch limit a lone wave running long straight-line code?  16384 FMAs (8 interleaved chains) per wave
// and launch, as a loop over a body of BODY FMAs (code size = BODY * 8 bytes): 64 (512 B) .. 16384 (128 KB, executed once).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int BODY>
__global__ __launch_bounds__(64) void k(float* out) {
  float a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 1e-3f + i;
  const float b = 0.999f, c = 1e-3f;
#pragma nounroll
  for (int it = 0; it < 16384 / BODY; it++) {
#pragma unroll
    for (int r = 0; r < BODY / 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = __builtin_fmaf(a[i], b, c);
    }
  }
  float s = 0; for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int BODY> void run(float* out) {
  for (int rep = 0; rep < 2; rep++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    for (int l = 0; l < 200; l++) hipLaunchKernelGGL(k<BODY>, dim3(64), dim3(64), 0, 0, out);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("body %6d FMAs (%7d B of code): %.2f us per launch, %.2f ns per FMA\n", BODY, BODY * 8, ms * 1e3 / 200, ms * 1e6 / 200 / 16384);
  }
}

int main() {
  float* out; hipMalloc(&out, 64 * 64 * 4);
  run<64>(out); run<1024>(out); run<4096>(out); run<16384>(out);
  return 0;
}
