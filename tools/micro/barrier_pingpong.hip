// Micro-benchmark: cost of the two-wave exchange pattern of step_kernel_arm2w on gfx950.
// One 128-thread workgroup per CU-ish (grid = 64): each "RHS" = W dependent FMAs, helper puts 21 floats, barrier, main reads 21,
// W2 FMAs, puts 6, barrier, helper reads 6.  Compared with the same FMA chains without any exchange.
// hipcc --offload-arch=gfx950 -O3 tools/micro/barrier_pingpong.hip -o gpurun_out/barrier_pingpong && ./gpurun_out/barrier_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>  // 0: no exchange, 1: exchange with __syncthreads
__global__ __launch_bounds__(128) void k(float* out, unsigned long long* cyc, int iters, int w_main, int w_help) {
  __shared__ float lds[27 * 64];
  const int lane = threadIdx.x & 63, role = __builtin_amdgcn_readfirstlane(int(threadIdx.x) >> 6);
  float a = 1.0f + lane * 1e-3f, b = 0.999f;
  float v[21];
  for (int q = 0; q < 21; q++) v[q] = a + q;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    const int w = role ? w_help : w_main;
    for (int j = 0; j < w; j++) a = __builtin_fmaf(a, b, 1e-3f);
    if (MODE == 1) {
      if (role) {
        for (int q = 0; q < 21; q++) lds[q * 64 + lane] = v[q] + a;
        __syncthreads();
        __syncthreads();
        for (int q = 0; q < 6; q++) a += lds[(21 + q) * 64 + lane];
      } else {
        __syncthreads();
        for (int q = 0; q < 21; q++) a += lds[q * 64 + lane];
        for (int j = 0; j < 100; j++) a = __builtin_fmaf(a, b, 1e-3f);
        for (int q = 0; q < 6; q++) lds[(21 + q) * 64 + lane] = a + q;
        __syncthreads();
      }
    } else {
      if (!role) for (int j = 0; j < 100; j++) a = __builtin_fmaf(a, b, 1e-3f);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 128 + threadIdx.x] = a;
  if (lane == 0) cyc[blockIdx.x * 2 + role] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 128 * 4); hipMalloc(&cyc, 64 * 2 * 8);
  std::vector<unsigned long long> h(128);
  const int iters = 64;
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 2; mode++)
      for (int wm : {400}) for (int wh : {100, 350, 500}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode) hipLaunchKernelGGL(k<1>, dim3(64), dim3(128), 0, 0, out, cyc, iters, wm, wh);
        else hipLaunchKernelGGL(k<0>, dim3(64), dim3(128), 0, 0, out, cyc, iters, wm, wh);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, 128 * 8, hipMemcpyDeviceToHost);
        if (rep) printf("mode %d w_main %d(+100) w_help %d: kernel %.2f us, per-iter main %.0f helper %.0f (s_memtime ticks @100MHz? raw)\n", mode, wm, wh, ms * 1e3,
                        double(h[0]) / iters, double(h[1]) / iters);
      }
  return 0;
}
