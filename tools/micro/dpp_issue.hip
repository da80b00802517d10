// Micro-benchmark: clocks per instruction for ONE wave alone -- plain FMAs against DPP-carrying instructions (v_mov_b32_dpp, v_add_f32_dpp,
// v_mul_f32_dpp), as dependent chains and as NC interleaved independent chains.  The lane-team kernels spend a third of their instructions on DPP.
// hipcc --offload-arch=gfx950 -O3 tools/micro/dpp_issue.hip -o tools/micro/dpp_issue && ./tools/micro/dpp_issue
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float dppq(float v) {   // quad_perm [1,2,0,3]
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xC9, 0xF, 0xF, true));
}
__device__ __forceinline__ float dppr(float v) {   // row_ror:4
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
}

// MODE 0: a = fma(a, b, c)            1: a = dpp_quad(a) (v_mov_b32_dpp)      2: a = a + dpp_quad(a) (v_add_f32_dpp)
//      3: a = a * dpp_quad(a) * b -> v_mul_f32_dpp + v_mul   4: a = fma(a, b, c); a = a + dpp_quad(a)   5: row_ror add   6: a = fma(dpp_quad(a) [mov], b, c)
template <int MODE, int NC>
__global__ __launch_bounds__(64) void k(float* out, long long* clk, int iters) {
  float a[NC];
  for (int i = 0; i < NC; i++) a[i] = threadIdx.x * 1e-3f + i + 1.0f;
  const float b = 0.9999f, c = 1e-4f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 32 / NC; r++) {
#pragma unroll
      for (int i = 0; i < NC; i++) {
        if (MODE == 0) a[i] = __builtin_fmaf(a[i], b, c);
        else if (MODE == 1) { a[i] = dppq(a[i]); asm volatile("" : "+v"(a[i])); }
        else if (MODE == 2) a[i] = a[i] + dppq(a[i]);
        else if (MODE == 3) a[i] = (b * dppq(a[i]));
        else if (MODE == 4) { a[i] = __builtin_fmaf(a[i], b, c); a[i] = a[i] + dppq(a[i]); }
        else if (MODE == 5) a[i] = a[i] * b + dppr(a[i]);
        else if (MODE == 6) { float t = dppq(a[i]); asm volatile("" : "+v"(t)); a[i] = __builtin_fmaf(t, b, c); }
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < NC; i++) s += a[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE, int NC> void run(float* out, long long* clk, const char* what, int per) {
  const int iters = 2000;
  long long h = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL((k<MODE, NC>), dim3(1), dim3(64), 0, 0, out, clk, iters);
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  }
  printf("%-58s chains %2d: %6.2f counter ticks per instruction\n", what, NC, double(h) / (double(iters) * 32 * per));
}

int main() {
  float* out; long long* clk; hipMalloc(&out, 4096); hipMalloc(&clk, 8);
  run<0, 1>(out, clk, "v_fma_f32", 1); run<0, 2>(out, clk, "v_fma_f32", 1); run<0, 4>(out, clk, "v_fma_f32", 1); run<0, 8>(out, clk, "v_fma_f32", 1);
  run<1, 1>(out, clk, "v_mov_b32_dpp quad_perm", 1); run<1, 4>(out, clk, "v_mov_b32_dpp quad_perm", 1); run<1, 8>(out, clk, "v_mov_b32_dpp quad_perm", 1);
  run<2, 1>(out, clk, "v_add_f32_dpp quad_perm", 1); run<2, 2>(out, clk, "v_add_f32_dpp quad_perm", 1); run<2, 4>(out, clk, "v_add_f32_dpp quad_perm", 1); run<2, 8>(out, clk, "v_add_f32_dpp quad_perm", 1);
  run<3, 1>(out, clk, "v_mul_f32_dpp quad_perm", 1); run<3, 4>(out, clk, "v_mul_f32_dpp quad_perm", 1); run<3, 8>(out, clk, "v_mul_f32_dpp quad_perm", 1);
  run<4, 1>(out, clk, "v_fma_f32 ; v_add_f32_dpp (per instruction)", 2); run<4, 4>(out, clk, "v_fma_f32 ; v_add_f32_dpp (per instruction)", 2); run<4, 8>(out, clk, "v_fma_f32 ; v_add_f32_dpp (per instruction)", 2);
  run<5, 1>(out, clk, "v_mul ; v_add_f32_dpp row_ror:4 (per instruction)", 2); run<5, 4>(out, clk, "v_mul ; v_add_f32_dpp row_ror:4 (per instruction)", 2);
  run<6, 1>(out, clk, "v_mov_b32_dpp ; v_fma_f32 (per instruction)", 2); run<6, 4>(out, clk, "v_mov_b32_dpp ; v_fma_f32 (per instruction)", 2); run<6, 8>(out, clk, "v_mov_b32_dpp ; v_fma_f32 (per instruction)", 2);
  printf("(the cycle counter of this build: __builtin_readcyclecounter = s_memtime, 100 MHz reference or shader clock -- compare ratios)\n");
  return 0;
}
