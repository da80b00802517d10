// Checks the operand / result lane maps of v_mfma_f32_16x16x32_bf16 that csrc/amenv_team_policy.hpp relies on, with exact small integers
// and an asymmetric pattern:  lane l holds A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15]; D: col = l&15, row = 4(l>>4)+reg.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_layout_check.hip -o tools/micro/mfma_layout_check && tools/micro/mfma_layout_check
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {   // A [16][32], B [32][16], D [16][16] row-major
  const int l = threadIdx.x, r = l & 15, kq = l >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; j++) { a[j] = (__bf16)A[r * 32 + 8 * kq + j]; b[j] = (__bf16)B[(8 * kq + j) * 16 + r]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int q = 0; q < 4; q++) D[(4 * kq + q) * 16 + r] = c[q];
}
int main() {
  float hA[512], hB[512], hD[256], ref[256];
  for (int i = 0; i < 16; i++) for (int kk = 0; kk < 32; kk++) hA[i * 32 + kk] = float((i * 7 + kk * 3) % 5 - 2);
  for (int kk = 0; kk < 32; kk++) for (int j = 0; j < 16; j++) hB[kk * 16 + j] = float((kk * 5 + j * 11) % 7 - 3);
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { float s = 0; for (int kk = 0; kk < 32; kk++) s += hA[i * 32 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
  float e = 0, et = 0;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { e = fmaxf(e, fabsf(hD[i * 16 + j] - ref[i * 16 + j])); et = fmaxf(et, fabsf(hD[j * 16 + i] - ref[i * 16 + j])); }
  printf("max |D - A.B| with the assumed maps: %g   (transposed result would give %g)\n", e, et);
  return e == 0.0f ? 0 : 1;
}
