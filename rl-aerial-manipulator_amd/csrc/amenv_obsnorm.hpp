// amenv_obsnorm.hpp -- GPU running observation normaliser, the on-device equivalent of
// `VecNormalize(env, norm_obs=True, norm_reward=False)` (v1/rl_train_vecN.py:10-11, v1/rl_checkpoint_train_vecN.py:19-26).
// Third-party semantics (stable-baselines3 2.6.0, common/running_mean_std.py + vec_normalize.py; not vendored by the
// reference, its vec_normalize.pkl is a pickle and is never loaded): running mean / population variance / count
// (initial 0 / 1 / 1e-4) merged with each batch by the parallel-moments formula, then
// obs_n = clip((obs - mean) / sqrt(var + 1e-8), -10, 10).  Parity unpinned; checked against a numpy restatement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amenv_dev {

// buffer layout (doubles): [0,d) mean | [d,2d) var | [2d] count | [2d+1, 3d+1) batch sum | [3d+1, 4d+1) batch sum of squares
__host__ __device__ inline int obsnorm_words(int d) { return 4 * d + 1; }

// Column sums of a row-major [n, d] f32 matrix.  Consecutive lanes read consecutive elements (coalesced); the grid
// stride is a multiple of d, so a thread stays on one column and accumulates in registers (fp64); per-column block
// totals are combined with LDS atomics and one global atomic per block and column.
__global__ __launch_bounds__(256) void obsnorm_sum_kernel(const float* __restrict__ obs, long long n_elems, int d, long long stride,
                                                          double* __restrict__ buf) {
  extern __shared__ double lds_sums[];  // [2d]
  for (int j = threadIdx.x; j < 2 * d; j += blockDim.x) lds_sums[j] = 0.0;
  __syncthreads();
  const long long e0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e0 < stride) {  // stride is rounded up to a multiple of d: lanes past it would break the fixed-column property
    const int col = int(e0 % d);
    double s = 0.0, q = 0.0;
    for (long long e = e0; e < n_elems; e += stride) { const double x = double(obs[e]); s += x; q += x * x; }
    atomicAdd(&lds_sums[col], s);
    atomicAdd(&lds_sums[d + col], q);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < 2 * d; j += blockDim.x) atomicAdd(&buf[2 * d + 1 + j], lds_sums[j]);
}

// RunningMeanStd.update_from_moments (sb3 common/running_mean_std.py) with the batch moments from the sums; clears the sums.
__global__ void obsnorm_merge_kernel(double* __restrict__ buf, int d, double batch_count) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const double count = buf[2 * d];
  if (j < d) {
    const double b_mean = buf[2 * d + 1 + j] / batch_count;
    double b_var = buf[3 * d + 1 + j] / batch_count - b_mean * b_mean;   // np.var: population variance
    b_var = b_var > 0.0 ? b_var : 0.0;
    const double mean = buf[j], var = buf[d + j];
    const double delta = b_mean - mean, tot = count + batch_count;
    const double m2 = var * count + b_var * batch_count + delta * delta * count * batch_count / tot;
    buf[j] = mean + delta * batch_count / tot;
    buf[d + j] = m2 / tot;
    buf[2 * d + 1 + j] = 0.0; buf[3 * d + 1 + j] = 0.0;
  }
  __syncthreads();
  if (j == 0) buf[2 * d] = count + batch_count;   // single block: every lane has read `count` before the barrier
}

// VecNormalize.normalize_obs: clip((obs - mean) / sqrt(var + eps), -clip, clip); in place allowed.
__global__ __launch_bounds__(256) void obsnorm_apply_kernel(const float* __restrict__ in, float* __restrict__ out, long long n_elems, int d,
                                                            const double* __restrict__ buf, float clip, double eps) {
  // mean and 1/sigma stay fp64: a nearly constant column (quaternion w ~ 1, var ~ 1e-7) has 1/sigma ~ 3e3, which would
  // amplify an fp32-rounded mean into the 1e-4 range
  extern __shared__ double lds_md[];  // [2d]: mean, 1/sqrt(var+eps)
  for (int j = threadIdx.x; j < d; j += blockDim.x) { lds_md[j] = buf[j]; lds_md[d + j] = 1.0 / sqrt(buf[d + j] + eps); }
  __syncthreads();
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_elems; e += stride) {
    const int col = int(e % d);
    float v = float((double(in[e]) - lds_md[col]) * lds_md[d + col]);
    v = v < -clip ? -clip : (v > clip ? clip : v);
    out[e] = v;
  }
}

}  // namespace amenv_dev
