// amenv_baseline.hpp -- the PID + minimum-snap baseline controller on the GPU.
// What the reference runs one vehicle at a time in Python (v2 = initial-implementation-v2):
//   * pid_core / pid_run_kernel:   `v2/PID Controller/pid_controller.py:37-115` (cascaded PID; gains :16-21, integral clamp :34,66-67,107-108)
//                                  with the attitude it reads through `Quadcopter.attitude()` (model/quadcopter.py:57-59 -> utils/utils.py:11-15)
//   * minsnap_inverse_kernel,
//     minsnap_coeff_kernel:        `v2/PID Controller/trajGen3D.py` MST (:211-292): the constraint matrix of n 7th-order segments depends on n
//                                  only, so its inverse is formed once (Gauss-Jordan, partial pivoting, fp64, one workgroup) and every
//                                  trajectory's 8n x 3 coefficients are one small matrix product
//   * minsnap_eval_kernel:         generate_trajectory (:76-187; ends with yaw = yawdot = 0, :183-184)
//   * pid_policy_kernel:           the two wired to the waypoint environment (`runsim.py:26-31` flies its waypoint list the same way): per
//                                  episode a one-segment rest-to-rest minimum-snap trajectory to the waypoint, tracked by the PID; observation
//                                  row in, action row out, ONE launch per control step (the torch restatement in baselines.py needs ~80)
// One lane per vehicle; nothing here is bandwidth- or ALU-critical (a launch is a few microseconds of latency), so the code is written for
// exact agreement with the reference's arithmetic order, fp64 build = logic gate, fp32 build = product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amenv_dev {

struct PidParams {
  double dt, mass, g, max_integral;
  double gain[6][3];       // (k_p, k_d, k_i) for x, y, z, phi, theta, psi
};

struct PidPolicyParams {
  PidParams pid;
  double speed, moment_scale;
  double m_gain[3];        // inertia of this vehicle / inertia of the reference quadrotor, per axis
  int32_t obs_dim, act_dim, tool_mode, pad;
};

template <typename T> __device__ __forceinline__ T clamp_(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float bl_sin(float a) { return sinf(a); }
__device__ __forceinline__ double bl_sin(double a) { return sin(a); }
__device__ __forceinline__ float bl_cos(float a) { return cosf(a); }
__device__ __forceinline__ double bl_cos(double a) { return cos(a); }
__device__ __forceinline__ float bl_asin(float a) { return asinf(a); }
__device__ __forceinline__ double bl_asin(double a) { return asin(a); }
__device__ __forceinline__ float bl_atan2(float a, float b) { return atan2f(a, b); }
__device__ __forceinline__ double bl_atan2(double a, double b) { return atan2(a, b); }
__device__ __forceinline__ float bl_sqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ double bl_sqrt(double a) { return sqrt(a); }

// RotToRPY(R(q / |q|)) (utils/utils.py:11-15): phi = asin(R[1,2]), theta = atan2(-R[0,2] / cos phi, R[2,2] / cos phi),
// psi = atan2(-R[1,0] / cos phi, R[1,1] / cos phi), from the five entries of R it reads.
template <typename T>
__device__ __forceinline__ void rot_to_rpy(T w, T x, T y, T z, T& phi, T& theta, T& psi) {
  const T inv = T(1) / bl_sqrt(w * w + x * x + y * y + z * z);
  w *= inv; x *= inv; y *= inv; z *= inv;
  const T r12 = T(2) * (y * z - w * x), r02 = T(2) * (x * z + w * y), r22 = T(1) - T(2) * (x * x + y * y);
  const T r10 = T(2) * (x * y + w * z), r11 = T(1) - T(2) * (x * x + z * z);
  phi = bl_asin(clamp_(r12, T(-1), T(1)));
  const T c = bl_cos(phi);
  theta = bl_atan2(-r02 / c, r22 / c);
  psi = bl_atan2(-r10 / c, r11 / c);
}

// pid_controller.run (:51-113) for one vehicle.  I[6] = the module's integral memory (x y z phi theta psi), updated in place.
template <typename T>
__device__ __forceinline__ void pid_core(const PidParams& P, const T pos[3], const T vel[3], T phi, T theta, T psi, const T om[3],
                                         const T dpos[3], const T dvel[3], const T dacc[3], T dyaw, T dyawdot, T I[6], T& F, T M[3]) {
  const T dt = T(P.dt), mi = T(P.max_integral);
  T acc[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const T ep = dpos[k] - pos[k], ev = dvel[k] - vel[k];                  // :51-56
    I[k] = clamp_(I[k] + ep * dt, -mi, mi);                                // :59-67
    acc[k] = dacc[k] + T(P.gain[k][1]) * ev + T(P.gain[k][0]) * ep + T(P.gain[k][2]) * I[k];   // :70-83
  }
  F = T(P.mass) * (T(P.g) + acc[2]);                                       // :86
  const T s = bl_sin(dyaw), c = bl_cos(dyaw);
  const T des_phi = T(1) / T(P.g) * (acc[0] * s - acc[1] * c);             // :89
  const T des_theta = T(1) / T(P.g) * (acc[0] * c + acc[1] * s);           // :90
  const T ea[3] = {des_phi - phi, des_theta - theta, dyaw - psi};          // :91-96
  const T ew[3] = {-om[0], -om[1], dyawdot - om[2]};                       // :97-99
#pragma unroll
  for (int k = 0; k < 3; k++) {
    I[3 + k] = clamp_(I[3 + k] + ea[k] * dt, -mi, mi);                     // :102-108
    M[k] = T(P.gain[3 + k][0]) * ea[k] + T(P.gain[3 + k][1]) * ew[k] + T(P.gain[3 + k][2]) * I[3 + k];   // :111-115
  }
}

// state [n, 13] (p, v, q = (w, x, y, z), body rates: Quadcopter.state), des [n, 11] (pos, vel, acc, yaw, yawdot), integral [n, 6] in/out.
template <typename T>
__global__ __launch_bounds__(256) void pid_run_kernel(PidParams P, const T* __restrict__ state, const T* __restrict__ des, T* __restrict__ integral,
                                                      T* __restrict__ F_out, T* __restrict__ M_out, T* __restrict__ rpy_out, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T* s = state + i * 13;
  const T* d = des + i * 11;
  T I[6];
#pragma unroll
  for (int k = 0; k < 6; k++) I[k] = integral[i * 6 + k];
  T phi, theta, psi, F, M[3];
  rot_to_rpy<T>(s[6], s[7], s[8], s[9], phi, theta, psi);
  pid_core<T>(P, s, s + 3, phi, theta, psi, s + 10, d, d + 3, d + 6, d[9], d[10], I, F, M);
#pragma unroll
  for (int k = 0; k < 6; k++) integral[i * 6 + k] = I[k];
  F_out[i] = F;
#pragma unroll
  for (int k = 0; k < 3; k++) M_out[i * 3 + k] = M[k];
  if (rpy_out) { rpy_out[i * 3] = phi; rpy_out[i * 3 + 1] = theta; rpy_out[i * 3 + 2] = psi; }
}

// ---- minimum snap -----------------------------------------------------------------------------------------------------------------

// get_poly_cc(8, k, t)[i] (:189-209): coefficient of a_i in the k-th derivative of sum_i a_i t^i at t.
__device__ __forceinline__ double poly_cc8(int k, int i, double t) {
  if (i < k) return 0.0;
  double c = 1.0;
  for (int j = 0; j < k; j++) c *= double(i - j);
  for (int j = 0; j < i - k; j++) c *= t;
  return c;
}

// MST's constraint matrix A (8n x 8n, rows in the reference's order :262-281) next to the first 2n columns of the identity, reduced by
// Gauss-Jordan with partial pivoting: W [8n][8n + 2n] -> its right block = the first 2n columns of A^-1 (only the 2n waypoint rows of
// the right-hand side are non-zero, :252-260).  One workgroup of 256 threads; n <= 16.
__global__ __launch_bounds__(256) void minsnap_inverse_kernel(int n, double* __restrict__ W) {
  const int N = 8 * n, LD = N + 2 * n, tid = threadIdx.x;
  __shared__ double s_val[256];
  __shared__ int s_idx[256];
  for (int e = tid; e < N * LD; e += 256) {
    const int r = e / LD, c = e % LD;
    double v = 0.0;
    if (c >= N) v = (c - N == r) ? 1.0 : 0.0;
    else {
      const int seg = c / 8, i = c % 8;
      if (r < n) v = (seg == r) ? poly_cc8(0, i, 0.0) : 0.0;                               // segment i starts on waypoint i
      else if (r < 2 * n) v = (seg == r - n) ? poly_cc8(0, i, 1.0) : 0.0;                  // ... and ends on waypoint i + 1
      else if (r < 2 * n + 3) v = (seg == 0) ? poly_cc8(r - 2 * n + 1, i, 0.0) : 0.0;      // rest at the start (derivatives 1..3)
      else if (r < 2 * n + 6) v = (seg == n - 1) ? poly_cc8(r - 2 * n - 2, i, 1.0) : 0.0;  // rest at the end
      else {                                                                               // derivatives 1..6 continuous at knot j
        const int j = (r - 2 * n - 6) / 6, k = (r - 2 * n - 6) % 6 + 1;
        if (seg == j) v = poly_cc8(k, i, 1.0);
        else if (seg == j + 1) v = -poly_cc8(k, i, 0.0);
      }
    }
    W[e] = v;
  }
  __syncthreads();
  for (int c = 0; c < N; c++) {
    double best = -1.0;
    int bi = c;
    for (int r = c + tid; r < N; r += 256) { const double a = fabs(W[r * LD + c]); if (a > best) { best = a; bi = r; } }
    s_val[tid] = best; s_idx[tid] = bi;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
      if (tid < h && (s_val[tid + h] > s_val[tid] || (s_val[tid + h] == s_val[tid] && s_idx[tid + h] < s_idx[tid]))) {
        s_val[tid] = s_val[tid + h]; s_idx[tid] = s_idx[tid + h];
      }
      __syncthreads();
    }
    const int p = s_idx[0];
    const double piv = W[p * LD + c];
    __syncthreads();
    for (int j = tid; j < LD; j += 256) {          // swap rows c <-> p, scale the pivot row
      const double a = W[p * LD + j], b = W[c * LD + j];
      W[p * LD + j] = b;
      W[c * LD + j] = a / piv;
    }
    __syncthreads();
    for (int e = tid; e < N * (LD - c); e += 256) {  // eliminate column c from every other row (columns < c are already unit vectors)
      const int r = e / (LD - c), j = c + e % (LD - c);
      if (r != c) {
        const double f = W[r * LD + c];
        if (j > c && f != 0.0) W[r * LD + j] -= f * W[c * LD + j];
      }
    }
    __syncthreads();
    for (int r = tid; r < N; r += 256) if (r != c) W[r * LD + c] = 0.0;
    __syncthreads();
  }
}

// coeff [B][8n][3] = A^-1[:, :2n] . [w_0 .. w_{n-1}; w_1 .. w_n]; segment times T [B][n] = |w_i - w_{i+1}| / speed and their running
// sum S [B][n + 1] (:97-103).  waypoints [B][n + 1][3].
__global__ __launch_bounds__(256) void minsnap_coeff_kernel(int n, int64_t B, double speed, const double* __restrict__ W, const double* __restrict__ wp,
                                                            double* __restrict__ coeff, double* __restrict__ T, double* __restrict__ S) {
  const int N = 8 * n, LD = N + 2 * n;
  const int64_t e = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= B * N) return;
  const int64_t b = e / N;
  const int r = int(e % N);
  const double* w = wp + b * (n + 1) * 3;
  double cx = 0.0, cy = 0.0, cz = 0.0;
  for (int j = 0; j < 2 * n; j++) {
    const double a = W[r * LD + N + j];
    const double* q = w + 3 * (j < n ? j : j - n + 1);
    cx += a * q[0]; cy += a * q[1]; cz += a * q[2];
  }
  coeff[e * 3] = cx; coeff[e * 3 + 1] = cy; coeff[e * 3 + 2] = cz;
  if (r == 0) {
    double s = 0.0;
    S[b * (n + 1)] = 0.0;
    for (int i = 0; i < n; i++) {
      const double dx = w[3 * i] - w[3 * i + 3], dy = w[3 * i + 1] - w[3 * i + 4], dz = w[3 * i + 2] - w[3 * i + 5];
      const double t = sqrt(dx * dx + dy * dy + dz * dz) / speed;
      T[b * n + i] = t;
      s += t;
      S[b * (n + 1) + i + 1] = s;
    }
  }
}

// generate_trajectory(t) for m queries: query i evaluates trajectory traj[i] (traj NULL: trajectory i) at time tq[i];
// des [m][11] = pos, vel, acc, yaw, yawdot.  t == 0 -> the first waypoint at rest (:108-111), t past the end -> the last (:121,176-179).
template <typename T>
__global__ __launch_bounds__(256) void minsnap_eval_kernel(int n, int64_t m, const double* __restrict__ coeff, const double* __restrict__ Tseg,
                                                           const double* __restrict__ S, const double* __restrict__ wp, const int64_t* __restrict__ traj,
                                                           const double* __restrict__ tq, T* __restrict__ des) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int64_t b = traj ? traj[i] : i;
  const double t = tq[i];
  const double* Sb = S + b * (n + 1);
  int idx = -1;
  for (int k = 0; k <= n; k++) idx += (t >= Sb[k]) ? 1 : 0;                 // np.where(t >= S)[0][-1] (:105)
  idx = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
  const double Ti = Tseg[b * n + idx], scale = (t - Sb[idx]) / Ti;            // :123
  const double* c = coeff + (b * 8 * n + 8 * idx) * 3;
  double p[3] = {0, 0, 0}, v[3] = {0, 0, 0}, a[3] = {0, 0, 0};
  double pw[8];                                                               // scale^k
  pw[0] = 1.0;
  for (int k = 1; k < 8; k++) pw[k] = pw[k - 1] * scale;
  for (int k = 0; k < 8; k++)
    for (int x = 0; x < 3; x++) {
      p[x] += pw[k] * c[k * 3 + x];
      if (k >= 1) v[x] += double(k) * pw[k - 1] * c[k * 3 + x];
      if (k >= 2) a[x] += double(k * (k - 1)) * pw[k - 2] * c[k * 3 + x];
    }
  const bool first = (t == 0.0), after = (t > Sb[n]);
  const double* w0 = wp + b * (n + 1) * 3;
  T* d = des + i * 11;
  for (int x = 0; x < 3; x++) {
    d[x] = T(first ? w0[x] : (after ? w0[3 * n + x] : p[x]));
    d[3 + x] = T((first || after) ? 0.0 : v[x] / Ti);
    d[6 + x] = T((first || after) ? 0.0 : a[x] / (Ti * Ti));
  }
  d[9] = T(0); d[10] = T(0);                                                  // :183-184
}

// ---- the baseline as an action source for the waypoint environment ------------------------------------------------------------------
// pst [n][14]: t, start (3), goal (3), integral (6), fresh (1 = the next call starts an episode).  obs rows in the v2 layout
// (v2/rl_env_scaledObs.py:98-121: p/10, v/5, q, w/5, (waypoint - task point)/2, ...; with the arm obs[26:29] = (tool - base)/0.5).
// tool_mode = 1 (arm vehicle whose task measures from the tool point): the position loop tracks the TOOL point.
template <typename T>
__global__ __launch_bounds__(256) void pid_policy_kernel(PidPolicyParams P, const float* __restrict__ obs, const uint8_t* __restrict__ done,
                                                         T* __restrict__ pst, float* __restrict__ actions, int64_t n) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* o = obs + i * P.obs_dim;
  T* s = pst + i * 14;
  T pos[3], vel[3], om[3], goal[3], q[4];
#pragma unroll
  for (int k = 0; k < 3; k++) { pos[k] = T(o[k]) * T(10); vel[k] = T(o[3 + k]) * T(5); om[k] = T(o[10 + k]) * T(5); }
#pragma unroll
  for (int k = 0; k < 4; k++) q[k] = T(o[6 + k]);
  if (P.tool_mode) {
#pragma unroll
    for (int k = 0; k < 3; k++) pos[k] += T(o[26 + k]) * T(0.5);
  }
#pragma unroll
  for (int k = 0; k < 3; k++) goal[k] = pos[k] + T(o[13 + k]) * T(2);
  T t = s[0], start[3] = {s[1], s[2], s[3]}, g0[3] = {s[4], s[5], s[6]}, I[6];
#pragma unroll
  for (int k = 0; k < 6; k++) I[k] = s[7 + k];
  const bool fresh = (s[13] != T(0)) || (done && done[i]);
  if (fresh) {
#pragma unroll
    for (int k = 0; k < 3; k++) { start[k] = pos[k]; g0[k] = goal[k]; }
    t = T(0);
#pragma unroll
    for (int k = 0; k < 6; k++) I[k] = T(0);
  }
  // waypoint switched inside an episode (multi-waypoint tasks): a new segment from the current position
  const T mx = goal[0] - g0[0], my = goal[1] - g0[1], mz = goal[2] - g0[2];
  if (bl_sqrt(mx * mx + my * my + mz * mz) > T(1e-3)) {
#pragma unroll
    for (int k = 0; k < 3; k++) { start[k] = pos[k]; g0[k] = goal[k]; }
    t = T(0);
  }
  const T d[3] = {g0[0] - start[0], g0[1] - start[1], g0[2] - start[2]};
  T Tt = bl_sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) / T(P.speed);
  Tt = Tt < T(P.pid.dt) ? T(P.pid.dt) : Tt;
  const T tau = clamp_(t / Tt, T(0), T(1));
  // one segment, rest to rest: MST's solution is s(tau) = 35 tau^4 - 84 tau^5 + 70 tau^6 - 20 tau^7
  const T t2 = tau * tau, t3 = t2 * tau, t4 = t2 * t2;
  const T s0 = t4 * (T(35) + tau * (T(-84) + tau * (T(70) - T(20) * tau)));
  const T s1 = t3 * (T(140) + tau * (T(-420) + tau * (T(420) - T(140) * tau))) / Tt;
  const T s2 = t2 * (T(420) + tau * (T(-1680) + tau * (T(2100) - T(840) * tau))) / (Tt * Tt);
  T dpos[3], dvel[3], dacc[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { dpos[k] = start[k] + d[k] * s0; dvel[k] = d[k] * s1; dacc[k] = d[k] * s2; }
  T phi, theta, psi, F, M[3];
  rot_to_rpy<T>(q[0], q[1], q[2], q[3], phi, theta, psi);
  pid_core<T>(P.pid, pos, vel, phi, theta, psi, om, dpos, dvel, dacc, T(0), T(0), I, F, M);
  t = t + T(P.pid.dt);
  s[0] = t;
#pragma unroll
  for (int k = 0; k < 3; k++) { s[1 + k] = start[k]; s[4 + k] = g0[k]; }
#pragma unroll
  for (int k = 0; k < 6; k++) s[7 + k] = I[k];
  s[13] = T(0);
  // the reference hands F, M to the mixer unclipped (runsim.py:30); the env's action box clips per axis, so the moment VECTOR is
  // scaled into the box (direction kept) before the clip
  T am[3], big = T(1);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    am[k] = M[k] * T(P.m_gain[k]) / T(P.moment_scale);
    const T a = am[k] < T(0) ? -am[k] : am[k];
    big = a > big ? a : big;
  }
  float* a = actions + i * P.act_dim;
  a[0] = float(clamp_(F / T(P.pid.mass * P.pid.g), T(0), T(2)));
#pragma unroll
  for (int k = 0; k < 3; k++) a[1 + k] = float(clamp_(am[k] / big, T(-1), T(1)));
  for (int k = 4; k < P.act_dim; k++) a[k] = 0.0f;          // arm joints: commanded to the middle of their range (home)
}

}  // namespace amenv_dev
