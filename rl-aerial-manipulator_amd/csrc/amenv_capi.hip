// amenv_capi.hip -- host side of libamenv.so: the C ABI declared in include/amenv.h.
// Owns the SoA episode state on one device, validates arguments, picks the kernel
// instantiation (arithmetic type x rotor count x workgroup size) and enqueues it on the
// caller's stream.  No synchronisation and no allocation after amenv_create().
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "amenv_kernels.hpp"
#include "amenv_team.hpp"
#include "amenv_team_host.hpp"
#include "amenv_quad.hpp"
#include "amenv_team_policy.hpp"
#include "amenv_quad_policy.hpp"
#include "amenv_lane_policy.hpp"
#include "amenv_obsnorm.hpp"
#include "amenv_policy.hpp"
#include "amenv_train.hpp"
#include "amenv_mlp_train.hpp"
#include "amenv_baseline.hpp"

using namespace amenv_dev;

struct amenv {
  amenv_config cfg;
  int device = -1;
  int obs_dim = 20, act_dim = kActDim, nf = 0;
  void* blob = nullptr;            // tiled episode state (amenv_kernels.hpp "Data layout")
  unsigned long long* stats = nullptr;
  size_t blob_bytes = 0, fbytes = 0, ibytes = 0;
  uint32_t tile_bytes = 0;
  int n_tiles = 0;
  int block = 64;
  bool quadk = false;              // rigid vehicles in the latency regime: lane-quad kernel (4 lanes per env, amenv_quad.hpp)
  bool armk = false;               // hexacopter + z,x,x arm between the team kernel's range and the throughput regime: stage-wave kernel (step_kernel_armk)
  bool arm2w = false;              // hexacopter + z,x,x arm at small batches: two-wave step kernel (amenv_kernels.hpp)
  bool pwave = false;              // rigid vehicles at small batches: second wave per tile computes the reset RNG words (step_kernel_pw)
  bool team = false;               // lane-team kernel (16 lanes per env): fp32 z,x,x-arm vehicle in the latency regime (amenv_team.hpp)
  void* team_consts = nullptr;     // per-lane constants of the team kernels (amenv_team_host.hpp): float4 pieces, or plain doubles for the fp64 build
  bool quad_ok = false;            // fp32 rigid vehicle with 4 or 6 rotors, single-waypoint v2 task: lane-quad kernels (step opt-in, closed-loop rollout)
  bool team_ok = false;            // the configuration has a team kernel (fp32, 6 rotors, z,x,x arm): constants are allocated
  uint32_t* pol_pack = nullptr;    // amenv_rollout_policy: policy parameters as MFMA fragments (re-packed on every call)
  // n-link arm with n < 3 (amenv_vehicle.n_joints = 1 or 2): inside, the vehicle is the 3-joint one with PHANTOM links behind the real ones (zero
  // mass / inertia / offset, joint limits 0 -> command 0, state 0: every term they add is an exact zero), so every kernel family serves it;
  // the C ABI keeps the caller's dimensions (4 + n actions, 20 + 2 n + 3 observations, 2 n joint fields): pack / unpack kernels at the boundary
  int pub_nj = 0;                  // the caller's n_joints (cfg.vehicle.n_joints is the internal 3 when this is 1 or 2)
  float* io_act = nullptr;         // [N][7]   padded actions
  float* io_obs = nullptr;         // [N][29]  internal observation rows
  float* io_term = nullptr;        // [N][29]  internal terminal-observation rows
  uint64_t steps = 0;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;  // amenv_step_timed only
  std::string err;
  std::string kname;
};

static thread_local std::string g_create_err;
constexpr int kTeamAutoMax = 6144;    // AUTO: lane-team arm kernel up to this batch (see amenv_create)
constexpr int kArmkAutoMax = 32768;   // AUTO: stage-wave arm kernel up to this batch (see amenv_create)

namespace {

constexpr int kStatsWords = kStampBase + kStampWaves * kStampSlots;  // totals + diagnostic stamp area

struct DeviceGuard {
  int prev = -1, dev;
  explicit DeviceGuard(int d) : dev(d) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
  }
};

int fail(amenv* e, int code, const std::string& msg) {
  if (e) e->err = msg; else g_create_err = msg;
  return code;
}

#define AMENV_HIP(e, call)                                                                       \
  do {                                                                                           \
    hipError_t _s = (call);                                                                      \
    if (_s != hipSuccess) return fail(e, AMENV_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_s)); \
  } while (0)

// small dense helpers (host, fp64) for the default vehicles
bool invert(int n, const double* a, double* out) {  // Gauss-Jordan, partial pivoting, n <= 4
  double m[4][8];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) { m[i][j] = a[i * n + j]; m[i][n + j] = (i == j) ? 1.0 : 0.0; }
  for (int c = 0; c < n; c++) {
    int p = c;
    for (int r = c + 1; r < n; r++) if (std::fabs(m[r][c]) > std::fabs(m[p][c])) p = r;
    if (std::fabs(m[p][c]) < 1e-300) return false;
    if (p != c) for (int j = 0; j < 2 * n; j++) std::swap(m[c][j], m[p][j]);
    const double d = m[c][c];
    for (int j = 0; j < 2 * n; j++) m[c][j] /= d;
    for (int r = 0; r < n; r++) if (r != c) { const double f = m[r][c]; for (int j = 0; j < 2 * n; j++) m[r][j] -= f * m[c][j]; }
  }
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) out[i * n + j] = m[i][n + j];
  return true;
}

// alloc = A^T (A A^T)^-1  (right pseudo-inverse of the 4 x n mixer; = A^-1 for n = 4)
bool allocation_from_mix(int n, const double* A /*[4][n]*/, double* alloc /*[n][4]*/) {
  double G[16], Gi[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) { double s = 0; for (int r = 0; r < n; r++) s += A[i * n + r] * A[j * n + r]; G[i * 4 + j] = s; }
  if (!invert(4, G, Gi)) return false;
  for (int r = 0; r < n; r++)
    for (int j = 0; j < 4; j++) { double s = 0; for (int i = 0; i < 4; i++) s += A[i * n + r] * Gi[i * 4 + j]; alloc[r * 4 + j] = s; }
  return true;
}

void fill_task_defaults(amenv_task* t, int K) {
  t->variant = AMENV_TASK_V2_SCALED20;
  t->num_waypoints = K;          // rl_env_scaledObs.py:47
  t->max_episode_steps = 2000;   // :56
  t->counter_limit = 500;        // :59
  t->rk4_substeps = 1;
  t->ee_task = AMENV_EE_TASK_BASE;
  t->dt = 1.0 / 200.0;           // :30
  const double pi = 3.14159265358979323846;
  for (int k = 1; k <= K; k++) {
    const double tt = double(k) / double(K);
    t->traj_sin[k - 1] = std::sin(2.0 * tt * pi);   // utils2/utils.py:39
    t->traj_cos[k - 1] = std::cos(tt * 2.0 * pi);   // utils2/utils.py:83-86
  }
}

// the reference's 0.18 kg quadrotor: v2/simul_files/model/params.py:10-36
bool vehicle_quad(amenv_vehicle* v) {
  v->n_rotors = 4; v->n_joints = 0;
  v->mass = 0.18; v->g = 9.81;
  const double I[9] = {0.00025, 0, 2.55e-6, 0, 0.000232, 0, 2.55e-6, 0, 0.0003738};
  std::memcpy(v->inertia, I, sizeof(I));
  if (!invert(3, I, v->inv_inertia)) return false;
  const double L = 0.086, r = 1.5e-9 / 6.11e-8;
  const double A[16] = {1, 1, 1, 1, 0, L, 0, -L, -L, 0, L, 0, r, -r, r, -r};
  std::memcpy(v->mix, A, sizeof(A));
  if (!invert(4, A, v->alloc)) return false;
  const double maxF = 2.0 * v->mass * v->g, minF = 0.0;
  for (int i = 0; i < 4; i++) { v->t_min[i] = minF / 4; v->t_max[i] = maxF / 4; }
  v->moment_scale = 0.1;  // rl_env_scaledObs.py:126
  return true;
}

// The repo's 6-rotor airframe (hexacopter_description/custom_hexa/model.sdf + airframe/4022_*):
// no reference dynamics code exists for it -> parity unpinned; constants derived in DESIGN.md "hexa".
bool vehicle_hexa(amenv_vehicle* v) {
  v->n_rotors = 6; v->n_joints = 0;
  v->mass = 2.7211; v->g = 9.81;
  // composite inertia about the CoG: 27 SDF links composed with the parallel-axis theorem by
  // tools/hexa_params.py (total 2.7211 kg, CoG z = 0.04122 m; off-diagonals < 1e-14 dropped)
  const double I[9] = {4.4024422324e-02, 0, 0, 0, 4.4059346318e-02, 0, 0, 0, 7.7083344191e-02};
  std::memcpy(v->inertia, I, sizeof(I));
  if (!invert(3, I, v->inv_inertia)) return false;
  // rotor positions (FLU, m) and spin (custom_hexa_arm/model.sdf:653,769,882,995,1108,1221; plugins :1631-1733)
  const double x[6] = {-0.255691, 0.255691, -0.255691, 0.255691, 0.0, 0.0};
  const double y[6] = {0.1475, -0.1475, -0.1475, 0.1475, 0.295, -0.295};
  const double sg[6] = {+1, -1, -1, +1, -1, +1};  // +1 cw, -1 ccw (reaction torque sign in z-up)
  const double mc = 0.0168;                       // moment constant, m
  double A[4 * 6];
  for (int r = 0; r < 6; r++) { A[0 * 6 + r] = 1.0; A[1 * 6 + r] = y[r]; A[2 * 6 + r] = -x[r]; A[3 * 6 + r] = sg[r] * mc; }
  std::memcpy(v->mix, A, sizeof(A));
  if (!allocation_from_mix(6, A, v->alloc)) return false;
  const double kf = 2.11e-5;  // N/(rad/s)^2 ; speed limits 150..820 rad/s (airframe/4022_*:64-76)
  for (int r = 0; r < 6; r++) { v->t_min[r] = kf * 150.0 * 150.0; v->t_max[r] = kf * 820.0 * 820.0; }
  v->moment_scale = 1.0;  // free parameter (no reference value): +-1 action = +-1 N m
  return true;
}

// BASELINE config 3: the hexacopter carrying the 3-joint arm (custom_hexa_arm/model.sdf:1741-1759 attaches
// Manipulator/.../sdf/manipulator.sdf to base_link).  All numbers printed by tools/arm_params.py from those SDF files.
// Parity unpinned (no reference dynamics); servo gains are this build's choice (DESIGN.md "arm").
bool vehicle_hexa_arm(amenv_vehicle* v) {
  if (!vehicle_hexa(v)) return false;
  v->n_joints = 3;
  v->mass = 3.2121;  // total: base body 2.8561 (27 hexacopter links + base_plate lump) + links 0.082 + 0.054 + 0.220
  // base body inertia about its own CoM (= body-frame origin O; model-frame z = 0.038523)
  const double I[9] = {4.4499781211e-02, 0, 8.3146456634e-06, 0, 4.4545235795e-02, 0, 8.3146456634e-06, 0, 7.7122576031e-02};
  std::memcpy(v->inertia, I, sizeof(I));
  if (!invert(3, I, v->inv_inertia)) return false;
  const double jo[9] = {0.0094667379, -0.01, -0.103522586,   // joint_1 in the body frame (manipulator.sdf:99, rel. O)
                        0.0, 0.0125, 0.0,                     // joint_2 in link 1 (:159)
                        0.0, 0.0, -0.106};                    // joint_3 in link 2 (:233)
  const double ja[9] = {0, 0, 1, 1, 0, 0, 1, 0, 0};          // axes z, x, x (:103,163,237)
  const double lm[3] = {0.082, 0.054, 0.220};                // :131,191,265 (+ 2 x 0.02 closed gripper fingers in link 3)
  const double lc[9] = {0, 0, 0, 0, 0, -0.052, 0.0058295455, -0.0054545455, -0.0822272727};
  const double li[27] = {2.10193333e-05, 0, 0, 0, 2.31308333e-05, 0, 0, 0, 3.62781667e-05,
                         1.980045e-04, 0, 0, 0, 2.266425e-04, 0, 0, 0, 3.4263e-05,
                         8.2716818176e-04, 2.6844545455e-05, 5.7916022727e-05, 2.6844545455e-05, 8.5215807772e-04, -5.1327272727e-05,
                         5.7916022727e-05, -5.1327272727e-05, 1.0384565341e-04};
  std::memcpy(v->joint_origin, jo, sizeof(jo)); std::memcpy(v->joint_axis, ja, sizeof(ja));
  std::memcpy(v->link_mass, lm, sizeof(lm)); std::memcpy(v->link_com, lc, sizeof(lc)); std::memcpy(v->link_inertia, li, sizeof(li));
  v->joint_kp = 100.0; v->joint_kd = 20.0;   // critically damped position servo, 10 rad/s bandwidth (this build's choice)
  v->joint_acc_max = 8.0;                    // manipulator_moveit/config/joint_limits.yaml:9-51
  v->joint_reserved = 0.0;
  const double lim[6] = {-3.14, 3.14, -1.57, 1.57, -1.57, 1.57};   // manipulator.sdf:105-106,165-166,239-240
  std::memcpy(v->joint_limit, lim, sizeof(lim));
  const double tool[3] = {-0.0015, 0.003, -0.125};   // midpoint of the gripper-finger joint origins in link 3's frame (manipulator.sdf:371,450)
  std::memcpy(v->tool_offset, tool, sizeof(tool));
  return true;
}

template <typename T, int NR>
HotParams<T, NR> make_hot(const amenv& e) {
  const amenv_config& c = e.cfg;
  const amenv_vehicle& v = c.vehicle;
  HotParams<T, NR> P;
  std::memset(&P, 0, sizeof(P));
  for (int r = 0; r < v.n_rotors; r++) {
    for (int j = 0; j < 4; j++) P.alloc[r][j] = T(v.alloc[r * 4 + j]);
    for (int j = 0; j < 3; j++) P.mixm[j][r] = T(v.mix[(1 + j) * v.n_rotors + r]);
    P.tmin[r] = T(v.t_min[r]); P.tmax[r] = T(v.t_max[r]);
  }
  const double* I = v.inertia; const double* J = v.inv_inertia;
  P.Ixx = T(I[0]); P.Ixy = T(I[1]); P.Ixz = T(I[2]); P.Iyy = T(I[4]); P.Iyz = T(I[5]); P.Izz = T(I[8]);
  P.Jxx = T(J[0]); P.Jxy = T(J[1]); P.Jxz = T(J[2]); P.Jyy = T(J[4]); P.Jyz = T(J[5]); P.Jzz = T(J[8]);
  const int ns = c.task.rk4_substeps > 0 ? c.task.rk4_substeps : 1;
  P.inv_mass = T(1.0 / v.mass); P.g = T(v.g); P.h = T(c.task.dt / ns);
  P.mass_f = float(v.mass); P.g_f = float(v.g); P.mscale_f = float(v.moment_scale);
  P.n_rotors = v.n_rotors; P.substeps = ns;
  P.max_steps = c.task.max_episode_steps; P.counter_limit = c.task.counter_limit;
  P.flags = c.flags; P.K = c.task.num_waypoints; P.raw_obs = c.task.variant == AMENV_TASK_V1_RAW17 ? 1 : 0;
  P.ee_task = (v.n_joints > 0 && c.task.ee_task == AMENV_EE_TASK_TOOL) ? 1 : 0;
  return P;
}

template <typename T>
ArmParams<T> make_arm(const amenv& e) {
  const amenv_vehicle& v = e.cfg.vehicle;
  ArmParams<T> A;
  std::memset(&A, 0, sizeof(A));
  for (int k = 0; k < 3; k++) {
    for (int c = 0; c < 3; c++) { A.jo[k][c] = T(v.joint_origin[3 * k + c]); A.ja[k][c] = T(v.joint_axis[3 * k + c]); A.lc[k][c] = T(v.link_com[3 * k + c]); }
    A.lm[k] = T(v.link_mass[k]);
    const double* I = &v.link_inertia[9 * k];
    A.li[k][0] = T(I[0]); A.li[k][1] = T(I[1]); A.li[k][2] = T(I[2]); A.li[k][3] = T(I[4]); A.li[k][4] = T(I[5]); A.li[k][5] = T(I[8]);
    const float lo = float(v.joint_limit[2 * k]), hi = float(v.joint_limit[2 * k + 1]);
    A.half[k] = 0.5f * (hi - lo); A.mid[k] = 0.5f * (hi + lo);

  }
  A.kp = T(v.joint_kp); A.kd = T(v.joint_kd); A.amax = T(v.joint_acc_max);
  const double zxx[9] = {0, 0, 1, 1, 0, 0, 1, 0, 0};
  A.generic_axes = std::memcmp(v.joint_axis, zxx, sizeof(zxx)) == 0 ? 0 : 1;
  A.mtot = T(v.mass); A.inv_mtot = T(1.0 / v.mass);
  for (int c = 0; c < 3; c++) {
    A.tool[c] = T(v.tool_offset[c]);
    // arm at home: every joint rotation is the identity, whatever the axes
    A.ee_home[c] = v.n_joints > 0 ? T(v.joint_origin[c] + v.joint_origin[3 + c] + v.joint_origin[6 + c] + v.tool_offset[c]) : T(0);
  }
  return A;
}

// per-lane constant table / wave-uniform parameters of the team kernels: amenv_team_host.hpp (shared with the host emulation of tests/emu)
std::vector<float> team_table_f32(const amenv_config& c) { return team_const_table<float>(c, true); }
template <typename T> TeamParamsT<T> make_team(const amenv& e) { return make_team_params<T>(e.cfg, e.team_consts); }
// the device copy of the parameters behind the per-lane table (the step kernel's source; the other team kernels take them as arguments)
template <typename T> hipError_t team_write_params(amenv* e) {
  const TeamParamsT<T> P = make_team<T>(*e);
  return hipMemcpy(static_cast<char*>(e->team_consts) + team_table_bytes<T>(), &P, sizeof(P), hipMemcpyHostToDevice);
}

QuadParams make_quad(const amenv& e) {
  const amenv_config& c = e.cfg;
  const amenv_vehicle& v = c.vehicle;
  QuadParams P;
  std::memset(&P, 0, sizeof(P));
  const int six[6] = {0, 1, 2, 4, 5, 8};
  for (int j = 0; j < 6; j++) { P.I[j] = float(v.inertia[six[j]]); P.Iinv[j] = float(v.inv_inertia[six[j]]); }
  P.inv_mass = float(1.0 / v.mass);
  const int ns = c.task.rk4_substeps > 0 ? c.task.rk4_substeps : 1;
  P.h = float(c.task.dt / ns); P.substeps = ns;
  for (int r = 0; r < 6 && r < v.n_rotors; r++) { P.tmin[r] = float(v.t_min[r]); P.tmax[r] = float(v.t_max[r]); }
  P.max_steps = c.task.max_episode_steps; P.counter_limit = c.task.counter_limit; P.flags = c.flags;
  P.ee_task = 0; P.K = 1;
  P.consts = reinterpret_cast<const float4*>(e.team_consts);
  return P;
}

ColdParams make_cold(const amenv& e) {
  const amenv_config& c = e.cfg;
  ColdParams C;
  for (int k = 0; k < AMENV_MAX_WAYPOINTS; k++) { C.traj_sin[k] = float(c.task.traj_sin[k]); C.traj_cos[k] = float(c.task.traj_cos[k]); }
  C.seed_lo = uint32_t(c.seed); C.seed_hi = uint32_t(c.seed >> 32);
  C.gid0 = c.env_id_offset;
  return C;
}

bool is_v1(const amenv_config* c) { return c->task.variant == AMENV_TASK_V1_SCALED17 || c->task.variant == AMENV_TASK_V1_RAW17; }
int obs_dim_of(const amenv_config* c) { return is_v1(c) ? 17 : 20 + 2 * c->vehicle.n_joints + (c->vehicle.n_joints ? 3 : 0); }
int act_dim_of(const amenv_config* c) { return kActDim + c->vehicle.n_joints; }

int n_float_fields(const amenv_config* c) { return AMENV_F_WP0 + 3 * c->task.num_waypoints + 2 * c->vehicle.n_joints; }

const char* validate(const amenv_config* c) {
  if (!c) return "config is NULL";
  if (c->struct_size != sizeof(amenv_config)) return "amenv_config.struct_size mismatch (ABI)";
  if (c->abi_version != AMENV_ABI_VERSION) return "amenv_config.abi_version mismatch";
  if (c->num_envs <= 0) return "num_envs must be > 0";
  if (c->dtype != AMENV_F32 && c->dtype != AMENV_F64) return "dtype must be AMENV_F32 or AMENV_F64";
  if (c->vehicle.n_rotors < 1 || c->vehicle.n_rotors > AMENV_MAX_ROTORS) return "n_rotors out of range";
  if (c->vehicle.n_joints < 0 || c->vehicle.n_joints > AMENV_MAX_JOINTS) return "n_joints must be 0..3";
  if (c->vehicle.n_joints > 0 && (c->vehicle.n_rotors != 6 || is_v1(c)))
    return "arm vehicles are built for the 6-rotor airframe and the v2 task (BASELINE config 3; 1 waypoint on every kernel, 2..4 on the lane kernel)";
  if (c->task.variant != AMENV_TASK_V2_SCALED20 && !is_v1(c)) return "unknown task variant";
  if (is_v1(c) && c->task.num_waypoints > 2) return "v1 tasks draw 1..2 waypoints per episode: num_waypoints (storage bound) must be 1 or 2";
  if (c->task.num_waypoints < 1 || c->task.num_waypoints > AMENV_MAX_WAYPOINTS) return "num_waypoints out of range";
  if (c->task.max_episode_steps < 1 || c->task.counter_limit < 0) return "bad episode limits";
  if (c->task.rk4_substeps < 0 || c->task.rk4_substeps > 64) return "rk4_substeps out of range";
  if (!(c->task.dt > 0.0) || !(c->vehicle.mass > 0.0)) return "dt and mass must be positive";
  for (int r = 0; r < c->vehicle.n_rotors; r++)
    if (std::fabs(c->vehicle.mix[r] - 1.0) > 1e-12) return "mix row 0 must be all ones: total thrust is the plain sum of rotor thrusts (quadcopter.py:111)";
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < i; j++)
      if (std::fabs(c->vehicle.inertia[i * 3 + j] - c->vehicle.inertia[j * 3 + i]) > 1e-15 ||
          std::fabs(c->vehicle.inv_inertia[i * 3 + j] - c->vehicle.inv_inertia[j * 3 + i]) > 1e-6 * std::fabs(c->vehicle.inv_inertia[i * 3 + i]))
        return "inertia and inv_inertia must be symmetric";
  // the blob holds whole 256-lane groups of tiles and padding lanes run unguarded: the workgroup size must divide 256
  if (c->block_size != 0 && c->block_size != 64 && c->block_size != 128 && c->block_size != 256) return "block_size must be 0, 64, 128 or 256";
  if (c->step_kernel < AMENV_KERNEL_AUTO || c->step_kernel > AMENV_KERNEL_STAGED) return "unknown step_kernel";
  if (c->task.ee_task != AMENV_EE_TASK_BASE && c->task.ee_task != AMENV_EE_TASK_TOOL) return "unknown task.ee_task";
  return nullptr;
}

template <typename T, int NROT, int KW, int VAR, int NJ = 0>
hipError_t launch_step(const amenv& e, const StepIO& io, int T_steps, hipStream_t s, bool timed) {
  ArmArg<T, NJ> AA;
  if constexpr (NJ > 0) AA.p = make_arm<T>(e); else AA.unused = 0;
  const HotParams<T, NROT> P = make_hot<T, NROT>(e);
  const ColdParams C = make_cold(e);
  const int bs = e.block, n_pad = e.n_tiles * 64;
  const dim3 grid((n_pad + bs - 1) / bs), block(bs);
  const size_t lds = size_t(bs) * ObsDim<VAR, NJ>::value * sizeof(float);
  const StepTail tl{io.terminal_obs, io.ep_return, io.ep_len, io.stats};
  const uint32_t tb = e.tile_bytes;
  const int32_t n = e.cfg.num_envs;
  if constexpr (NJ == 3) {
    if (e.team) {    // 16 lanes per env, 4 envs per workgroup (step: main wave + episode-end helper wave; rollout: one wave); fp64 = logic-gate build, step only
      const dim3 g2(e.n_tiles * 16), b2(64), b2s(128);
      const TeamParamsT<T> TP = make_team<T>(e);
      const float* act = reinterpret_cast<const float*>(io.actions);
      if (T_steps > 0) {
        if constexpr (sizeof(T) == 4) {
          hipLaunchKernelGGL((rollout_kernel_team<NROT>), g2, b2, 0, s, e.blob, tb, n, act, io.obs, static_cast<float*>(io.reward), io.done, io.info, T_steps, tl, C, TP);
          return hipGetLastError();
        } else {
          return hipErrorInvalidValue;   // (amenv_rollout refuses the fp64 team build before it gets here)
        }
      }
      const int32_t nb = int32_t(g2.x);   // (the step kernel reads its parameters from the device block behind the table: amenv_create wrote them there)
      if (timed) hipExtLaunchKernelGGL((step_kernel_team<T, NROT>), g2, b2s, 0, s, e.ev_start, e.ev_stop, 0, e.blob, n, nb, act, TP.consts, io.obs, static_cast<T*>(io.reward), io.done,
                                       io.info, tl, C);
      else hipLaunchKernelGGL((step_kernel_team<T, NROT>), g2, b2s, 0, s, e.blob, n, nb, act, TP.consts, io.obs, static_cast<T*>(io.reward), io.done, io.info, tl, C);
      return hipGetLastError();
    }
  }
  if constexpr (NJ == 3) {
    if (T_steps == 0 && e.armk) {
      if constexpr (sizeof(T) == 8) {   // the fp64 logic-gate build exchanges its aggregates in fp64: > 64 KB of dynamic LDS needs the attribute
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&step_kernel_armk<T, NROT>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (ea != hipSuccess) return ea;
      }    // one tile per 320-thread workgroup: four stage waves + main wave
      const dim3 g2(e.n_tiles), b2(320);
      const size_t lds2 = size_t(64 * ObsDim<VAR, NJ>::value + 12 * 64) * sizeof(float) + size_t((4 * kAggSlots + 6) * 64) * sizeof(T);   // obs rows | reset words | aggregates, joints (T)
      if (timed) hipExtLaunchKernelGGL((step_kernel_armk<T, NROT>), g2, b2, lds2, s, e.ev_start, e.ev_stop, 0, e.blob, tb, n, io.actions, io.obs, io.reward,
                                       io.done, io.info, tl, P, C, AA);
      else hipLaunchKernelGGL((step_kernel_armk<T, NROT>), g2, b2, lds2, s, e.blob, tb, n, io.actions, io.obs, io.reward, io.done, io.info, tl, P, C, AA);
      return hipGetLastError();
    }
  }
  if constexpr (NJ == 3 && sizeof(T) == 4) {
    if (T_steps == 0 && e.arm2w) {   // one tile per 128-thread workgroup: main + helper wave
      const dim3 g2(e.n_tiles), b2(128);
      const size_t lds2 = size_t(64 * ObsDim<VAR, NJ>::value + (kArmXchgSlots + 12) * 64) * sizeof(float);   // obs rows | RK4 exchange | reset words
      if (timed) hipExtLaunchKernelGGL((step_kernel_arm2w<T, NROT>), g2, b2, lds2, s, e.ev_start, e.ev_stop, 0, e.blob, tb, n, io.actions, io.obs, io.reward,
                                       io.done, io.info, tl, P, C, AA);
      else hipLaunchKernelGGL((step_kernel_arm2w<T, NROT>), g2, b2, lds2, s, e.blob, tb, n, io.actions, io.obs, io.reward, io.done, io.info, tl, P, C, AA);
      return hipGetLastError();
    }
  }
  if constexpr (NJ == 0 && sizeof(T) == 4 && KW == 1 && VAR == VAR_V2 && (NROT == 4 || NROT == 6)) {
    if (e.quadk) {   // 4 lanes per env, 16 envs per workgroup (step: main wave + episode-end helper wave; rollout: one wave)
      const dim3 g2(e.n_tiles * 4), b1(64), b2(128);
      const QuadParams QP = make_quad(e);
      if (T_steps > 0) {
        hipLaunchKernelGGL((rollout_kernel_quad<NROT>), g2, b1, 0, s, e.blob, tb, n, io.actions, io.obs, static_cast<float*>(io.reward), io.done, io.info, T_steps, tl, C, QP);
        return hipGetLastError();
      }
      if (timed) hipExtLaunchKernelGGL((step_kernel_quad<NROT>), g2, b2, 0, s, e.ev_start, e.ev_stop, 0, e.blob, tb, n, io.actions, io.obs, static_cast<float*>(io.reward), io.done,
                                       io.info, tl, C, QP);
      else hipLaunchKernelGGL((step_kernel_quad<NROT>), g2, b2, 0, s, e.blob, tb, n, io.actions, io.obs, static_cast<float*>(io.reward), io.done, io.info, tl, C, QP);
      return hipGetLastError();
    }
  }
  if constexpr (NJ == 0) {
    if (T_steps == 0 && e.pwave) {   // one tile per 128-thread workgroup: main wave + reset-RNG wave
      const dim3 g2(e.n_tiles), b2((KW == 1 && VAR == VAR_V2) ? 256 : 128);   // + observation and Monitor waves for the single-waypoint v2 task
      const size_t lds2 = size_t(64 * ObsDim<VAR, 0>::value + 12 * 64) * sizeof(float);
      if (timed) hipExtLaunchKernelGGL((step_kernel_pw<T, NROT, KW, VAR>), g2, b2, lds2, s, e.ev_start, e.ev_stop, 0, e.blob, tb, n, io.actions, io.obs, io.reward,
                                       io.done, io.info, tl, P, C);
      else hipLaunchKernelGGL((step_kernel_pw<T, NROT, KW, VAR>), g2, b2, lds2, s, e.blob, tb, n, io.actions, io.obs, io.reward, io.done, io.info, tl, P, C);
      return hipGetLastError();
    }
  }
  if (T_steps > 0) {
    hipLaunchKernelGGL((rollout_kernel<T, NROT, KW, VAR, NJ>), grid, block, lds, s, e.blob, tb, n, io.actions, io.obs, io.reward, io.done, io.info, T_steps, tl, P, C, AA);
  } else if (timed) {  // same kernel, launched with dispatch-stamped start/stop events
    hipExtLaunchKernelGGL((step_kernel<T, NROT, KW, VAR, NJ>), grid, block, lds, s, e.ev_start, e.ev_stop, 0, e.blob, tb, n, io.actions, io.obs, io.reward, io.done,
                          io.info, tl, P, C, AA);
  } else {
    hipLaunchKernelGGL((step_kernel<T, NROT, KW, VAR, NJ>), grid, block, lds, s, e.blob, tb, n, io.actions, io.obs, io.reward, io.done, io.info, tl, P, C, AA);
  }
  return hipGetLastError();
}

template <typename T, int NROT>
hipError_t dispatch_k(const amenv& e, const StepIO& io, int T_steps, hipStream_t s, bool timed) {
  if (is_v1(&e.cfg)) return launch_step<T, NROT, 2, VAR_V1>(e, io, T_steps, s, timed);   // v1: up to 2 waypoints per episode
  if (e.cfg.task.num_waypoints == 1) return launch_step<T, NROT, 1, VAR_V2>(e, io, T_steps, s, timed);
  return launch_step<T, NROT, AMENV_MAX_WAYPOINTS, VAR_V2>(e, io, T_steps, s, timed);
}

template <typename T>
hipError_t dispatch_step(const amenv& e, const StepIO& io, int T_steps, hipStream_t s, bool timed = false) {
  const int nr = e.cfg.vehicle.n_rotors;
  if (e.cfg.vehicle.n_joints == 3) {
    if (e.cfg.task.num_waypoints == 1) return launch_step<T, 6, 1, VAR_V2, 3>(e, io, T_steps, s, timed);   // BASELINE config 3
    return launch_step<T, 6, AMENV_MAX_WAYPOINTS, VAR_V2, 3>(e, io, T_steps, s, timed);                    // arm + 2..4 waypoints: the lane kernel
  }
  if (nr == 4) return dispatch_k<T, 4>(e, io, T_steps, s, timed);
  if (nr == 6) return dispatch_k<T, 6>(e, io, T_steps, s, timed);
  return dispatch_k<T, AMENV_MAX_ROTORS>(e, io, T_steps, s, timed);
}

template <typename T>
hipError_t launch_reset(const amenv& e, const uint8_t* mask, float* obs, int pad_only, hipStream_t s) {
  const int bs = 256, n_pad = e.n_tiles * 64;
  hipLaunchKernelGGL((reset_kernel<T>), dim3((n_pad + bs - 1) / bs), dim3(bs), 0, s, e.cfg.num_envs, n_pad, e.cfg.task.num_waypoints, e.cfg.task.variant, e.cfg.vehicle.n_joints,
                     e.cfg.task.ee_task, e.tile_bytes, make_cold(e), make_arm<T>(e), e.blob, mask, obs, pad_only);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_observe(const amenv& e, float* obs, float* ee, hipStream_t s) {
  const int n = e.cfg.num_envs, bs = 256, K = e.cfg.task.num_waypoints, nj = e.cfg.vehicle.n_joints;
  if (obs) hipLaunchKernelGGL((observe_kernel<T>), dim3((n + bs - 1) / bs), dim3(bs), 0, s, n, K, e.cfg.task.variant, nj, e.cfg.task.ee_task, e.tile_bytes, make_arm<T>(e),
                              (const void*)e.blob, obs);
  if (ee) hipLaunchKernelGGL((ee_position_kernel<T>), dim3((n + bs - 1) / bs), dim3(bs), 0, s, n, K, nj, e.tile_bytes, make_arm<T>(e), (const void*)e.blob, ee);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_transpose(const amenv& e, void* f, int32_t* i, int to_api, hipStream_t s) {
  const int bs = 256, n = e.cfg.num_envs;
  hipLaunchKernelGGL((transpose_state_kernel<T>), dim3((n + bs - 1) / bs), dim3(bs), 0, s, n, e.nf, e.cfg.task.num_waypoints, e.pub_nj /* the caller's joint fields: th[0..n), thd[0..n) */, e.tile_bytes, e.blob, (T*)f, i, to_api);
  return hipGetLastError();
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename V>
__global__ void calibration_copy_kernel(const V* __restrict__ src, V* __restrict__ dst, size_t n) {
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) dst[i] = src[i];
}

}  // namespace

// workspace of amenv_ppo_mlp_step: advantage partials | split-weight streams of both nets | per-workgroup gradient slabs of both nets
namespace {
constexpr size_t kMlpWsAdv = size_t(2) * kPpoMaxBlocks * sizeof(double);
constexpr size_t kMlpWsWt = size_t(2) * kMlpFragsPerNet * 1024;
constexpr size_t kMlpWsPart = size_t(2) * kMlpMaxBlocks * kAccSize * sizeof(float);
constexpr size_t kMlpLds = kMlpLdsBytes;
template <int D, int A>
hipError_t launch_mlp_step(const float* Pm, const u32x4* WS, const float* obs, const float* actions, const float* old_logp, const float* adv, const float* ret,
                           const int64_t* index, int64_t n,
                           float clip, float vf, int normalize, const double* adv_part, int adv_blocks, float* part, int blocks, hipStream_t s) {
  // > 64 KB of dynamic LDS needs the attribute on the device the launch goes to (the PPO entry points run on the CURRENT device): kept per
  // device, under a lock -- a process that drives several GPUs would otherwise launch without it on the second one
  {
    static std::mutex mu;
    static bool attr_set[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ppo_mlp_fused_kernel<D, A>), hipFuncAttributeMaxDynamicSharedMemorySize, int(kMlpLds));
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
  }
  hipLaunchKernelGGL((ppo_mlp_fused_kernel<D, A>), dim3(blocks, 2), dim3(256), kMlpLds, s, Pm, WS, obs, actions, old_logp, adv, ret, index, n, clip, vf, normalize, adv_part,
                     adv_blocks, part);
  return hipGetLastError();
}
}  // namespace

extern "C" {

const char* amenv_version(void) { return "amenv 0.2 (gfx950, abi 2)"; }

int amenv_default_config(const char* vehicle_name, int32_t num_envs, amenv_config* cfg) {
  if (!cfg || !vehicle_name) return fail(nullptr, AMENV_ERR_INVALID, "amenv_default_config: NULL argument");
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->struct_size = uint32_t(sizeof(*cfg));
  cfg->abi_version = AMENV_ABI_VERSION;
  cfg->num_envs = num_envs;
  cfg->dtype = AMENV_F32;
  cfg->flags = AMENV_FLAG_AUTO_RESET;
  cfg->block_size = 0;
  cfg->seed = 0;
  cfg->env_id_offset = 0;
  bool ok;
  if (!std::strcmp(vehicle_name, "quad")) ok = vehicle_quad(&cfg->vehicle);
  else if (!std::strcmp(vehicle_name, "hexa")) ok = vehicle_hexa(&cfg->vehicle);
  else if (!std::strcmp(vehicle_name, "hexa_arm")) ok = vehicle_hexa_arm(&cfg->vehicle);
  else return fail(nullptr, AMENV_ERR_INVALID, std::string("unknown vehicle '") + vehicle_name + "' (quad | hexa | hexa_arm)");
  if (!ok) return fail(nullptr, AMENV_ERR_INVALID, "singular vehicle matrices");
  fill_task_defaults(&cfg->task, 1);
  // north_star: "arm forward kinematics, waypoint reward" -- with the arm the task measures from the tool point
  cfg->task.ee_task = cfg->vehicle.n_joints > 0 ? AMENV_EE_TASK_TOOL : AMENV_EE_TASK_BASE;
  return AMENV_OK;
}

int amenv_config_set_task(amenv_config* cfg, int32_t variant) {
  if (!cfg) return AMENV_ERR_INVALID;
  if (variant == AMENV_TASK_V2_SCALED20) {
    fill_task_defaults(&cfg->task, 1);
  } else if (variant == AMENV_TASK_V1_SCALED17 || variant == AMENV_TASK_V1_RAW17) {
    fill_task_defaults(&cfg->task, 2);        // storage bound; K in {1,2} is drawn per episode (v1/rl_env_scaledObs.py:38)
    cfg->task.variant = variant;
    cfg->task.max_episode_steps = 1200;       // v1/rl_env_scaledObs.py:43
  } else {
    return fail(nullptr, AMENV_ERR_INVALID, "amenv_config_set_task: unknown variant");
  }
  return AMENV_OK;
}

int amenv_dims(const amenv_config* cfg, int32_t* obs_dim, int32_t* act_dim, int32_t* nff, int32_t* nif) {
  if (!cfg) return AMENV_ERR_INVALID;
  if (obs_dim) *obs_dim = obs_dim_of(cfg);
  if (act_dim) *act_dim = act_dim_of(cfg);
  if (nff) *nff = n_float_fields(cfg);
  if (nif) *nif = AMENV_I_NFIELDS;
  return AMENV_OK;
}

// DESIGN.md "Algorithmic bytes": what one env-step of amenv_step() must move through HBM.
int64_t amenv_bytes_per_env_step(const amenv_config* cfg) {
  if (!cfg) return 0;
  const int64_t ts = cfg->dtype == AMENV_F64 ? 8 : 4;
  const int64_t K = cfg->task.num_waypoints;
  const int64_t nj = cfg->vehicle.n_joints;
  const int64_t rd = ts * (13 /*state*/ + 1 /*final_yaw*/ + 1 /*last_distance*/ + 1 /*ep_return*/ + 3 * K /*waypoints*/ + 2 * nj /*joints*/) +
                     4 * 3 /*step,counter,flags*/ + 4 * act_dim_of(cfg) /*action*/;
  const int64_t wr = ts * (13 + 1 + 1 + 2 * nj) + 4 * 3 + 4 * obs_dim_of(cfg) /*obs*/ + ts /*reward*/ + 1 /*done*/ + 4 /*info*/;
  return rd + wr;
}

// ---- n-link arm, n < 3: boundary adapters (see struct amenv) ---------------------------------------------------------------------
namespace {
__global__ void arm_pad_actions_kernel(const float* __restrict__ a, int n, int nj, float* __restrict__ out) {   // [N][4 + nj] -> [N][7]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 7) return;
  const int e = i / 7, k = i % 7;
  out[i] = k < 4 + nj ? a[e * (4 + nj) + k] : 0.0f;
}
// internal rows [N][29] = 20 | th(3) | thd(3) | tool(3)  ->  public rows [N][20 + 2 nj + 3]; rows_of: only rows with a non-zero byte (terminal rows)
__global__ void arm_cut_obs_kernel(const float* __restrict__ in, int n, int nj, const uint8_t* __restrict__ rows_of, float* __restrict__ out) {
  const int od = 23 + 2 * nj;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * od) return;
  const int e = i / od, k = i % od;
  if (rows_of && !rows_of[e]) return;
  const int src = k < 20 ? k : (k < 20 + nj ? k : (k < 20 + 2 * nj ? 23 + (k - 20 - nj) : 26 + (k - 20 - 2 * nj)));
  out[i] = in[e * 29 + src];
}
// the 3-joint vehicle with phantom links behind the caller's n (1 or 2) real ones
amenv_config pad_arm_config(const amenv_config& c) {
  amenv_config p = c;
  amenv_vehicle& v = p.vehicle;
  for (int k = c.vehicle.n_joints; k < 3; k++) {
    v.link_mass[k] = 0.0;
    for (int j = 0; j < 3; j++) { v.joint_origin[3 * k + j] = 0.0; v.link_com[3 * k + j] = 0.0; v.joint_axis[3 * k + j] = j == 0 ? 1.0 : 0.0; }   // x axis: keeps a z[,x] arm on the z,x,x kernels
    for (int j = 0; j < 9; j++) v.link_inertia[9 * k + j] = 0.0;
    v.joint_limit[2 * k] = v.joint_limit[2 * k + 1] = 0.0;
  }
  v.n_joints = 3;
  return p;
}
}  // namespace

int amenv_create(const amenv_config* cfg, int device, amenv** out) {
  if (!out) return fail(nullptr, AMENV_ERR_INVALID, "amenv_create: out is NULL");
  *out = nullptr;
  if (const char* why = validate(cfg)) return fail(nullptr, AMENV_ERR_INVALID, std::string("amenv_create: ") + why);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, AMENV_ERR_NO_DEVICE, "amenv_create: no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(nullptr, AMENV_ERR_INVALID, "amenv_create: device index out of range");
  hipDeviceProp_t prop;
  AMENV_HIP(nullptr, hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, AMENV_ERR_NO_DEVICE, std::string("amenv_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  amenv* e = new (std::nothrow) amenv();
  if (!e) return fail(nullptr, AMENV_ERR_ALLOC, "amenv_create: out of host memory");
  const amenv_config user_cfg = *cfg;                       // the caller's dimensions (obs / action / state fields)
  e->pub_nj = user_cfg.vehicle.n_joints;
  e->cfg = (e->pub_nj == 1 || e->pub_nj == 2) ? pad_arm_config(user_cfg) : user_cfg;
  cfg = &e->cfg;                                            // everything below sets up the (internal) vehicle the kernels run
  e->device = device;
  e->nf = n_float_fields(&user_cfg);
  const size_t n = size_t(cfg->num_envs), ts = cfg->dtype == AMENV_F64 ? 8 : 4;
  e->fbytes = size_t(e->nf) * n * ts;
  e->ibytes = size_t(AMENV_I_NFIELDS) * n * sizeof(int32_t);
  // tiles are allocated in multiples of 4 (= 256 lanes, the largest workgroup): every launch geometry stays inside the blob and
  // every padding lane holds a valid environment (initialised below), so kernels never need a per-lane bounds branch on the state
  e->n_tiles = int((n + 255) / 256) * 4;
  e->tile_bytes = tile_bytes_for(cfg->task.num_waypoints, cfg->vehicle.n_joints, int(ts));
  e->blob_bytes = size_t(e->n_tiles) * e->tile_bytes;
  // latency regime (few waves per CU): one wave per workgroup spreads the waves over more CUs;
  // throughput regime: 256-thread workgroups
  e->block = cfg->block_size ? cfg->block_size : (cfg->num_envs <= 65536 ? 64 : 256);
  DeviceGuard g(device);
  hipError_t s;
  if ((s = hipMalloc(&e->blob, e->blob_bytes)) != hipSuccess ||
      (s = hipMalloc((void**)&e->stats, sizeof(unsigned long long) * kStatsWords)) != hipSuccess ||
      (s = hipMemset(e->blob, 0, e->blob_bytes)) != hipSuccess ||
      (s = hipMemset(e->stats, 0, sizeof(unsigned long long) * kStatsWords)) != hipSuccess ||
      // give the padding lanes of the last tile a valid state (real envs stay untouched: episode 0)
      (s = (cfg->dtype == AMENV_F64 ? launch_reset<double>(*e, nullptr, nullptr, 1, nullptr) : launch_reset<float>(*e, nullptr, nullptr, 1, nullptr))) != hipSuccess ||
      (s = hipDeviceSynchronize()) != hipSuccess) {
    std::string msg = std::string("amenv_create: device allocation failed: ") + hipGetErrorString(s);
    amenv_destroy(e);
    return fail(nullptr, AMENV_ERR_ALLOC, msg);
  }
  // Kernel choice (amenv_config.step_kernel; AUTO by batch size).  HELPER variants exist for fp32 z,x,x-arm vehicles (two-wave kernel:
  // measured faster up to 65536 envs, 18.0 vs 19.5 us there) and for rigid vehicles with the default workgroup size (reset-RNG /
  // observation helper waves: faster while the launch is latency-bound, up to 32768 envs).
  const int want = cfg->step_kernel;
  if (cfg->vehicle.n_joints == 3 && cfg->dtype == AMENV_F32) {
    const ArmParams<float> ap = make_arm<float>(*e);
    e->arm2w = !ap.generic_axes && cfg->task.num_waypoints == 1 && (want == AMENV_KERNEL_AUTO ? cfg->num_envs <= 65536 : want == AMENV_KERNEL_HELPER);
  }
  if (cfg->vehicle.n_joints == 0 && cfg->block_size == 0)
    e->pwave = want == AMENV_KERNEL_AUTO ? cfg->num_envs <= 32768 : (want == AMENV_KERNEL_HELPER && cfg->num_envs <= 64 * kStatsReplicas);
  if (want == AMENV_KERNEL_HELPER && !e->arm2w && !e->pwave) {
    amenv_destroy(e);
    return fail(nullptr, AMENV_ERR_INVALID, "amenv_create: AMENV_KERNEL_HELPER is built for fp32 z,x,x-arm vehicles and for rigid vehicles with block_size = 0 "
                "and at most 65536 envs (its Monitor wave owns one of the 1024 replicas of the running totals per tile)");
  }
  // lane-team kernel (16 lanes per env): one wavefront per SIMD up to 4096 envs; measured against the two-wave kernel on MI355X:
  // 4.9 vs 7.5 us at 2048 envs, 5.1 vs 7.6 at 4096, 7.5 vs 7.7 at 6144, 11.4 vs 7.8 at 8192 (a team workgroup is a main wave + an
  // episode-end helper wave: above 4096 envs the SIMDs hold more than two waves) -> AUTO up to 6144 envs with the body-parallel RK4.  With the
  // stage-parallel RK4 (tools/gpu_cross2.sh; against the stage-wave kernel): 4.69 vs 6.91 us at 4096 envs, 6.39 vs 6.96 at 5120, 6.64 vs 6.96 at 6144,
  // 6.71 vs 7.01 at 7168, 6.81 vs 7.07 at 8192 (two main waves per SIMD), 10.3 vs 7.2 at 10240 -> AUTO up to 8192 envs.  Round 3 (the rewritten kernel
  // keeps its per-lane constants and all loads in flight in registers: 155 VGPRs = three wavefronts per SIMD, profiles/r03/crossover_team_vs_stage_wave.txt):
  // 4.51 vs 6.93 us at 4096 envs, 5.47 vs 6.94 at 5120, 5.96 vs 6.98 at 6144, 7.04 vs 6.99 at 7168, 8.06 vs 7.04 at 8192 -> AUTO up to 6144 envs
  if (cfg->vehicle.n_joints == 3 && cfg->dtype == AMENV_F32 && cfg->vehicle.n_rotors == 6 && !make_arm<float>(*e).generic_axes && cfg->task.num_waypoints == 1)
    e->team = want == AMENV_KERNEL_AUTO ? cfg->num_envs <= kTeamAutoMax : want == AMENV_KERNEL_TEAM;
  // fp64 logic-gate build of the SAME kernel (DPP on register pairs): opt-in only, amenv_step only
  if (cfg->vehicle.n_joints == 3 && cfg->dtype == AMENV_F64 && cfg->vehicle.n_rotors == 6 && !make_arm<double>(*e).generic_axes && cfg->task.num_waypoints == 1 &&
      want == AMENV_KERNEL_TEAM)
    e->team = true;
  // lane-quad kernel (4 lanes per env) for the rigid vehicles: fp32, 4 or 6 rotors, single-waypoint v2 task, default workgroup size
  const bool quad_ok = cfg->vehicle.n_joints == 0 && cfg->dtype == AMENV_F32 && (cfg->vehicle.n_rotors == 4 || cfg->vehicle.n_rotors == 6) && !is_v1(cfg) &&
                       cfg->task.num_waypoints == 1 && cfg->block_size == 0;
  // Opt-in only (AMENV_KERNEL_TEAM), never AUTO: measured on MI355X (tools/gpu_quad.sh, profiles/r02/crossover_quad_vs_pw.txt) it ties with the
  // helper-wave kernel where the launch is latency-bound (2.85 vs 2.86 us at 1024 envs, 2.98 vs 3.00 at 4096) and loses above (3.3 vs 3.2 us at
  // 8192, 5.3 vs 4.3 at 32768: four times the wavefronts): at these sizes the rigid step sits on the dependent-launch floor (1.7 us) plus one
  // load -> compute -> store round trip of memory latency, and an instruction stream half as long changes nothing.
  e->quad_ok = quad_ok;
  e->quadk = quad_ok && want == AMENV_KERNEL_TEAM;
  if (e->quadk) e->pwave = false;
  if (want == AMENV_KERNEL_TEAM && !e->team && !e->quadk) {
    amenv_destroy(e);
    return fail(nullptr, AMENV_ERR_INVALID, "amenv_create: AMENV_KERNEL_TEAM is built for the 6-rotor vehicle with the z,x,x arm (16 lanes per env; fp64 = logic-gate build) and for "
                "fp32 rigid vehicles with 4 or 6 rotors, the single-waypoint v2 task and block_size = 0 (4 lanes per env)");
  }
  e->team_ok = cfg->vehicle.n_joints == 3 && cfg->dtype == AMENV_F32 && cfg->vehicle.n_rotors == 6 && !make_arm<float>(*e).generic_axes &&
               !is_v1(cfg) && cfg->task.num_waypoints == 1;
  // stage-wave kernel (four RK4 stage waves + a main wave per 64-env tile): the fp32 6-rotor vehicle with the z,x,x arm, single-waypoint v2
  // task, one RK4 sub-step
  const bool armk_ok = e->team_ok && cfg->task.rk4_substeps == 1;
  // measured on MI355X (tools/gpu_armk.sh, gpu_armk2.sh): 7.0 vs 7.8 us (two-wave kernel) at 6400 envs, 7.4 vs 8.1 at 12288, 9.0 vs 10.2 at 24576, 9.3 vs
  // 10.6 at 32768; level from 36864 (10.6 vs 10.8) to 49152; a CU holds three of its workgroups (49 KB of LDS each), so above 49152 envs the
  // launch takes a second round of workgroups: 14.4 vs 11.8 us at 53248
  e->armk = armk_ok && (want == AMENV_KERNEL_AUTO ? (cfg->num_envs > kTeamAutoMax && cfg->num_envs <= kArmkAutoMax) : want == AMENV_KERNEL_STAGED);
  // fp64 logic-gate build of the SAME kernel (aggregates exchanged in fp64 through LDS): opt-in only
  if (cfg->dtype == AMENV_F64 && cfg->vehicle.n_joints == 3 && cfg->vehicle.n_rotors == 6 && !make_arm<double>(*e).generic_axes && !is_v1(cfg) &&
      cfg->task.num_waypoints == 1 && cfg->task.rk4_substeps == 1 && want == AMENV_KERNEL_STAGED)
    e->armk = true;
  if (want == AMENV_KERNEL_STAGED && !e->armk) {
    amenv_destroy(e);
    return fail(nullptr, AMENV_ERR_INVALID, "amenv_create: AMENV_KERNEL_STAGED is built for the 6-rotor vehicle with the z,x,x arm (fp64 = logic-gate build), the single-waypoint v2 task "
                "and rk4_substeps = 1");
  }
  if (e->team || e->armk) e->arm2w = false;
  if (e->quad_ok) {    // the lane-quad kernels' constants (step kernel: opt-in; closed-loop rollout: amenv_rollout_policy) + the packed policy
    const std::vector<float> tc = team_table_f32(*cfg);
    if ((s = hipMalloc((void**)&e->team_consts, tc.size() * sizeof(float))) != hipSuccess ||
        (s = hipMemcpy(e->team_consts, tc.data(), tc.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess ||
        (s = hipMalloc((void**)&e->pol_pack, size_t(kPolPackWords) * sizeof(uint32_t))) != hipSuccess) {
      std::string msg = std::string("amenv_create: quad constants: ") + hipGetErrorString(s);
      amenv_destroy(e);
      return fail(nullptr, AMENV_ERR_ALLOC, msg);
    }
  }
  if (e->team_ok) {
    if ((s = hipMalloc((void**)&e->pol_pack, size_t(kPolPackWords) * sizeof(uint32_t))) != hipSuccess) {
      std::string msg = std::string("amenv_create: policy pack: ") + hipGetErrorString(s);
      amenv_destroy(e);
      return fail(nullptr, AMENV_ERR_ALLOC, msg);
    }
    const std::vector<float> tc = team_table_f32(*cfg);
    if ((s = hipMalloc((void**)&e->team_consts, team_block_bytes<float>())) != hipSuccess ||
        (s = hipMemcpy(e->team_consts, tc.data(), tc.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess ||
        (s = team_write_params<float>(e)) != hipSuccess) {
      std::string msg = std::string("amenv_create: team constants: ") + hipGetErrorString(s);
      amenv_destroy(e);
      return fail(nullptr, AMENV_ERR_ALLOC, msg);
    }
  }
  if (e->team && cfg->dtype == AMENV_F64) {
    const std::vector<double> tc = team_const_table<double>(*cfg, false);
    if ((s = hipMalloc((void**)&e->team_consts, team_block_bytes<double>())) != hipSuccess ||
        (s = hipMemcpy(e->team_consts, tc.data(), tc.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (s = team_write_params<double>(e)) != hipSuccess) {
      std::string msg = std::string("amenv_create: team constants (fp64): ") + hipGetErrorString(s);
      amenv_destroy(e);
      return fail(nullptr, AMENV_ERR_ALLOC, msg);
    }
  }
  char buf[200];
  if (e->pwave) std::snprintf(buf, sizeof(buf), "step_kernel_pw<%s,NROT=%d,KW=%d,%s> (main wave + reset wave [+ observation wave + Monitor wave] per 64-env tile)",
                              cfg->dtype == AMENV_F64 ? "double" : "float",
                              (cfg->vehicle.n_rotors == 4 || cfg->vehicle.n_rotors == 6) ? cfg->vehicle.n_rotors : AMENV_MAX_ROTORS,
                              is_v1(cfg) ? 2 : (cfg->task.num_waypoints == 1 ? 1 : AMENV_MAX_WAYPOINTS), is_v1(cfg) ? "v1" : "v2");
  else if (e->quadk) std::snprintf(buf, sizeof(buf), "step_kernel_quad<NROT=%d,v2> (4 lanes per env, 16 envs per wave + episode-end helper wave)", cfg->vehicle.n_rotors);
  else if (e->team) std::snprintf(buf, sizeof(buf), "step_kernel_team<%s,NROT=6,v2+arm3> (16 lanes per env: 4 RK4 stages x 4 components, 4 envs per wave + episode-end helper wave)",
                                  cfg->dtype == AMENV_F64 ? "double" : "float");
  else if (e->armk) std::snprintf(buf, sizeof(buf), "step_kernel_armk<%s,NROT=6> block=320 (4 RK4 stage waves + main wave per 64-env tile)", cfg->dtype == AMENV_F64 ? "double" : "float");
  else if (e->arm2w) std::snprintf(buf, sizeof(buf), "step_kernel_arm2w<float,NROT=6> block=128 (2 waves per 64-env tile)");
  else std::snprintf(buf, sizeof(buf), "step_kernel<%s,NROT=%d,KW=%d,%s> block=%d", cfg->dtype == AMENV_F64 ? "double" : "float",
                (cfg->vehicle.n_rotors == 4 || cfg->vehicle.n_rotors == 6) ? cfg->vehicle.n_rotors : AMENV_MAX_ROTORS,
                is_v1(cfg) ? 2 : (cfg->task.num_waypoints == 1 ? 1 : AMENV_MAX_WAYPOINTS), cfg->vehicle.n_joints ? "v2+arm3" : (is_v1(cfg) ? "v1" : "v2"), e->block);
  e->obs_dim = obs_dim_of(&user_cfg);
  e->act_dim = act_dim_of(&user_cfg);
  if (e->pub_nj == 1 || e->pub_nj == 2) {
    if ((s = hipMalloc((void**)&e->io_act, n * 7 * sizeof(float))) != hipSuccess || (s = hipMalloc((void**)&e->io_obs, n * 29 * sizeof(float))) != hipSuccess ||
        (s = hipMalloc((void**)&e->io_term, n * 29 * sizeof(float))) != hipSuccess) {
      std::string msg = std::string("amenv_create: n-link adapter buffers: ") + hipGetErrorString(s);
      amenv_destroy(e);
      return fail(nullptr, AMENV_ERR_ALLOC, msg);
    }
    buf[sizeof(buf) - 1] = 0;
    std::string kn = std::string(buf) + " [" + std::to_string(e->pub_nj) + "-joint arm: phantom links inside, pack / unpack at the C ABI]";
    std::snprintf(buf, sizeof(buf), "%s", kn.c_str());
  }
  e->kname = buf;
  *out = e;
  return AMENV_OK;
}

int amenv_destroy(amenv* e) {
  if (!e) return AMENV_OK;
  {
    DeviceGuard g(e->device);
    if (e->blob) (void)hipFree(e->blob);
    if (e->stats) (void)hipFree(e->stats);
    if (e->team_consts) (void)hipFree(e->team_consts);
    if (e->pol_pack) (void)hipFree(e->pol_pack);
    if (e->io_act) (void)hipFree(e->io_act);
    if (e->io_obs) (void)hipFree(e->io_obs);
    if (e->io_term) (void)hipFree(e->io_term);
    if (e->ev_start) (void)hipEventDestroy(e->ev_start);
    if (e->ev_stop) (void)hipEventDestroy(e->ev_stop);
  }
  delete e;
  return AMENV_OK;
}

const char* amenv_last_error(const amenv* e) { return e ? e->err.c_str() : g_create_err.c_str(); }
const char* amenv_kernel_name(const amenv* e) { return e ? e->kname.c_str() : ""; }

namespace {
hipError_t pad_actions(const amenv& e, const float* actions, hipStream_t s) {
  const int n = e.cfg.num_envs;
  hipLaunchKernelGGL(arm_pad_actions_kernel, dim3((n * 7 + 255) / 256), dim3(256), 0, s, actions, n, e.pub_nj, e.io_act);
  return hipGetLastError();
}
hipError_t cut_obs(const amenv& e, const float* rows29, const uint8_t* rows_of, float* out, hipStream_t s) {
  const int n = e.cfg.num_envs, od = 23 + 2 * e.pub_nj;
  hipLaunchKernelGGL(arm_cut_obs_kernel, dim3((n * od + 255) / 256), dim3(256), 0, s, rows29, n, e.pub_nj, rows_of, out);
  return hipGetLastError();
}
}  // namespace

int amenv_set_seed(amenv* e, uint64_t seed) {
  if (!e) return AMENV_ERR_INVALID;
  e->cfg.seed = seed;
  return AMENV_OK;
}

int amenv_reset(amenv* e, const uint8_t* mask, float* obs_out, void* stream) {
  if (!e) return AMENV_ERR_INVALID;
  DeviceGuard g(e->device);
  hipStream_t s = (hipStream_t)stream;
  float* o = (e->io_obs && obs_out) ? e->io_obs : obs_out;
  AMENV_HIP(e, e->cfg.dtype == AMENV_F64 ? launch_reset<double>(*e, mask, o, 0, s) : launch_reset<float>(*e, mask, o, 0, s));
  if (o != obs_out) AMENV_HIP(e, cut_obs(*e, e->io_obs, nullptr, obs_out, s));
  return AMENV_OK;
}

int amenv_observe(amenv* e, float* obs_out, void* stream) {
  if (!e || !obs_out) return fail(e, AMENV_ERR_INVALID, "amenv_observe: NULL argument");
  DeviceGuard g(e->device);
  hipStream_t s = (hipStream_t)stream;
  float* o = e->io_obs ? e->io_obs : obs_out;
  AMENV_HIP(e, e->cfg.dtype == AMENV_F64 ? launch_observe<double>(*e, o, nullptr, s) : launch_observe<float>(*e, o, nullptr, s));
  if (o != obs_out) AMENV_HIP(e, cut_obs(*e, e->io_obs, nullptr, obs_out, s));
  return AMENV_OK;
}

int amenv_ee_position(amenv* e, float* ee_out, void* stream) {
  if (!e || !ee_out) return fail(e, AMENV_ERR_INVALID, "amenv_ee_position: NULL argument");
  DeviceGuard g(e->device);
  hipStream_t s = (hipStream_t)stream;
  AMENV_HIP(e, e->cfg.dtype == AMENV_F64 ? launch_observe<double>(*e, nullptr, ee_out, s) : launch_observe<float>(*e, nullptr, ee_out, s));
  return AMENV_OK;
}

int amenv_step(amenv* e, const float* actions, float* obs, void* reward, uint8_t* done, uint32_t* info_bits, float* terminal_obs,
               float* ep_return, int32_t* ep_len, void* stream) {
  if (!e) return AMENV_ERR_INVALID;
  if (!actions || !obs || !reward || !done || !info_bits) return fail(e, AMENV_ERR_INVALID, "amenv_step: actions/obs/reward/done/info_bits must be non-NULL");
  if (!aligned16(actions) || !aligned16(obs) || (terminal_obs && !aligned16(terminal_obs)))
    return fail(e, AMENV_ERR_INVALID, "amenv_step: actions/obs/terminal_obs must be 16-byte aligned");
  DeviceGuard g(e->device);
  hipStream_t s = (hipStream_t)stream;
  StepIO io{reinterpret_cast<const float4*>(actions), obs, reward, done, info_bits, terminal_obs, ep_return, ep_len, e->stats};
  if (e->io_act) { AMENV_HIP(e, pad_actions(*e, actions, s)); io.actions = reinterpret_cast<const float4*>(e->io_act); io.obs = e->io_obs; io.terminal_obs = terminal_obs ? e->io_term : nullptr; }
  hipError_t st = e->cfg.dtype == AMENV_F64 ? dispatch_step<double>(*e, io, 0, s, false) : dispatch_step<float>(*e, io, 0, s, false);
  AMENV_HIP(e, st);
  if (e->io_act) { AMENV_HIP(e, cut_obs(*e, e->io_obs, nullptr, obs, s)); if (terminal_obs) AMENV_HIP(e, cut_obs(*e, e->io_term, done, terminal_obs, s)); }
  e->steps += uint64_t(e->cfg.num_envs);
  return AMENV_OK;
}

int amenv_step_timed(amenv* e, const float* actions, float* obs, void* reward, uint8_t* done, uint32_t* info_bits,
                     float* terminal_obs, float* ep_return, int32_t* ep_len, void* stream, float* kernel_us) {
  if (!e) return AMENV_ERR_INVALID;
  if (!actions || !obs || !reward || !done || !info_bits || !kernel_us) return fail(e, AMENV_ERR_INVALID, "amenv_step_timed: NULL argument");
  if (!aligned16(actions) || !aligned16(obs) || (terminal_obs && !aligned16(terminal_obs)))
    return fail(e, AMENV_ERR_INVALID, "amenv_step_timed: actions/obs/terminal_obs must be 16-byte aligned");
  DeviceGuard g(e->device);
  if (!e->ev_start) { AMENV_HIP(e, hipEventCreate(&e->ev_start)); AMENV_HIP(e, hipEventCreate(&e->ev_stop)); }
  StepIO io{reinterpret_cast<const float4*>(actions), obs, reward, done, info_bits, terminal_obs, ep_return, ep_len, e->stats};
  hipStream_t s = (hipStream_t)stream;
  if (e->io_act) { AMENV_HIP(e, pad_actions(*e, actions, s)); io.actions = reinterpret_cast<const float4*>(e->io_act); io.obs = e->io_obs; io.terminal_obs = terminal_obs ? e->io_term : nullptr; }
  hipError_t st = e->cfg.dtype == AMENV_F64 ? dispatch_step<double>(*e, io, 0, s, true) : dispatch_step<float>(*e, io, 0, s, true);
  AMENV_HIP(e, st);
  if (e->io_act) { AMENV_HIP(e, cut_obs(*e, e->io_obs, nullptr, obs, s)); if (terminal_obs) AMENV_HIP(e, cut_obs(*e, e->io_term, done, terminal_obs, s)); }
  AMENV_HIP(e, hipEventSynchronize(e->ev_stop));
  float ms = 0.f;
  AMENV_HIP(e, hipEventElapsedTime(&ms, e->ev_start, e->ev_stop));
  *kernel_us = ms * 1000.0f;
  e->steps += uint64_t(e->cfg.num_envs);
  return AMENV_OK;
}

int amenv_rollout(amenv* e, int32_t n_steps, const float* actions, float* obs, void* reward, uint8_t* done, uint32_t* info_bits,
                  void* stream) {
  if (!e) return AMENV_ERR_INVALID;
  if (n_steps <= 0 || !actions) return fail(e, AMENV_ERR_INVALID, "amenv_rollout: n_steps must be > 0 and actions non-NULL");
  if (!aligned16(actions) || (obs && !aligned16(obs))) return fail(e, AMENV_ERR_INVALID, "amenv_rollout: actions/obs must be 16-byte aligned");
  if (e->team && e->cfg.dtype == AMENV_F64) return fail(e, AMENV_ERR_INVALID, "amenv_rollout: the fp64 lane-team build is a logic gate of amenv_step only");
  if (e->io_act) return fail(e, AMENV_ERR_INVALID, "amenv_rollout: arms with 1 or 2 joints are served through amenv_step (the adapters at the C ABI are per step)");
  DeviceGuard g(e->device);
  StepIO io{reinterpret_cast<const float4*>(actions), obs, reward, done, info_bits, nullptr, nullptr, nullptr, e->stats};
  hipStream_t s = (hipStream_t)stream;
  hipError_t st = e->cfg.dtype == AMENV_F64 ? dispatch_step<double>(*e, io, n_steps, s) : dispatch_step<float>(*e, io, n_steps, s);
  AMENV_HIP(e, st);
  e->steps += uint64_t(e->cfg.num_envs) * uint64_t(n_steps);
  return AMENV_OK;
}

int amenv_rollout_policy(amenv* e, int32_t n_steps, const float* flat_params, uint64_t seed, uint32_t draw0, float* obs, float* actions, float* logp,
                         float* values, float* rewards, uint8_t* dones, uint32_t* info_bits, float* terminal_obs, void* stream) {
  if (!e) return AMENV_ERR_INVALID;
  if (!e->team_ok && !e->quad_ok)
    return fail(e, AMENV_ERR_INVALID, "amenv_rollout_policy: built for fp32 vehicles on the single-waypoint v2 task: rigid with 4 or 6 rotors (default workgroup size), or the "
                "6-rotor vehicle with the z,x,x arm");
  if (e->io_act) return fail(e, AMENV_ERR_INVALID, "amenv_rollout_policy: arms with 1 or 2 joints are served through amenv_step");
  if (n_steps <= 0 || !flat_params || !obs || !actions || !logp || !values || !rewards || !dones)
    return fail(e, AMENV_ERR_INVALID, "amenv_rollout_policy: n_steps must be > 0 and flat_params / obs / actions / logp / values / rewards / dones non-NULL");
  DeviceGuard g(e->device);
  hipStream_t s = (hipStream_t)stream;
  const int obs_dim = e->obs_dim, act_dim = e->act_dim;   // 29, 7
  // parameters -> bf16 MFMA fragments + per-lane action constants (they change every PPO iteration): a tiny kernel in front, no host sync
  const int pack_threads = 4 * (kPolFrags + kPolBias) * 64 + 64 + 4 * 4 * 64;
  hipLaunchKernelGGL(policy_pack_kernel, dim3((pack_threads + 255) / 256), dim3(256), 0, s, flat_params, obs_dim, act_dim, e->pol_pack);
  PolicyIO io;
  io.pack = reinterpret_cast<const uint4*>(e->pol_pack);
  io.seed_lo = uint32_t(seed); io.seed_hi = uint32_t(seed >> 32); io.draw0 = draw0;
  io.obs = obs; io.actions = actions; io.logp = logp; io.values = values; io.rewards = rewards; io.dones = dones; io.info = info_bits;
  io.terminal_obs = terminal_obs;
  if (e->quad_ok) {   // rigid vehicle: 16 envs per workgroup, the lane-quad step inside (amenv_quad_policy.hpp)
    const QuadParams QP = make_quad(*e);
    const dim3 gq(e->n_tiles * 4), bq(256);
    if (e->cfg.vehicle.n_rotors == 4) hipLaunchKernelGGL((rollout_policy_kernel_quad<4>), gq, bq, 0, s, e->blob, e->tile_bytes, e->cfg.num_envs, (int)n_steps, io, e->stats, make_cold(*e), QP);
    else hipLaunchKernelGGL((rollout_policy_kernel_quad<6>), gq, bq, 0, s, e->blob, e->tile_bytes, e->cfg.num_envs, (int)n_steps, io, e->stats, make_cold(*e), QP);
    AMENV_HIP(e, hipGetLastError());
    e->steps += uint64_t(e->cfg.num_envs) * uint64_t(n_steps);
    return AMENV_OK;
  }
  // The env part follows the step kernel's choice: 16 lanes per env where amenv_step runs the lane-team kernel (small batches), else one
  // lane per env (amenv_lane_policy.hpp; 64 envs per workgroup up to 16384 envs, 128 above: one workgroup per CU either way).
  if (!e->team) {
    ArmArg<float, 3> AA;
    AA.p = make_arm<float>(*e);
    const HotParams<float, 6> HP = make_hot<float, 6>(*e);
    if (e->cfg.num_envs <= 16384)
      hipLaunchKernelGGL((rollout_policy_kernel_lane<6, 1>), dim3(e->n_tiles), dim3(320), 0, s, e->blob, e->tile_bytes, e->cfg.num_envs, (int)n_steps, io, e->stats, HP,
                         make_cold(*e), AA);
    else
      hipLaunchKernelGGL((rollout_policy_kernel_lane<6, 2>), dim3((e->n_tiles + 1) / 2), dim3(384), 0, s, e->blob, e->tile_bytes, e->cfg.num_envs, (int)n_steps, io, e->stats,
                         HP, make_cold(*e), AA);
    AMENV_HIP(e, hipGetLastError());
    e->steps += uint64_t(e->cfg.num_envs) * uint64_t(n_steps);
    return AMENV_OK;
  }
  const TeamParams TP = make_team<float>(*e);
  // one 16-env workgroup per CU up to 4096 envs (5.15 vs 5.22 us per step there); above that the variant compiled for two wavefronts per SIMD pays
  // (measured on MI355X at 8192 envs: 7.9 vs 10.1 us per step)
  const int occ = e->cfg.num_envs <= 4096 ? 1 : 2;
  if (occ == 1) hipLaunchKernelGGL((rollout_policy_kernel_team<6, 1>), dim3(e->n_tiles * 4), dim3(256), 0, s, e->blob, e->tile_bytes, e->cfg.num_envs, (int)n_steps, io,
                                   e->stats, make_cold(*e), TP);
  else hipLaunchKernelGGL((rollout_policy_kernel_team<6, 2>), dim3(e->n_tiles * 4), dim3(256), 0, s, e->blob, e->tile_bytes, e->cfg.num_envs, (int)n_steps, io, e->stats,
                          make_cold(*e), TP);
  AMENV_HIP(e, hipGetLastError());
  e->steps += uint64_t(e->cfg.num_envs) * uint64_t(n_steps);
  return AMENV_OK;
}

int amenv_get_state(amenv* e, void* fstate, int32_t* istate, void* stream) {
  if (!e) return AMENV_ERR_INVALID;
  DeviceGuard g(e->device);
  hipStream_t s = (hipStream_t)stream;
  AMENV_HIP(e, e->cfg.dtype == AMENV_F64 ? launch_transpose<double>(*e, fstate, istate, 1, s) : launch_transpose<float>(*e, fstate, istate, 1, s));
  return AMENV_OK;
}

int amenv_set_state(amenv* e, const void* fstate, const int32_t* istate, void* stream) {
  if (!e) return AMENV_ERR_INVALID;
  DeviceGuard g(e->device);
  hipStream_t s = (hipStream_t)stream;
  AMENV_HIP(e, e->cfg.dtype == AMENV_F64 ? launch_transpose<double>(*e, const_cast<void*>(fstate), const_cast<int32_t*>(istate), 0, s)
                                          : launch_transpose<float>(*e, const_cast<void*>(fstate), const_cast<int32_t*>(istate), 0, s));
  return AMENV_OK;
}

int amenv_stats_read(amenv* e, amenv_stats* out, int reset, void* stream) {
  if (!e || !out) return fail(e, AMENV_ERR_INVALID, "amenv_stats_read: NULL argument");
  DeviceGuard g(e->device);
  hipStream_t s = (hipStream_t)stream;
  // the totals live in kStatsReplicas copies (contention-free atomics, amenv_kernels.hpp): one copy of all of them, summed here
  static_assert(S_COUNT <= kStatsStride, "stats replica stride");
  std::vector<unsigned long long> rep(size_t(kStatsReplicas) * kStatsStride);
  AMENV_HIP(e, hipMemcpyAsync(rep.data(), e->stats, rep.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  if (reset) AMENV_HIP(e, hipMemsetAsync(e->stats, 0, rep.size() * sizeof(unsigned long long), s));
  AMENV_HIP(e, hipStreamSynchronize(s));
  unsigned long long h[S_COUNT] = {0};
  for (int r = 0; r < kStatsReplicas; r++)
    for (int k = 0; k < S_COUNT; k++) h[k] += rep[size_t(r) * kStatsStride + k];
  out->steps = e->steps;
  out->episodes = h[S_EPISODES]; out->terminated = h[S_TERMINATED]; out->truncated = h[S_TRUNCATED];
  out->success = h[S_SUCCESS]; out->crashed = h[S_CRASHED]; out->oob = h[S_OOB]; out->nonfinite = h[S_NONFINITE];
  out->length_sum = h[S_LENGTH]; out->return_sum_q10 = (int64_t)h[S_RETURN_Q10];
  if (reset) e->steps = 0;
  return AMENV_OK;
}

// ---- observation normaliser ------------------------------------------------------------------------------------------
struct amenv_obsnorm {
  int dim = 0, device = 0;
  double* buf = nullptr;   // obsnorm_words(dim) doubles on the device
};

int amenv_obsnorm_create(int32_t dim, int device, amenv_obsnorm** out) {
  if (!out || dim <= 0 || dim > 1024) return fail(nullptr, AMENV_ERR_INVALID, "amenv_obsnorm_create: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(nullptr, AMENV_ERR_NO_DEVICE, "amenv_obsnorm_create: no such HIP device");
  amenv_obsnorm* h = new (std::nothrow) amenv_obsnorm();
  if (!h) return AMENV_ERR_ALLOC;
  h->dim = dim; h->device = device;
  DeviceGuard g(device);
  if (hipMalloc((void**)&h->buf, sizeof(double) * obsnorm_words(dim)) != hipSuccess) { delete h; return fail(nullptr, AMENV_ERR_ALLOC, "amenv_obsnorm_create: hipMalloc failed"); }
  std::string zero_one(sizeof(double) * obsnorm_words(dim), '\0');
  double* init = reinterpret_cast<double*>(&zero_one[0]);
  for (int j = 0; j < dim; j++) init[dim + j] = 1.0;   // var = 1
  init[2 * dim] = 1e-4;                                 // count = epsilon (sb3 RunningMeanStd default)
  if (hipMemcpy(h->buf, init, zero_one.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(h->buf); delete h; return fail(nullptr, AMENV_ERR_HIP, "amenv_obsnorm_create: init copy failed"); }
  *out = h;
  return AMENV_OK;
}

int amenv_obsnorm_destroy(amenv_obsnorm* h) {
  if (!h) return AMENV_OK;
  { DeviceGuard g(h->device); if (h->buf) (void)hipFree(h->buf); }
  delete h;
  return AMENV_OK;
}

int amenv_obsnorm_update(amenv_obsnorm* h, const float* obs, int64_t n, void* stream) {
  if (!h || !obs || n <= 0) return AMENV_ERR_INVALID;
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  const int d = h->dim, bs = 256;
  const long long n_elems = (long long)n * d;
  long long blocks = (n_elems + bs - 1) / bs;
  if (blocks > 1024) blocks = 1024;
  long long stride = blocks * bs;
  stride = ((stride + d - 1) / d) * d;             // multiple of d: a thread stays on one column
  blocks = (stride + bs - 1) / bs;
  hipLaunchKernelGGL(obsnorm_sum_kernel, dim3((unsigned)blocks), dim3(bs), sizeof(double) * 2 * d, s, obs, n_elems, d, stride, h->buf);
  hipLaunchKernelGGL(obsnorm_merge_kernel, dim3(1), dim3(((d + 63) / 64) * 64), 0, s, h->buf, d, double(n));
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

int amenv_obsnorm_apply(amenv_obsnorm* h, const float* in, float* out, int64_t n, float clip, double eps, void* stream) {
  if (!h || !in || !out || n <= 0) return AMENV_ERR_INVALID;
  DeviceGuard g(h->device);
  const int d = h->dim, bs = 256;
  const long long n_elems = (long long)n * d;
  long long blocks = (n_elems + bs - 1) / bs;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(obsnorm_apply_kernel, dim3((unsigned)blocks), dim3(bs), sizeof(double) * 2 * d, (hipStream_t)stream, in, out, n_elems, d, (const double*)h->buf, clip, eps);
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

int amenv_obsnorm_get(amenv_obsnorm* h, double* mean, double* var, double* count, void* stream) {
  if (!h || !mean || !var || !count) return AMENV_ERR_INVALID;
  DeviceGuard g(h->device);
  const int d = h->dim;
  std::string staging(sizeof(double) * (2 * d + 1), '\0');   // one synchronous copy of [mean | var | count]
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemcpy(&staging[0], h->buf, staging.size(), hipMemcpyDeviceToHost) != hipSuccess)
    return AMENV_ERR_HIP;
  const double* src = reinterpret_cast<const double*>(staging.data());
  std::memcpy(mean, src, sizeof(double) * d);
  std::memcpy(var, src + d, sizeof(double) * d);
  *count = src[2 * d];
  return AMENV_OK;
}

int amenv_obsnorm_set(amenv_obsnorm* h, const double* mean, const double* var, double count, void* stream) {
  if (!h || !mean || !var) return AMENV_ERR_INVALID;
  DeviceGuard g(h->device);
  const int d = h->dim;
  std::string staging(sizeof(double) * (2 * d + 1), '\0');
  double* dst = reinterpret_cast<double*>(&staging[0]);
  std::memcpy(dst, mean, sizeof(double) * d);
  std::memcpy(dst + d, var, sizeof(double) * d);
  dst[2 * d] = count;
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemcpy(h->buf, staging.data(), staging.size(), hipMemcpyHostToDevice) != hipSuccess)
    return AMENV_ERR_HIP;
  return AMENV_OK;
}

// ---- PPO helpers (row f3): GAE over a [T, N] rollout buffer, Gaussian action sampling ------------------------------
int amenv_gae(const float* rewards, const float* values, const uint8_t* dones, const float* last_values, float* advantages,
              float* returns, int32_t n_steps, int64_t n_envs, float gamma, float gae_lambda, void* stream) {
  if (!rewards || !values || !dones || !last_values || !advantages || !returns || n_steps <= 0 || n_envs <= 0 ||
      !(gamma >= 0.0f && gamma <= 1.0f) || !(gae_lambda >= 0.0f && gae_lambda <= 1.0f))
    return AMENV_ERR_INVALID;
  const int bs = n_envs <= 65536 ? 64 : 256;
  hipLaunchKernelGGL(gae_kernel, dim3((unsigned)((n_envs + bs - 1) / bs)), dim3(bs), 0, (hipStream_t)stream, rewards, values, dones,
                     last_values, advantages, returns, (int)n_steps, (int64_t)n_envs, gamma, gae_lambda);
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

int amenv_gaussian_act(const float* mean, const float* log_std, const float* low, const float* high, float* raw, float* clipped,
                       float* logp, int64_t n_envs, int32_t act_dim, uint64_t seed, uint32_t draw, int64_t env_id_offset, void* stream) {
  if (!mean || !log_std || !low || !high || !raw || !clipped || !logp || n_envs <= 0 || env_id_offset < 0) return AMENV_ERR_INVALID;
  const int bs = n_envs <= 65536 ? 64 : 256;
  const dim3 grid((unsigned)((n_envs + bs - 1) / bs)), block(bs);
  const uint32_t s_lo = (uint32_t)seed, s_hi = (uint32_t)(seed >> 32);
  switch (act_dim) {
    case 4: hipLaunchKernelGGL(gaussian_act_kernel<4>, grid, block, 0, (hipStream_t)stream, mean, log_std, low, high, raw, clipped, logp, (int64_t)n_envs, s_lo, s_hi, draw, (int64_t)env_id_offset); break;
    case 7: hipLaunchKernelGGL(gaussian_act_kernel<7>, grid, block, 0, (hipStream_t)stream, mean, log_std, low, high, raw, clipped, logp, (int64_t)n_envs, s_lo, s_hi, draw, (int64_t)env_id_offset); break;
    default: return AMENV_ERR_INVALID;   // 4 = quad/hexa, 7 = hexa + 3 joints
  }
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

int amenv_policy_forward(const float* flat_params, int32_t obs_dim, int32_t act_dim, const float* obs, int64_t n, float* mean_out,
                         float* value_out, void* stream) {
  if (!flat_params || !obs || n <= 0 || (!mean_out && !value_out)) return AMENV_ERR_INVALID;
  const dim3 grid((unsigned)((n + 63) / 64), 2), block(64 * kPolWaves);
  hipStream_t s = (hipStream_t)stream;
  if (obs_dim == 20 && act_dim == 4) hipLaunchKernelGGL((policy_forward_kernel<20, 4>), grid, block, 0, s, flat_params, obs, (int64_t)n, mean_out, value_out);
  else if (obs_dim == 29 && act_dim == 7) hipLaunchKernelGGL((policy_forward_kernel<29, 7>), grid, block, 0, s, flat_params, obs, (int64_t)n, mean_out, value_out);
  else if (obs_dim == 17 && act_dim == 4) hipLaunchKernelGGL((policy_forward_kernel<17, 4>), grid, block, 0, s, flat_params, obs, (int64_t)n, mean_out, value_out);
  else return AMENV_ERR_INVALID;   // (20,4) v2 | (29,7) hexacopter + arm | (17,4) v1
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

int amenv_policy_forward_mfma(const float* flat_params, int32_t obs_dim, int32_t act_dim, const float* obs, int64_t n, float* mean_out, float* value_out,
                              void* workspace, void* stream) {
  if (!flat_params || !obs || n <= 0 || (!mean_out && !value_out) || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 15u)) return AMENV_ERR_INVALID;
  if (!((obs_dim == 20 && act_dim == 4) || (obs_dim == 29 && act_dim == 7) || (obs_dim == 17 && act_dim == 4))) return AMENV_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  uint16_t* WS = reinterpret_cast<uint16_t*>(static_cast<char*>(workspace) + kMlpWsAdv);   // the split-weight area of amenv_ppo_mlp_step's workspace
  hipLaunchKernelGGL(mlp_pack_kernel, dim3((2 * kMlpPackThreadsPerNet + 255) / 256), dim3(256), 0, s, flat_params, (int)obs_dim, (int)act_dim, WS);
  const int64_t ntiles = (n + 31) / 32;
  const int nets = (mean_out ? 1 : 0) + (value_out ? 1 : 0);   // two wavefronts per SIMD (242 registers): 512 workgroups fill the chip
  const dim3 grid((unsigned)std::min<int64_t>(512 / nets, (ntiles + 3) / 4), 2), block(256);
  const u32x4* ws = reinterpret_cast<const u32x4*>(WS);
  if (obs_dim == 20) hipLaunchKernelGGL((mlp_forward_kernel<20, 4>), grid, block, 0, s, flat_params, ws, obs, (int64_t)n, mean_out, value_out);
  else if (obs_dim == 29) hipLaunchKernelGGL((mlp_forward_kernel<29, 7>), grid, block, 0, s, flat_params, ws, obs, (int64_t)n, mean_out, value_out);
  else hipLaunchKernelGGL((mlp_forward_kernel<17, 4>), grid, block, 0, s, flat_params, ws, obs, (int64_t)n, mean_out, value_out);
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

size_t amenv_ppo_workspace_bytes(void) { return size_t(kPpoMaxBlocks) * (2 * sizeof(double) + kPpoPartial * sizeof(float)); }

int amenv_ppo_loss_grad(const float* mean, const float* value, const float* log_std, const float* actions, const float* old_logp,
                        const float* advantages, const float* returns, int64_t n, int32_t act_dim, float clip_range, float ent_coef,
                        float vf_coef, int32_t normalize_advantage, float* d_mean, float* d_value, float* d_log_std, float* stats4,
                        void* workspace, void* stream) {
  if (!mean || !value || !log_std || !actions || !old_logp || !advantages || !returns || !d_mean || !d_value || !d_log_std || !stats4 ||
      !workspace || n <= 0 || !(clip_range >= 0.0f) || (reinterpret_cast<uintptr_t>(workspace) & 7u))
    return AMENV_ERR_INVALID;
  const int blocks = int(std::min<int64_t>(kPpoMaxBlocks, (n + kPpoBlock - 1) / kPpoBlock));
  double* adv_part = static_cast<double*>(workspace);
  float* part = reinterpret_cast<float*>(adv_part + 2 * kPpoMaxBlocks);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ppo_adv_partials, dim3(blocks), dim3(kPpoBlock), 0, s, advantages, (int64_t)n, adv_part);
  switch (act_dim) {
    case 4: hipLaunchKernelGGL(ppo_loss_grad<4>, dim3(blocks), dim3(kPpoBlock), 0, s, mean, value, log_std, actions, old_logp, advantages, returns,
                               (int64_t)n, clip_range, vf_coef, (int)normalize_advantage, (const double*)adv_part, blocks, d_mean, d_value, part); break;
    case 7: hipLaunchKernelGGL(ppo_loss_grad<7>, dim3(blocks), dim3(kPpoBlock), 0, s, mean, value, log_std, actions, old_logp, advantages, returns,
                               (int64_t)n, clip_range, vf_coef, (int)normalize_advantage, (const double*)adv_part, blocks, d_mean, d_value, part); break;
    default: return AMENV_ERR_INVALID;
  }
  hipLaunchKernelGGL(ppo_finalize, dim3(1), dim3(64), 0, s, (const float*)part, blocks, (int)act_dim, (int64_t)n, log_std, ent_coef, d_log_std, stats4);
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

// Bench / profiling utility: a copy of known size with a chosen access width per lane, so that the PMC traffic counters (FETCH_SIZE,
// WRITE_SIZE) can be calibrated on the access pattern of the kernel under test (16 B per lane: one-lane-per-env kernels; 4 B per lane:
// lane-team kernels) -- /opt/skills/guides/MI355X_MICROARCH.md, HBM section.
int amenv_calibration_copy(const void* src, void* dst, size_t bytes, int32_t bytes_per_lane, void* stream) {
  if (!src || !dst || (bytes_per_lane != 4 && bytes_per_lane != 16) || bytes % 16) return AMENV_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (bytes_per_lane == 4) hipLaunchKernelGGL((calibration_copy_kernel<float>), dim3(2048), dim3(256), 0, s, (const float*)src, (float*)dst, bytes / 4);
  else hipLaunchKernelGGL((calibration_copy_kernel<float4>), dim3(2048), dim3(256), 0, s, (const float4*)src, (float4*)dst, bytes / 16);
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

size_t amenv_ppo_mlp_workspace_bytes(void) { return kMlpWsAdv + kMlpWsWt + kMlpWsPart; }

int amenv_ppo_mlp_step(const float* flat_params, int32_t obs_dim, int32_t act_dim, const float* obs, const float* actions, const float* old_logp,
                       const float* advantages, const float* returns, const int64_t* index, int64_t n, float clip_range, float ent_coef, float vf_coef,
                       int32_t normalize_advantage, float* flat_grad, float* stats4, void* workspace, void* stream) {
  if (!flat_params || !obs || !actions || !old_logp || !advantages || !returns || !flat_grad || !stats4 || !workspace || n <= 0 || !(clip_range >= 0.0f) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15u))
    return AMENV_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  char* ws = static_cast<char*>(workspace);
  double* adv_part = reinterpret_cast<double*>(ws);
  uint16_t* WS = reinterpret_cast<uint16_t*>(ws + kMlpWsAdv);
  float* part = reinterpret_cast<float*>(ws + kMlpWsAdv + kMlpWsWt);
  const int adv_blocks = int(std::min<int64_t>(kPpoMaxBlocks, (n + kPpoBlock - 1) / kPpoBlock));
  const int64_t ntiles = (n + 31) / 32;
  const int blocks = int(std::min<int64_t>(128, (ntiles + 3) / 4));   // 128 x 2 nets x 4 wavefronts = one wavefront per SIMD
  hipLaunchKernelGGL(ppo_mlp_prologue_kernel, dim3(adv_blocks + (2 * kMlpPackThreadsPerNet + kPpoBlock - 1) / kPpoBlock), dim3(kPpoBlock), 0, s, advantages, (int64_t)n, adv_part, index,
                     adv_blocks, flat_params, (int)obs_dim, (int)act_dim, WS);
  hipError_t st;
  if (obs_dim == 20 && act_dim == 4) st = launch_mlp_step<20, 4>(flat_params, reinterpret_cast<const u32x4*>(WS), obs, actions, old_logp, advantages, returns, index, n, clip_range, vf_coef, normalize_advantage, adv_part, adv_blocks, part, blocks, s);
  else if (obs_dim == 29 && act_dim == 7) st = launch_mlp_step<29, 7>(flat_params, reinterpret_cast<const u32x4*>(WS), obs, actions, old_logp, advantages, returns, index, n, clip_range, vf_coef, normalize_advantage, adv_part, adv_blocks, part, blocks, s);
  else if (obs_dim == 17 && act_dim == 4) st = launch_mlp_step<17, 4>(flat_params, reinterpret_cast<const u32x4*>(WS), obs, actions, old_logp, advantages, returns, index, n, clip_range, vf_coef, normalize_advantage, adv_part, adv_blocks, part, blocks, s);
  else return AMENV_ERR_INVALID;
  if (st != hipSuccess) return AMENV_ERR_HIP;
  const int trunk = kH1 * obs_dim + kH1 + kH2 * kH1 + kH2 + kH3 * kH2 + kH3;
  const int total = act_dim + 2 * trunk + act_dim * kH3 + act_dim + kH3 + 1;
  hipLaunchKernelGGL(mlp_grad_reduce_kernel, dim3((total + 4 + 63) / 64), dim3(64 * kRedGroups), 0, s, (const float*)part, blocks, (int)obs_dim, (int)act_dim, (int64_t)n, flat_params,
                     ent_coef, flat_grad, stats4);
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

int amenv_ppo_adam_step(float* flat_params, const float* flat_grad, float* exp_avg, float* exp_avg_sq, float* step, int64_t n, const float* hyper6, float* grad_norm_out,
                        uint32_t* ticket, void* stream) {
  if (!flat_params || !flat_grad || !exp_avg || !exp_avg_sq || !step || !hyper6 || !ticket || n <= 0 || !aligned16(flat_grad)) return AMENV_ERR_INVALID;
  const int blocks = int(std::min<int64_t>(kAdamMaxBlocks, (n + kAdamBlock - 1) / kAdamBlock));
  hipLaunchKernelGGL(adam_clip_kernel, dim3(blocks), dim3(kAdamBlock), 0, (hipStream_t)stream, flat_params, (const float*)flat_grad, exp_avg, exp_avg_sq, step, (int64_t)n, hyper6,
                     grad_norm_out, ticket);
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

// Right-hand side of the arm vehicle for n states, per-link (form 0) or staged / aggregated (form 1) formulation, fp32 or fp64: the logic gate of
// the arithmetic the stage-wave and lane-team kernels run (tests compare the fp64 instantiation with the oracle's orc_arm_rhs).
int amenv_arm_rhs(const amenv_config* cfg, int32_t form, int32_t dtype, const void* state19, const void* wrench4, const void* cmd3, void* deriv19, int64_t n,
                  void* stream) {
  if (validate(cfg) || cfg->vehicle.n_joints != 3 || cfg->vehicle.n_rotors != 6 || (form != 0 && form != 1) || (dtype != AMENV_F32 && dtype != AMENV_F64) ||
      !state19 || !wrench4 || !cmd3 || !deriv19 || n <= 0)
    return AMENV_ERR_INVALID;
  amenv tmp;
  tmp.cfg = *cfg;
  if (make_arm<float>(tmp).generic_axes) return AMENV_ERR_INVALID;   // z,x,x arm
  const dim3 grid((unsigned)((n + 63) / 64)), block(64);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == AMENV_F64) {
    const HotParams<double, 6> P = make_hot<double, 6>(tmp);
    const ArmParams<double> A = make_arm<double>(tmp);
    if (form == 0) hipLaunchKernelGGL((arm_rhs_kernel<double, 0, HotParams<double, 6>>), grid, block, 0, s, P, A, (const double*)state19, (const double*)wrench4, (const double*)cmd3, (double*)deriv19, (int64_t)n);
    else hipLaunchKernelGGL((arm_rhs_kernel<double, 1, HotParams<double, 6>>), grid, block, 0, s, P, A, (const double*)state19, (const double*)wrench4, (const double*)cmd3, (double*)deriv19, (int64_t)n);
  } else {
    const HotParams<float, 6> P = make_hot<float, 6>(tmp);
    const ArmParams<float> A = make_arm<float>(tmp);
    if (form == 0) hipLaunchKernelGGL((arm_rhs_kernel<float, 0, HotParams<float, 6>>), grid, block, 0, s, P, A, (const float*)state19, (const float*)wrench4, (const float*)cmd3, (float*)deriv19, (int64_t)n);
    else hipLaunchKernelGGL((arm_rhs_kernel<float, 1, HotParams<float, 6>>), grid, block, 0, s, P, A, (const float*)state19, (const float*)wrench4, (const float*)cmd3, (float*)deriv19, (int64_t)n);
  }
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

// ---- PID + minimum-snap baseline controller (row f4; csrc/amenv_baseline.hpp) ---------------------------------------------------------
int amenv_pid_default_params(amenv_pid_params* p) {
  if (!p) return AMENV_ERR_INVALID;
  static const double kGains[18] = {3.0, 30.0, 1.0, 3.0, 30.0, 1.0, 1000.0, 200.0, 10.0,      // x, y, z       (pid_controller.py:16-18)
                                    160.0, 3.0, 1.0, 160.0, 3.0, 1.0, 80.0, 5.0, 1.0};        // phi theta psi (:19-21)
  p->dt = 0.01; p->mass = 0.18; p->g = 9.81; p->max_integral = 100.0;
  std::memcpy(p->gain, kGains, sizeof(kGains));
  return AMENV_OK;
}

namespace {
bool pid_params_ok(const amenv_pid_params* p) { return p && p->dt > 0.0 && p->mass > 0.0 && p->g > 0.0 && p->max_integral >= 0.0; }
PidParams to_dev(const amenv_pid_params& p) {
  PidParams d;
  d.dt = p.dt; d.mass = p.mass; d.g = p.g; d.max_integral = p.max_integral;
  for (int k = 0; k < 6; k++) for (int j = 0; j < 3; j++) d.gain[k][j] = p.gain[3 * k + j];
  return d;
}
}  // namespace

int amenv_pid_run(const amenv_pid_params* p, int32_t dtype, const void* state, const void* des, void* integral, void* F_out, void* M_out,
                  void* rpy_out, int64_t n, void* stream) {
  if (!pid_params_ok(p) || !state || !des || !integral || !F_out || !M_out || n <= 0 || (dtype != AMENV_F32 && dtype != AMENV_F64)) return AMENV_ERR_INVALID;
  const dim3 grid((unsigned)((n + 63) / 64)), block(64);
  const PidParams d = to_dev(*p);
  if (dtype == AMENV_F64)
    hipLaunchKernelGGL(pid_run_kernel<double>, grid, block, 0, (hipStream_t)stream, d, (const double*)state, (const double*)des, (double*)integral,
                       (double*)F_out, (double*)M_out, (double*)rpy_out, (int64_t)n);
  else
    hipLaunchKernelGGL(pid_run_kernel<float>, grid, block, 0, (hipStream_t)stream, d, (const float*)state, (const float*)des, (float*)integral,
                       (float*)F_out, (float*)M_out, (float*)rpy_out, (int64_t)n);
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

size_t amenv_minsnap_workspace_bytes(int32_t n_segments) {
  return (n_segments < 1 || n_segments > 16) ? 0 : size_t(8 * n_segments) * size_t(10 * n_segments) * sizeof(double);
}

int amenv_minsnap_solve(int32_t n_segments, int64_t n_traj, double speed, const double* waypoints, double* coeff, double* seg_time,
                        double* seg_start, void* workspace, void* stream) {
  if (n_segments < 1 || n_segments > 16 || n_traj <= 0 || !(speed > 0.0) || !waypoints || !coeff || !seg_time || !seg_start || !workspace)
    return AMENV_ERR_INVALID;
  hipLaunchKernelGGL(minsnap_inverse_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (int)n_segments, (double*)workspace);
  const int64_t rows = n_traj * 8 * n_segments;
  hipLaunchKernelGGL(minsnap_coeff_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)n_segments, (int64_t)n_traj,
                     speed, (const double*)workspace, waypoints, coeff, seg_time, seg_start);
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

int amenv_minsnap_eval(int32_t n_segments, int64_t n_query, const double* coeff, const double* seg_time, const double* seg_start,
                       const double* waypoints, const int64_t* traj, const double* t, int32_t dtype, void* des, void* stream) {
  if (n_segments < 1 || n_segments > 16 || n_query <= 0 || !coeff || !seg_time || !seg_start || !waypoints || !t || !des ||
      (dtype != AMENV_F32 && dtype != AMENV_F64))
    return AMENV_ERR_INVALID;
  const dim3 grid((unsigned)((n_query + 63) / 64)), block(64);
  if (dtype == AMENV_F64)
    hipLaunchKernelGGL(minsnap_eval_kernel<double>, grid, block, 0, (hipStream_t)stream, (int)n_segments, (int64_t)n_query, coeff, seg_time, seg_start,
                       waypoints, traj, t, (double*)des);
  else
    hipLaunchKernelGGL(minsnap_eval_kernel<float>, grid, block, 0, (hipStream_t)stream, (int)n_segments, (int64_t)n_query, coeff, seg_time, seg_start,
                       waypoints, traj, t, (float*)des);
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

int amenv_pid_policy(const amenv_pid_policy_params* p, int32_t dtype, const float* obs, const uint8_t* done, void* pstate, float* actions,
                     int64_t n, void* stream) {
  if (!p || !pid_params_ok(&p->pid) || !obs || !pstate || !actions || n <= 0 || (dtype != AMENV_F32 && dtype != AMENV_F64) || !(p->speed > 0.0) ||
      !(p->moment_scale > 0.0) || p->obs_dim < 20 || p->act_dim < 4 || p->act_dim > 16 || (p->tool_mode != 0 && p->obs_dim < 29))
    return AMENV_ERR_INVALID;
  PidPolicyParams d;
  d.pid = to_dev(p->pid);
  d.speed = p->speed; d.moment_scale = p->moment_scale;
  for (int k = 0; k < 3; k++) d.m_gain[k] = p->inertia_ratio[k];
  d.obs_dim = p->obs_dim; d.act_dim = p->act_dim; d.tool_mode = p->tool_mode ? 1 : 0; d.pad = 0;
  const dim3 grid((unsigned)((n + 63) / 64)), block(64);
  if (dtype == AMENV_F64) hipLaunchKernelGGL(pid_policy_kernel<double>, grid, block, 0, (hipStream_t)stream, d, obs, done, (double*)pstate, actions, (int64_t)n);
  else hipLaunchKernelGGL(pid_policy_kernel<float>, grid, block, 0, (hipStream_t)stream, d, obs, done, (float*)pstate, actions, (int64_t)n);
  return hipGetLastError() == hipSuccess ? AMENV_OK : AMENV_ERR_HIP;
}

#ifdef AMENV_STAMPS
// diagnostic build only: an empty kernel with the step kernel's grid, to measure the dependent-launch floor
__global__ void noop_kernel(void* blob, uint32_t tile_bytes, int32_t n, unsigned long long* sink) {
  if (n < 0) sink[0] = tile_bytes;
}
__global__ void touch_kernel(void* blob, uint32_t tile_bytes, int32_t n, unsigned long long* sink) {
  // one 16-B load + one 16-B store per lane: the minimal memory round trip
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int4* p = reinterpret_cast<int4*>(static_cast<char*>(blob) + size_t(i >> 6) * tile_bytes) + (threadIdx.x & 63);
  int4 v = *p; v.w ^= 0; *p = v;
}
int amenv_debug_noop(amenv* e, int which, int block, void* stream) {
  DeviceGuard g(e->device);
  const int bs = block > 0 ? block : e->block, n_pad = e->n_tiles * 64;
  if (which >= 2) hipLaunchKernelGGL(noop_kernel, dim3(which), dim3(bs), 0, (hipStream_t)stream, e->blob, e->tile_bytes, e->cfg.num_envs, e->stats);   // explicit grid
  else if (which == 0) hipLaunchKernelGGL(noop_kernel, dim3((n_pad + bs - 1) / bs), dim3(bs), 0, (hipStream_t)stream, e->blob, e->tile_bytes, e->cfg.num_envs, e->stats);
  else hipLaunchKernelGGL(touch_kernel, dim3((n_pad + bs - 1) / bs), dim3(bs), 0, (hipStream_t)stream, e->blob, e->tile_bytes, e->cfg.num_envs, e->stats);
  AMENV_HIP(e, hipGetLastError());
  return AMENV_OK;
}

// diagnostic build only: copy the per-wave s_memtime stamps of the last step launch to host (synchronises)
int amenv_debug_stamps(amenv* e, unsigned long long* host_out /*[64][8]*/) {
  DeviceGuard g(e->device);
  AMENV_HIP(e, hipDeviceSynchronize());
  AMENV_HIP(e, hipMemcpy(host_out, e->stats + kStampBase, sizeof(unsigned long long) * kStampWaves * kStampSlots, hipMemcpyDeviceToHost));
  return AMENV_OK;
}
#endif

}  // extern "C"
