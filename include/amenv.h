/*
 * amenv.h -- C ABI of libamenv.so: the MI355X-resident batched waypoint environment.
 *
 * This is the drop-in boundary for ONE path of LahiruCooray/rl-aerial-manipulator:
 * `WaypointQuadEnv.step()/reset()` and the `simul_files` rigid-body integrator, stepped
 * for N independent environments per call instead of one.  The reference has no FFI
 * (it is pure Python); every entry point below names the reference interface it
 * replaces as  <file>:<line>  relative to the reference root, with
 *   v2 = initial-implementation-v2,  v1 = initial-implementation-v1.
 *
 * Conventions
 *  - plain C: pointers, sizes, fixed-width ints.  No torch / HIP types in signatures:
 *    a stream is passed as `void*` (a hipStream_t; NULL = the default stream).
 *  - all I/O buffers are DEVICE pointers owned by the caller (e.g. torch tensors);
 *    the library owns only its internal struct-of-arrays episode state.
 *  - every call only ENQUEUES work on the given stream: no hidden synchronisation, no
 *    allocation after amenv_create(), safe to capture into a hipGraph.
 *  - return value: 0 = AMENV_OK, negative = error code; text via amenv_last_error().
 *    Nothing throws, nothing calls exit().
 *  - one handle per device; calls on one handle are serialised by the caller.
 *    There is no global state: 8 handles on 8 GPUs can be driven from 8 processes/threads.
 */
#ifndef AMENV_H_
#define AMENV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMENV_ABI_VERSION 2

#define AMENV_MAX_ROTORS 8
#define AMENV_MAX_WAYPOINTS 4
#define AMENV_MAX_JOINTS 3

/* error codes */
#define AMENV_OK 0
#define AMENV_ERR_INVALID -1   /* bad argument / config                       */
#define AMENV_ERR_HIP -2       /* a HIP runtime call failed (see last_error)   */
#define AMENV_ERR_NO_DEVICE -3 /* no usable gfx950 device                      */
#define AMENV_ERR_ALLOC -4

/* arithmetic type of the dynamics + episode state */
#define AMENV_F32 0 /* product path: fp32 SoA state, fp32 reward                         */
#define AMENV_F64 1 /* logic-check build of the SAME kernel in fp64 (reward/state fp64)  */

/* amenv_config.flags */
#define AMENV_FLAG_AUTO_RESET 1u /* SB3 VecEnv semantics: done envs are reset inside step()      */
#define AMENV_FLAG_NAN_GUARD 2u  /* non-finite state => terminated|NONFINITE (documented deviation) */

/* amenv_config.step_kernel: which implementation of the SAME step amenv_step() launches.  All of them meet the parity
 * gate; LANE and HELPER are bit-identical to each other.  AUTO picks by batch size (latency regime vs throughput regime). */
#define AMENV_KERNEL_AUTO 0
#define AMENV_KERNEL_LANE 1   /* one lane per env, one wavefront per 64-env tile                                      */
#define AMENV_KERNEL_HELPER 2 /* LANE + helper wavefronts per tile (reset RNG words, observation rows, arm link 3)     */
#define AMENV_KERNEL_TEAM 3   /* arm vehicle: a team of 16 lanes (one DPP row) per env, AUTO up to 6144 envs; with AMENV_F64 the fp64 build of
                                 the SAME kernel (logic gate of its DPP plumbing; amenv_step only); rigid vehicles: 4 lanes (one DPP quad)
                                 per env, opt-in only (measured no faster than HELPER)                                       */
#define AMENV_KERNEL_STAGED 4 /* arm vehicle: the four RK4 stages' joint-configuration work on four wavefronts side by side,
                                 the base dynamics on their 36-number aggregates on a fifth (not bit-identical to LANE);
                                 AUTO for 6145..32768 envs; with AMENV_F64 the fp64 build of the same kernel (logic gate) */

/* amenv_task.ee_task (arm vehicles only; ignored without an arm) */
#define AMENV_EE_TASK_BASE 0 /* waypoint distance measured from the base position, as the reference's quadrotor task   */
#define AMENV_EE_TASK_TOOL 1 /* ... from the arm's tool point (forward kinematics): reward, reach test, obs[13:16]     */

/* task variants (which reference env file the step semantics follow) */
#define AMENV_TASK_V2_SCALED20 0 /* v2/rl_env_scaledObs.py  (20-D scaled obs) */
#define AMENV_TASK_V1_SCALED17 1 /* v1/rl_env_scaledObs.py  (17-D scaled obs) */
#define AMENV_TASK_V1_RAW17 2    /* v1/rl_env.py            (17-D raw obs)    */

/* info_bits[i], one uint32 per env per step (the reference's `info` dict + flags,
 * v2/rl_env_scaledObs.py:164,173,179,192,195,196) */
#define AMENV_INFO_TERMINATED 1u
#define AMENV_INFO_TRUNCATED 2u
#define AMENV_INFO_SUCCESS 4u
#define AMENV_INFO_STOPPED 8u
#define AMENV_INFO_CRASHED 16u
#define AMENV_INFO_OOB 32u
#define AMENV_INFO_NONFINITE 64u /* only with AMENV_FLAG_NAN_GUARD */
#define AMENV_INFO_WAS_RESET 128u /* this env was auto-reset at the end of this step */

/* Vehicle: n-rotor rigid body.  Replaces v2/simul_files/model/params.py:10-36.
 * All matrices row-major.  Rotor thrusts T = alloc . [F,Mx,My,Mz]^T, each clamped to
 * [t_min,t_max], then [F',Mx',My',Mz']^T = mix . T   (v2/simul_files/model/quadcopter.py:109-112). */
typedef struct amenv_vehicle {
  int32_t n_rotors; /* 1..AMENV_MAX_ROTORS */
  int32_t n_joints; /* 0 = rigid body; 1..3 = hexacopter + n-link arm (n < 3: the first n joints / links of the arrays below; same kernels,
                       the caller's dimensions: 4 + n actions, 20 + 2 n + 3 observations, 2 n joint state fields; amenv_step only) */
  double mass;      /* kg  (params.py:10) */
  double g;         /* m/s^2 (params.py:11) */
  double inertia[9];     /* body inertia I (params.py:12-14) */
  double inv_inertia[9]; /* I^-1 (params.py:16) */
  double alloc[AMENV_MAX_ROTORS * 4]; /* [n_rotors][4]  (params.py:36 invA) */
  double mix[4 * AMENV_MAX_ROTORS];   /* [4][n_rotors]  (params.py:31-34 A) */
  double t_min[AMENV_MAX_ROTORS];     /* per-rotor thrust floor (params.py:19, minF/4) */
  double t_max[AMENV_MAX_ROTORS];     /* per-rotor thrust cap   (params.py:20, maxF/4) */
  double moment_scale;   /* M = a[1:4]*moment_scale, evaluated in fp32 (rl_env_scaledObs.py:126: 0.1) */
  /* Arm (n_joints = 3): serial chain of revolute joints hanging from the base body; ignored for n_joints = 0.
   * With an arm, `mass` is the TOTAL mass (action scaling F = a0*mass*g), the base body's own mass is
   * mass - sum(link_mass), `inertia` is the base body's inertia about ITS CoM = the body-frame origin.
   * Parameters from Manipulator/src/manipulator_description/sdf/manipulator.sdf via tools/arm_params.py;
   * the joint servo model is this build's (no reference dynamics exist): DESIGN.md "arm". */
  double joint_origin[AMENV_MAX_JOINTS * 3]; /* joint k origin in its parent's frame (joint 1: body frame), :99,159,233 */
  double joint_axis[AMENV_MAX_JOINTS * 3];   /* unit axis (same in parent and child frame), :103,163,237 */
  double link_mass[AMENV_MAX_JOINTS];        /* :131,191,265 (+ closed gripper folded into link 3) */
  double link_com[AMENV_MAX_JOINTS * 3];     /* link CoM in its own frame */
  double link_inertia[AMENV_MAX_JOINTS * 9]; /* about the link CoM, link frame, row-major symmetric */
  double joint_kp, joint_kd;                 /* servo: thdd = clamp(kp*(cmd - th) - kd*thd, +-joint_acc_max) */
  double joint_acc_max;                      /* rad/s^2 (manipulator_moveit/config/joint_limits.yaml: 8) */
  double joint_reserved;
  double joint_limit[AMENV_MAX_JOINTS * 2];  /* [lower, upper] rad: action -1..1 maps onto it, :105-106,165-166,239-240 */
  double tool_offset[3];                     /* tool point in the last link's frame: midpoint of the two gripper-finger joint
                                                origins, manipulator.sdf:371,450 (forward kinematics -> amenv_ee_position, obs, task) */
} amenv_vehicle;

/* Task constants that the reference keeps as literals in rl_env_scaledObs.py. */
typedef struct amenv_task {
  int32_t variant;           /* AMENV_TASK_* */
  int32_t num_waypoints;     /* K; reference hard-codes 1 (rl_env_scaledObs.py:47); arm vehicles: 1 on every kernel, 2..4 on the LANE kernel */
  int32_t max_episode_steps; /* 2000 (rl_env_scaledObs.py:56) */
  int32_t counter_limit;     /* 500  (rl_env_scaledObs.py:59) */
  int32_t rk4_substeps;      /* RK4 sub-steps per control step; 1 */
  int32_t ee_task;           /* AMENV_EE_TASK_*: which point of an arm vehicle the waypoint task measures (default TOOL) */
  double dt;                 /* 1/200 (rl_env_scaledObs.py:30) */
  /* sin(2*pi*k/K), cos(2*pi*k/K), k = 1..K: used by the curved / helical waypoint
   * generators (v2/utils2/utils.py:39,46,53,83-87); filled by amenv_default_config(). */
  double traj_sin[AMENV_MAX_WAYPOINTS];
  double traj_cos[AMENV_MAX_WAYPOINTS];
} amenv_task;

typedef struct amenv_config {
  uint32_t struct_size; /* = sizeof(amenv_config): ABI guard */
  uint32_t abi_version; /* = AMENV_ABI_VERSION */
  int32_t num_envs;     /* N on THIS device */
  int32_t dtype;        /* AMENV_F32 | AMENV_F64 */
  uint32_t flags;       /* AMENV_FLAG_* */
  int32_t block_size;   /* 0 = auto; else threads per workgroup of the LANE kernel: 64, 128 or 256 */
  int32_t step_kernel;  /* AMENV_KERNEL_*: 0 = auto */
  int32_t reserved1;
  uint64_t seed;        /* reset RNG seed (Philox4x32-10 key) */
  int64_t env_id_offset; /* global id of local env 0: RNG is keyed by GLOBAL env id, so
                            results do not depend on how envs are sharded over GPUs */
  amenv_vehicle vehicle;
  amenv_task task;
} amenv_config;

/* Episode state, struct-of-arrays.  Float fields are `dtype` (fp32 or fp64), laid out
 * field-major: fstate[field][env], istate[field][env].  Replaces the attribute set of
 * WaypointQuadEnv (rl_env_scaledObs.py:26-38,53-77) + Quadcopter.state (quadcopter.py:28). */
enum amenv_float_field {
  AMENV_F_PX = 0, AMENV_F_PY, AMENV_F_PZ,          /* position            state[0:3]  */
  AMENV_F_VX, AMENV_F_VY, AMENV_F_VZ,              /* velocity            state[3:6]  */
  AMENV_F_QW, AMENV_F_QX, AMENV_F_QY, AMENV_F_QZ,  /* attitude quaternion state[6:10] */
  AMENV_F_WX, AMENV_F_WY, AMENV_F_WZ,              /* body rates p,q,r    state[10:13]*/
  AMENV_F_FINAL_YAW,                               /* rl_env_scaledObs.py:72          */
  AMENV_F_LAST_DISTANCE,                           /* :76 ; < 0 encodes None          */
  AMENV_F_EP_RETURN,                               /* Monitor-style running return    */
  AMENV_F_WP0                                      /* waypoints: WP0 + 3*k + {0,1,2}  */
  /* with n_joints = 3: joint angles (3) then joint rates (3) follow the waypoints: WP0 + 3*K + {0..5} */
};
enum amenv_int_field {
  AMENV_I_STEP = 0,   /* current_step (:55)                                         */
  AMENV_I_COUNTER,    /* counter (:57)                                              */
  AMENV_I_FLAGS,      /* bits 0-3 waypoint_index, bits 4-7 waypoints of this episode (v1 tasks; 0 = num_waypoints),
                         bit 8 final_waypoint_reached, bit 9 counter_activated */
  AMENV_I_EPISODE,    /* episodes started so far by this env (RNG counter)          */
  AMENV_I_NFIELDS
};
#define AMENV_FLAGBIT_FWR 256
#define AMENV_FLAGBIT_COUNTER_ACTIVE 512

/* Running totals since amenv_create / amenv_stats(reset=1): the Monitor / rollout
 * aggregates, reduced on the GPU (wavefront reduction + one atomic per wave). */
typedef struct amenv_stats {
  uint64_t steps;       /* env-steps executed */
  uint64_t episodes;    /* episodes finished  */
  uint64_t terminated, truncated, success, crashed, oob, nonfinite;
  uint64_t length_sum;  /* sum of finished episode lengths */
  int64_t return_sum_q10; /* sum of finished episode returns, fixed point 2^-10 (order-independent) */
} amenv_stats;

typedef struct amenv amenv; /* opaque */

/* ---- configuration helpers (host only, no device needed) ------------------------- */

/* Fill *cfg with the reference's own constants: the 0.18 kg quadrotor of
 * v2/simul_files/model/params.py and the v2 task literals.  vehicle_name:
 * "quad" (reference, oracle-pinned) | "hexa" | "hexa_arm" (hexacopter_description/
 * and Manipulator/ SDF parameters; no reference dynamics => parity unpinned). */
int amenv_default_config(const char* vehicle_name, int32_t num_envs, amenv_config* cfg);

/* Select which reference env file the task follows and fill that file's literals:
 * AMENV_TASK_V2_SCALED20 (v2/rl_env_scaledObs.py, 20-D obs, 2000 steps) or AMENV_TASK_V1_SCALED17 / _RAW17
 * (v1/rl_env_scaledObs.py / v1/rl_env.py: 17-D obs, 1..2 waypoints per episode, 1200 steps; used by v1/rl_train_vecN.py). */
int amenv_config_set_task(amenv_config* cfg, int32_t variant);

/* obs_dim / act_dim / number of float state fields for a config. */
int amenv_dims(const amenv_config* cfg, int32_t* obs_dim, int32_t* act_dim, int32_t* n_float_fields,
               int32_t* n_int_fields);

/* Algorithmic HBM bytes of one env-step of amenv_step() for this config (DESIGN.md formula). */
int64_t amenv_bytes_per_env_step(const amenv_config* cfg);

const char* amenv_version(void);

/* ---- lifetime --------------------------------------------------------------------- */

/* Replaces WaypointQuadEnv.__init__ (rl_env_scaledObs.py:10-38) x N and
 * make_vec_env(WaypointQuadEnv, n_envs=N) (v2/rl_train.py:24). */
int amenv_create(const amenv_config* cfg, int device, amenv** out);
int amenv_destroy(amenv* env);
/* Text of the last error on this handle (env may be NULL: last create error). */
const char* amenv_last_error(const amenv* env);

/* Re-key the reset RNG: episodes started after this call draw from (seed, global env id, episode).
 * Replaces the role of `np.random.seed(k)` in front of the reference env (its reset(seed=) is a no-op,
 * rl_env_scaledObs.py:41). */
int amenv_set_seed(amenv* env, uint64_t seed);

/* ---- the hot path ----------------------------------------------------------------- */

/* Replaces WaypointQuadEnv.reset (rl_env_scaledObs.py:40-79) for every env whose mask
 * byte is non-zero (mask NULL = all).  obs_out [N,obs_dim] f32 row-major may be NULL;
 * rows of un-reset envs are rewritten with their current observation. */
int amenv_reset(amenv* env, const uint8_t* mask, float* obs_out, void* stream);

/* Replaces, for all N envs in ONE kernel launch:
 *   WaypointQuadEnv.step          rl_env_scaledObs.py:123-196
 *   Quadcopter.update/state_dot   simul_files/model/quadcopter.py:66-114  (RK4 for odeint)
 *   _calculate_reward             rl_env_scaledObs.py:198-231
 *   _get_observation              rl_env_scaledObs.py:98-121
 *   quaternion_to_rpy             utils2/utils.py:4-9
 *   DummyVecEnv.step_wait auto-reset + Monitor episode stats (SB3; v2/rl_train.py:24)
 * actions      [N,act_dim] f32 row-major (already clipped by the caller, as SB3 does)
 * obs          [N,obs_dim] f32   next observation (post-reset for done envs when AUTO_RESET); obs_dim = 20 (v2 task),
 *              17 (v1 tasks), 29 with the arm: the 20 v2 entries + joint angles / pi (3) + joint rates / 5 (3) +
 *              (tool point - base position) / 0.5 in world axes (3)
 * reward       [N] f32 (f64 when dtype = AMENV_F64)
 * done         [N] u8   terminated|truncated
 * info_bits    [N] u32  AMENV_INFO_*
 * terminal_obs [N,obs_dim] f32 or NULL: row i written only when done[i]
 * ep_return    [N] f32 or NULL, ep_len [N] i32 or NULL: written only when done[i] */
int amenv_step(amenv* env, const float* actions, float* obs, void* reward, uint8_t* done,
               uint32_t* info_bits, float* terminal_obs, float* ep_return, int32_t* ep_len, void* stream);

/* amenv_step() plus the device-side duration of that one kernel: the launch carries start/stop
 * events stamped by the kernel's own dispatch (hipExtLaunchKernelGGL), then the call WAITS for the
 * stop event and writes the elapsed microseconds to *kernel_us (host pointer).  For bench/profiling
 * only: it synchronises. */
int amenv_step_timed(amenv* env, const float* actions, float* obs, void* reward, uint8_t* done,
                     uint32_t* info_bits, float* terminal_obs, float* ep_return, int32_t* ep_len, void* stream,
                     float* kernel_us);

/* T consecutive steps in ONE launch with open-loop actions [T,N,act_dim] (action replay /
 * action repeat).  Per-step outputs are [T,N,...] or NULL; state stays in registers between
 * steps.  Same per-step semantics as amenv_step (including auto-reset). */
int amenv_rollout(amenv* env, int32_t n_steps, const float* actions, float* obs, void* reward,
                  uint8_t* done, uint32_t* info_bits, void* stream);

/* ---- state access (parity injection, checkpoint/restore) --------------------------- */

/* Copy the whole SoA episode state to / from caller DEVICE buffers:
 * fstate [n_float_fields][N] of dtype, istate [AMENV_I_NFIELDS][N] int32. */
int amenv_get_state(amenv* env, void* fstate, int32_t* istate, void* stream);
int amenv_set_state(amenv* env, const void* fstate, const int32_t* istate, void* stream);

/* Recompute observations of the current state (no stepping): _get_observation, :98-121. */
int amenv_observe(amenv* env, float* obs_out, void* stream);

/* Forward kinematics of the arm (north_star "arm forward kinematics"): world position of the tool point of every env,
 * ee_out [N,3] f32.  Chain: base pose (p, q) -> joint origins / axes of manipulator.sdf:99,159,233 -> tool_offset (:371,450).
 * Without an arm (n_joints = 0) the tool point is the body origin.  The same quantity feeds obs[26:29] and, with
 * AMENV_EE_TASK_TOOL, the waypoint distance inside amenv_step. */
int amenv_ee_position(amenv* env, float* ee_out, void* stream);

/* Copy the running totals to *host_out (synchronises the stream); optionally zero them. */
int amenv_stats_read(amenv* env, amenv_stats* host_out, int reset, void* stream);

/* Bench / profiling only: copy `bytes` (a multiple of 16) with 4 or 16 bytes per lane -- a dispatch of known traffic in the access
 * pattern of the kernel under test, for calibrating the PMC counters FETCH_SIZE / WRITE_SIZE (tools/pmc_traffic.py). */
int amenv_calibration_copy(const void* src, void* dst, size_t bytes, int32_t bytes_per_lane, void* stream);

/* Name, VGPR count etc. of the step kernel chosen for this handle (for bench/profiles). */
const char* amenv_kernel_name(const amenv* env);

/* ---- observation normaliser: VecNormalize(norm_obs=True, norm_reward=False) (v1/rl_train_vecN.py:10-11) ------------
 * Running mean / variance / count of [n, dim] f32 observation batches on the device (SB3 2.6.0 RunningMeanStd:
 * initial mean 0, var 1, count 1e-4; parallel-moments merge), and obs_n = clip((obs-mean)/sqrt(var+eps), -clip, clip)
 * (SB3 defaults eps = 1e-8, clip = 10).  Third-party semantics, parity unpinned (the reference's vec_normalize.pkl is
 * a pickle and is not read). */
typedef struct amenv_obsnorm amenv_obsnorm;
int amenv_obsnorm_create(int32_t dim, int device, amenv_obsnorm** out);
int amenv_obsnorm_destroy(amenv_obsnorm* h);
/* obs_rms.update(obs): merge the moments of this batch (device pointer, row-major [n, dim]). */
int amenv_obsnorm_update(amenv_obsnorm* h, const float* obs, int64_t n, void* stream);
/* normalize_obs: out may alias in. */
int amenv_obsnorm_apply(amenv_obsnorm* h, const float* in, float* out, int64_t n, float clip, double eps, void* stream);
/* mean[dim], var[dim], count to / from HOST arrays (synchronises): save / load of the statistics. */
int amenv_obsnorm_get(amenv_obsnorm* h, double* mean, double* var, double* count, void* stream);
int amenv_obsnorm_set(amenv_obsnorm* h, const double* mean, const double* var, double count, void* stream);

/* ---- GPU-resident PPO helpers (SURVEY 8 row f3 / BASELINE config 5) ---------------------------------------------------
 * The reference trains with SB3 PPO (v2/rl_train.py:38-56).  The MLP stays in PyTorch-ROCm; these two entry points are
 * the per-env elementwise pieces of SB3's loop, on caller-owned device buffers of the CURRENT device.
 *
 * amenv_gae: RolloutBuffer.compute_returns_and_advantage (SB3 2.6.0) over time-major [n_steps, n_envs] f32 buffers;
 * dones[t, i] != 0 = env i's episode ended at step t (SB3's episode_starts shifted by one; the last row is its `dones`
 * argument); last_values[i] = V(obs after the last step).  gamma = .995, gae_lambda = .9 in the reference (:46-47). */
int amenv_gae(const float* rewards, const float* values, const uint8_t* dones, const float* last_values, float* advantages,
              float* returns, int32_t n_steps, int64_t n_envs, float gamma, float gae_lambda, void* stream);
/* DiagGaussianDistribution.sample + log_prob + the action-space clip of collect_rollouts: raw = mean + exp(log_std) z,
 * clipped = clip(raw, low, high) (bounds v2/rl_env_scaledObs.py:20-24), logp = log N(raw; mean, exp(log_std)) summed over
 * act_dim (4 or 7).  z from Philox4x32-10 keyed by (seed, env_id_offset + i, draw): pass a different `draw` per call. */
int amenv_gaussian_act(const float* mean, const float* log_std, const float* low, const float* high, float* raw, float* clipped,
                       float* logp, int64_t n_envs, int32_t act_dim, uint64_t seed, uint32_t draw, int64_t env_id_offset, void* stream);
/* Forward pass of the reference's policy network (SB3 MlpPolicy, net_arch [128, 64, 64], tanh, separate actor / critic trunks;
 * v2/rl_train.py:27-30) in ONE launch: mean_out [n, act_dim] = action_net(pi trunk(obs)), value_out [n] = value_net(vf trunk(obs));
 * either output may be NULL.  flat_params = the policy's parameters in SB3 state-dict order in one contiguous fp32 buffer
 * (log_std, mlp_extractor.policy_net.{0,2,4}.{weight,bias}, mlp_extractor.value_net.{0,2,4}.{weight,bias}, action_net, value_net:
 * 30,537 floats for 20-D / 4-D).  (obs_dim, act_dim) in {(20,4), (29,7), (17,4)}. */
int amenv_policy_forward(const float* flat_params, int32_t obs_dim, int32_t act_dim, const float* obs, int64_t n, float* mean_out,
                         float* value_out, void* stream);

/* amenv_policy_forward for large batches (the rollout buffer's values / log-probabilities, a 32768-env policy step): the same outputs
 * from the training kernel's arithmetic -- fp32 products as six bf16 MFMAs on exactly split operands (see amenv_ppo_mlp_step) -- in two
 * launches (weight packing, forward).  workspace: amenv_ppo_mlp_workspace_bytes() bytes, 16-byte aligned (the same area may serve
 * amenv_ppo_mlp_step: both rewrite its weight part from flat_params on every call). */
int amenv_policy_forward_mfma(const float* flat_params, int32_t obs_dim, int32_t act_dim, const float* obs, int64_t n, float* mean_out,
                              float* value_out, void* workspace, void* stream);

/* Closed-loop rollout in ONE launch (SB3 collect_rollouts, v2/rl_train.py:38-56, for n_steps steps): per step
 *   obs_t -> actor / critic MLPs ([128, 64, 64] tanh; bf16 matrix cores, fp32 accumulate) -> a_t = mean + exp(log_std) z  (Philox keyed
 *   by (seed, global env id, draw0 + t) as amenv_gaussian_act) -> clip to the action box -> env step (as amenv_step, auto-reset included).
 * State, per-lane constants and the policy weights stay in registers between steps.  Built for fp32 vehicles on the single-waypoint v2 task:
 * the rigid vehicles with 4 or 6 rotors -- the reference's quadrotor (obs_dim 20, act_dim 4; default workgroup size; the env part is the
 * lane-quad step, 4 lanes per env: replaying the recorded clipped actions through amenv_step on a handle created with AMENV_KERNEL_TEAM
 * reproduces every row bit for bit; the first layer runs on two-part bf16 inputs and weights, which holds the action within 3e-2 of the
 * fp32 policy's on the reference checkpoint) -- and the 6-rotor vehicle with the z,x,x 3-joint arm (obs_dim 29, act_dim 7); other
 * configurations return AMENV_ERR_INVALID.  For the arm vehicle the env part is the arithmetic of the kernel amenv_step
 * runs for this env (16 lanes per env where that is the lane-team kernel, else one lane per env with the arithmetic of the LANE / HELPER step
 * kernels: replaying the recorded clipped actions through amenv_step on such a handle reproduces every row bit for bit; a handle whose
 * amenv_step runs the STAGED kernel agrees to rounding); both forms draw the same noise.  An opt-in ROLLOUT mode: bf16
 * rounding perturbs the action means by ~1e-2 of their scale; log-probs are those of the samples under the means actually used.
 *   flat_params  fp32 policy parameters in SB3 state-dict order (see amenv_policy_forward)
 *   obs          [n_steps + 1, N, obs_dim] f32: row 0 <- observation at entry, row t + 1 <- after step t (post-reset for done envs)
 *   actions      [n_steps, N, act_dim] f32 raw (unclipped) samples;  logp, values, rewards [n_steps, N] f32;  dones [n_steps, N] u8
 *   info_bits    [n_steps, N] u32 or NULL;  terminal_obs [n_steps, N, obs_dim] f32 or NULL (rows written only where dones != 0) */
int amenv_rollout_policy(amenv* env, int32_t n_steps, const float* flat_params, uint64_t seed, uint32_t draw0, float* obs, float* actions,
                         float* logp, float* values, float* rewards, uint8_t* dones, uint32_t* info_bits, float* terminal_obs, void* stream);

/* The part of SB3's PPO.train between the network outputs and the backward pass, fused (three launches instead of ~60 torch
 * kernels): per-minibatch advantage normalisation (mean, unbiased std, eps 1e-8), Gaussian log-prob of `actions` under
 * (mean, log_std), ratio to old_logp, clipped surrogate, value MSE, entropy bonus -- and the gradient of
 *   L = policy_loss + ent_coef * entropy_loss + vf_coef * value_loss
 * with respect to mean [n, act_dim], value [n] and log_std [act_dim] (act_dim 4 or 7).  stats4 = {policy_loss, value_loss,
 * entropy_loss, clip_fraction}.  Deterministic (fixed-order reductions, no atomics).  workspace: device buffer of
 * amenv_ppo_workspace_bytes() bytes, 8-byte aligned.  Hyper-parameters of the reference: clip .2, ent 5e-4, vf .5 (v2/rl_train.py:38-53). */
size_t amenv_ppo_workspace_bytes(void);
int amenv_ppo_loss_grad(const float* mean, const float* value, const float* log_std, const float* actions, const float* old_logp,
                        const float* advantages, const float* returns, int64_t n, int32_t act_dim, float clip_range, float ent_coef,
                        float vf_coef, int32_t normalize_advantage, float* d_mean, float* d_value, float* d_log_std, float* stats4,
                        void* workspace, void* stream);

/* One whole PPO minibatch step of the reference's policy ([128, 64, 64] tanh actor + critic, v2/rl_train.py:27-30,38-53) in ONE fused
 * kernel (+ a launch in front that packs the weights as MFMA operands and a fixed-order reduction behind): forward of both MLPs, SB3's
 * loss as amenv_ppo_loss_grad computes it, backward through both MLPs, all weight / bias gradients.  fp32 in and out on the matrix
 * cores: every operand is split exactly into three bf16 parts (each rounded to nearest) and a product is the sum of the six partial products above 2^-16
 * (v_mfma_f32_32x32x16_bf16, fp32 accumulation; what is dropped is below one fp32 rounding), so the result is as close to the same loss in
 * fp64 as autograd on the fp32 torch modules is (tests/test_gpu_ppo.py).  flat_params / flat_grad: the policy's parameters / their gradients in SB3 state-dict order (see
 * amenv_policy_forward); obs [n, obs_dim], actions [n, act_dim], old_logp / advantages / returns [n] f32; stats4 as amenv_ppo_loss_grad.
 * (obs_dim, act_dim) in {(20,4), (29,7), (17,4)}.  index: NULL, or n row numbers -- minibatch sample s is row index[s] of obs / actions /
 * old_logp / advantages / returns (SB3 draws its minibatches as slices of a permutation of the rollout buffer, RolloutBuffer.get: the
 * gather happens inside the kernel, the rollout tensors stay where they are).  Deterministic: no atomics on the gradient path.
 * workspace: amenv_ppo_mlp_workspace_bytes() bytes, 16-byte aligned. */
size_t amenv_ppo_mlp_workspace_bytes(void);
int amenv_ppo_mlp_step(const float* flat_params, int32_t obs_dim, int32_t act_dim, const float* obs, const float* actions, const float* old_logp,
                       const float* advantages, const float* returns, const int64_t* index, int64_t n, float clip_range, float ent_coef,
                       float vf_coef, int32_t normalize_advantage, float* flat_grad, float* stats4, void* workspace, void* stream);

/* Gradient-norm clip + Adam on the flat parameter buffer (the torch.nn.utils.clip_grad_norm_ + torch.optim.Adam step of SB3's
 * PPO.train(); Adam with eps 1e-5, v2/rl_train.py:38 through SB3's defaults).  exp_avg / exp_avg_sq / step: torch.optim.Adam's state for
 * that parameter (step: ONE f32 on the device, as torch keeps it with capturable=True; incremented here).  hyper6 (device): lr, beta1,
 * beta2, eps, max_grad_norm (<= 0: no clipping), grad_scale (multiplies the gradient first: 1 / world size after a sum all-reduce).
 * flat_grad (16-byte aligned) is READ-ONLY (the clipped gradient is applied, not stored: every workgroup of the launch takes the norm of the whole buffer
 * as it was on entry), grad_norm_out (may be NULL) receives the norm before clipping.  ticket: one zero-initialised device word the
 * kernel uses and leaves zero. */
int amenv_ppo_adam_step(float* flat_params, const float* flat_grad, float* exp_avg, float* exp_avg_sq, float* step, int64_t n, const float* hyper6,
                        float* grad_norm_out, uint32_t* ticket, void* stream);

/* Parity gate of the arm vehicle's arithmetic (no reference dynamics exist for it; DESIGN.md "arm"): the 19 state derivatives of n states
 * [n, 19] = (p, v, q, w, th, thd) under post-mixer wrenches [n, 4] = (F, Mx, My, Mz) and joint commands [n, 3], deriv [n, 19], all of `dtype`
 * (AMENV_F32 | AMENV_F64), on the current device.  form 0: the per-link Newton-Euler sums the lane and two-wave kernels run; form 1: the staged
 * form (joint-configuration aggregates, then the base dynamics on them) the stage-wave and lane-team kernels run.  Tests compare both fp64
 * instantiations with the oracle's right-hand side (<= 1e-12).  cfg: an arm vehicle with the z,x,x joint axes. */
int amenv_arm_rhs(const amenv_config* cfg, int32_t form, int32_t dtype, const void* state19, const void* wrench4, const void* cmd3, void* deriv19,
                  int64_t n, void* stream);

/* ---- PID + minimum-snap baseline controller ------------------------------------------------------------
 * The reference's hand-tuned controller, `v2/PID Controller/{pid_controller,trajGen3D,runsim}.py`, for N vehicles per launch on caller-owned
 * device buffers of the CURRENT device.  `dtype` selects the arithmetic of the float buffers marked (dtype): AMENV_F64 = the logic gate
 * pinned to the reference's recorded run (tests/golden/pid_helix.npz), AMENV_F32 = the product build. */
typedef struct amenv_pid_params {
  double dt;           /* control period the integrals are advanced by (runsim.py:28: 0.01; the waypoint env: 1/200) */
  double mass, g;      /* params.py:10-11 */
  double max_integral; /* pid_controller.py:34: 100 */
  double gain[18];     /* [x, y, z, phi, theta, psi][k_p, k_d, k_i]  (pid_controller.py:16-21) */
} amenv_pid_params;
/* The committed gains and the reference quadrotor's mass / g; dt = 0.01. */
int amenv_pid_default_params(amenv_pid_params* p);

/* pid_controller.run (pid_controller.py:37-115) with Quadcopter.attitude() (model/quadcopter.py:57-59):
 *   state    [n, 13] (dtype) position, velocity, quaternion (w, x, y, z), body rates = Quadcopter.state
 *   des      [n, 11] (dtype) desired position, velocity, acceleration, yaw, yaw rate  (trajGen3D.DesiredState)
 *   integral [n, 6]  (dtype) in/out: the module's integral memory (x, y, z, phi, theta, psi), clamped to +-max_integral
 *   F_out [n], M_out [n, 3] (dtype): thrust and moments BEFORE the mixer;  rpy_out [n, 3] (dtype) or NULL: the attitude it used */
int amenv_pid_run(const amenv_pid_params* p, int32_t dtype, const void* state, const void* des, void* integral, void* F_out, void* M_out,
                  void* rpy_out, int64_t n, void* stream);

/* trajGen3D.get_MST_coefficients / MST (:211-292) for n_traj trajectories of n_segments (1..16) 7th-order segments each:
 *   waypoints [n_traj, n_segments + 1, 3] f64  ->  coeff [n_traj, 8 n_segments, 3] f64 (the reference's coeff_x / _y / _z as columns),
 *   seg_time [n_traj, n_segments] = |w_i - w_i+1| / speed and seg_start [n_traj, n_segments + 1] = their running sum (:97-103).
 * The 8n x 8n constraint matrix depends on n_segments only: it is inverted on the device (fp64 Gauss-Jordan, partial pivoting) into
 * `workspace` (amenv_minsnap_workspace_bytes(n_segments) bytes, 8-byte aligned), then every trajectory is one small matrix product. */
size_t amenv_minsnap_workspace_bytes(int32_t n_segments);
int amenv_minsnap_solve(int32_t n_segments, int64_t n_traj, double speed, const double* waypoints, double* coeff, double* seg_time,
                        double* seg_start, void* workspace, void* stream);
/* trajGen3D.generate_trajectory (:76-187) for n_query (trajectory, time) pairs: query i evaluates trajectory traj[i] (traj NULL: i)
 * at t[i] (f64) -> des [n_query, 11] (dtype): position, velocity, acceleration, yaw = yaw rate = 0 (:183-184); t == 0 returns the first
 * waypoint at rest (:108-111), t past the last segment the last waypoint at rest (:121,176-179). */
int amenv_minsnap_eval(int32_t n_segments, int64_t n_query, const double* coeff, const double* seg_time, const double* seg_start,
                       const double* waypoints, const int64_t* traj, const double* t, int32_t dtype, void* des, void* stream);

/* The controller wired to the waypoint environment as a closed-loop action source (the way runsim.py:26-31 flies its waypoint list):
 * per episode a ONE-segment rest-to-rest minimum-snap trajectory from where the episode started to the waypoint at `speed` m/s, tracked
 * by the PID; thrust scaled by mass, moments by inertia_ratio (this vehicle's inertia / the reference quadrotor's, per axis) and the
 * moment vector scaled into the action box.  One launch: observation rows in, action rows out.
 *   obs     [n, obs_dim] f32, v2 layout (rl_env_scaledObs.py:98-121), obs_dim >= 20 (29 with the arm)
 *   done    [n] u8 or NULL: envs whose episode ended on the previous step (auto-reset => a new trajectory starts)
 *   pstate  [n, 14] (dtype) in/out: t, start (3), goal (3), integrals (6), fresh; initialise to {0 x 13, 1}
 *   actions [n, act_dim] f32, act_dim >= 4: thrust / (m g) in [0, 2], moments in [-1, 1]; entries 4.. (arm joints) = 0 (home)
 * tool_mode = 1 (arm vehicle with AMENV_EE_TASK_TOOL, obs_dim 29): the position loop tracks the tool point instead of the base. */
typedef struct amenv_pid_policy_params {
  amenv_pid_params pid;
  double speed;            /* m/s along the segment (runsim.py:27 flies 1.2) */
  double moment_scale;     /* amenv_vehicle.moment_scale */
  double inertia_ratio[3];
  int32_t obs_dim, act_dim, tool_mode, reserved;
} amenv_pid_policy_params;
int amenv_pid_policy(const amenv_pid_policy_params* p, int32_t dtype, const float* obs, const uint8_t* done, void* pstate, float* actions,
                     int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMENV_H_ */
