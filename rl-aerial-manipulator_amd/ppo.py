"""GPU-resident PPO over `GpuWaypointEnv` (BASELINE config 5).

Mirrors what the reference's training entry does through Stable-Baselines3 (v2/rl_train.py:22-56):
`PPO("MlpPolicy", env, learning_rate=2e-4, n_steps=2048, batch_size=128, n_epochs=12, gamma=.995, gae_lambda=.9,
clip_range=.2, ent_coef=5e-4, policy_kwargs=dict(net_arch=[128,64,64], activation_fn=nn.Tanh))` and SB3's defaults
(vf_coef .5, max_grad_norm .5, normalize_advantage, Adam eps 1e-5, orthogonal init; values read from the `data` JSON of
`checkpoints_from_8_6M/ppo_model_2300000_steps.zip`).  SB3 2.6.0 is third-party and absent here: its published algorithm is
restated, **parity unpinned** except for the policy network itself, whose tensors load from the reference's `policy.pth`
(tests: the reference's best checkpoint flies the waypoint task on this env).

What runs where:
  * env step, GAE (`amenv_gae`), action sampling + log-prob + clip (`amenv_gaussian_act`): HIP kernels behind the C ABI;
  * the two 20->128->64->64 tanh MLPs, their backward and Adam: PyTorch-ROCm (rocBLAS GEMMs) -- plumbing, as SURVEY §1 says;
  * the whole rollout stays in HBM: the env writes obs / reward / done straight into rows of the rollout buffer;
  * multi-GPU: one process per GPU, envs sharded by global id, ONE all-reduce per minibatch of one flat fp32 gradient buffer
    (30,537 parameters = 122 KB for the 20-D / 4-D task) over RCCL; nothing on the step path.
"""
import ctypes as C
import io
import json
import math
import os
import zipfile

import torch
from torch import nn

from . import _lib as L


_SPLIT = int(os.environ.get("AMENV_PPO_SPLIT", "1024"))   # rows per partial product of the weight gradient (measured: 128..2048 within 5 %, 1024 best)


class _TallSkinnyLinearFn(torch.autograd.Function):
    """y = x W^T + b with a weight gradient shaped for this workload.  dW = dY^T X is a [out, in] <= 128 x 128 result
    reduced over the whole minibatch (65,536 rows): the library GEMM picked for that shape runs in one or two workgroups
    without split-K (rocprof, profiles/r01/ppo_update_kernel_stats_before.csv: ~200 us per call, 55 % of the update).
    Here the batch is cut into 1024-row slabs (256 for small batches), one strided-batched GEMM forms the per-slab products on all CUs and a sum
    over slabs finishes the reduction."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.addmm(b, x, w.t())

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ w if ctx.needs_input_grad[0] else None
        rows = x.shape[0]
        split = _SPLIT if rows >= 2 * _SPLIT else 256
        c = rows // split
        m = c * split
        gw = torch.bmm(gy[:m].view(c, split, -1).transpose(1, 2), x[:m].view(c, split, -1)).sum(0)
        if m < rows:
            gw = gw + gy[m:].t() @ x[m:]
        return gx, gw, gy.sum(0)


class _Linear(nn.Linear):
    """nn.Linear (same parameters and state-dict keys) that switches to the tall-skinny weight gradient for big batches."""

    def forward(self, x):
        if x.dim() == 2 and x.shape[0] >= 512 and torch.is_grad_enabled() and x.is_contiguous():
            return _TallSkinnyLinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)


def _mlp(sizes):
    layers = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        layers += [_Linear(a, b), nn.Tanh()]
    return nn.Sequential(*layers)


class _MlpExtractor(nn.Module):
    """Separate actor / critic trunks; attribute names give SB3's state-dict keys `mlp_extractor.policy_net.{0,2,4}.*`."""

    def __init__(self, obs_dim, net_arch):
        super().__init__()
        self.policy_net = _mlp([obs_dim, *net_arch])
        self.value_net = _mlp([obs_dim, *net_arch])


class ActorCritic(nn.Module):
    """SB3 `ActorCriticPolicy` for a Box action space: tanh MLP [128,64,64] x2, linear heads, state-independent log_std
    (v2/rl_train.py:27-30).  `state_dict()` has exactly the keys and shapes of the reference checkpoints' `policy.pth`."""

    def __init__(self, obs_dim=20, act_dim=4, net_arch=(128, 64, 64), log_std_init=0.0, ortho_init=True,
                 action_low=None, action_high=None):
        super().__init__()
        self.obs_dim, self.act_dim, self.net_arch = int(obs_dim), int(act_dim), tuple(net_arch)
        self.log_std = nn.Parameter(torch.full((self.act_dim,), float(log_std_init)))
        self.mlp_extractor = _MlpExtractor(self.obs_dim, self.net_arch)
        self.action_net = _Linear(self.net_arch[-1], self.act_dim)
        self.value_net = _Linear(self.net_arch[-1], 1)
        # action space of WaypointQuadEnv (v2/rl_env_scaledObs.py:20-24); joint commands of the arm are in [-1, 1] too
        lo = [0.0] + [-1.0] * (self.act_dim - 1) if action_low is None else action_low
        hi = [2.0] + [1.0] * (self.act_dim - 1) if action_high is None else action_high
        self.register_buffer("action_low", torch.as_tensor(lo, dtype=torch.float32), persistent=False)
        self.register_buffer("action_high", torch.as_tensor(hi, dtype=torch.float32), persistent=False)
        if ortho_init:  # SB3: gain sqrt(2) for the trunks, 0.01 for the action head, 1 for the value head
            for mod, gain in ((self.mlp_extractor, math.sqrt(2.0)), (self.action_net, 0.01), (self.value_net, 1.0)):
                for m in mod.modules():
                    if isinstance(m, nn.Linear):
                        nn.init.orthogonal_(m.weight, gain=gain)
                        nn.init.zeros_(m.bias)
        self.flat_param = self.flat_grad = None

    # ---- forward pieces -----------------------------------------------------------------------
    _FUSED_DIMS = {(20, 4), (29, 7), (17, 4)}
    MFMA_FORWARD_ROWS = 8192   # batches from here on take amenv_policy_forward_mfma (below: the VALU kernel's shorter latency wins)

    def fused_ok(self, obs):
        """The one-launch HIP forward (`amenv_policy_forward`) applies: inference on the GPU, parameters in the flat buffer,
        the reference's architecture.  Training passes (autograd) go through the torch modules."""
        return (self.flat_param is not None and self.flat_param.data_ptr() == self.log_std.data_ptr()   # still the parameters' home
                and obs.is_cuda and obs.device == self.flat_param.device and not torch.is_grad_enabled() and self.net_arch == (128, 64, 64)
                and (self.obs_dim, self.act_dim) in self._FUSED_DIMS and obs.dtype == torch.float32 and obs.dim() == 2)

    def forward_fused(self, obs, want_mean=True, want_value=True):
        """mean [n, A] and / or value [n] of a batch of observations in ONE kernel launch (weights as scalar operands of the FMAs,
        activations through LDS; csrc/amenv_policy.hpp) instead of 14 library kernels; large batches on the matrix cores (two launches)."""
        obs = obs.contiguous()
        n = obs.shape[0]
        mean = torch.empty(n, self.act_dim, dtype=torch.float32, device=obs.device) if want_mean else None
        value = torch.empty(n, dtype=torch.float32, device=obs.device) if want_value else None
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        if n >= self.MFMA_FORWARD_ROWS:   # large batches: the training kernel's forward passes (matrix cores, csrc/amenv_mlp_train.hpp)
            ws = self.__dict__.get("_fwd_ws")
            if ws is None or ws.device != obs.device:
                ws = self.__dict__["_fwd_ws"] = torch.empty(L.load().amenv_ppo_mlp_workspace_bytes() // 8 + 2, dtype=torch.float64, device=obs.device)
            rc = L.load().amenv_policy_forward_mfma(p(self.flat_param.detach()), self.obs_dim, self.act_dim, p(obs), n, p(mean), p(value), p(ws), stream)
        else:
            rc = L.load().amenv_policy_forward(p(self.flat_param.detach()), self.obs_dim, self.act_dim, p(obs), n, p(mean), p(value), stream)
        if rc != 0:
            raise L.AmenvError(f"amenv_policy_forward failed ({rc})")
        return mean, value

    def actor(self, obs):
        if self.fused_ok(obs):
            return self.forward_fused(obs, True, False)[0]
        return self.action_net(self.mlp_extractor.policy_net(obs))

    def critic(self, obs):
        if self.fused_ok(obs):
            return self.forward_fused(obs, False, True)[1]
        return self.value_net(self.mlp_extractor.value_net(obs)).squeeze(-1)

    def actor_critic(self, obs):
        """mean, value -- one launch on the GPU inference path."""
        if self.fused_ok(obs):
            return self.forward_fused(obs, True, True)
        return self.actor(obs), self.critic(obs)

    def evaluate_actions(self, obs, actions):
        """values, log pi(a|s), entropy -- SB3 ActorCriticPolicy.evaluate_actions for DiagGaussianDistribution."""
        mean = self.actor(obs)
        z = (actions - mean) * torch.exp(-self.log_std)
        logp = (-0.5 * z * z - self.log_std - 0.5 * math.log(2.0 * math.pi)).sum(-1)
        entropy = (0.5 + 0.5 * math.log(2.0 * math.pi) + self.log_std).sum().expand(obs.shape[0])
        return self.critic(obs), logp, entropy

    @torch.no_grad()
    def predict(self, obs, deterministic=True, generator=None):
        """Action for the env: mean (or a sample) clipped to the action space, as SB3's `policy.predict`."""
        if obs.is_cuda and (self.flat_param is None or self.flat_param.data_ptr() != self.log_std.data_ptr()):
            self.flatten_()          # inference on the GPU goes through the one-launch forward, which reads the flat buffer
        mean = self.actor(obs)
        if not deterministic:
            mean = mean + torch.exp(self.log_std) * torch.randn(mean.shape, device=mean.device, generator=generator)
        return torch.minimum(torch.maximum(mean, self.action_low), self.action_high)

    # ---- one flat parameter / gradient buffer ---------------------------------------------------
    def flatten_(self):
        """Re-home every parameter (and its gradient) as a view into ONE flat fp32 buffer: the optimiser updates one
        tensor and the multi-GPU gradient exchange is one all-reduce of `flat_grad`."""
        ps = list(self.parameters())
        n = sum(p.numel() for p in ps)
        flat = torch.empty(n, dtype=ps[0].dtype, device=ps[0].device)
        grad = torch.zeros_like(flat)
        off = 0
        for p in ps:
            k = p.numel()
            flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = flat[off:off + k].view_as(p)
            p.grad = grad[off:off + k].view_as(p)
            off += k
        self.flat_param, self.flat_grad = flat, grad
        if flat.is_cuda and (self.obs_dim, self.act_dim) in self._FUSED_DIMS and self.net_arch == (128, 64, 64):
            # workspace of the matrix-core forward (forward_fused, large batches): allocated here, never inside a graph capture
            self.__dict__["_fwd_ws"] = torch.empty(L.load().amenv_ppo_mlp_workspace_bytes() // 8 + 2, dtype=torch.float64, device=flat.device)
        return self

    def num_parameters(self):
        return sum(p.numel() for p in self.parameters())

    # ---- SB3 checkpoint tensors -------------------------------------------------------------------
    @staticmethod
    def read_sb3_state_dict(path):
        """Tensors of an SB3 checkpoint: a `.zip` as `PPO.save` writes (v2/rl_train.py:57, `policy.pth` inside) or a bare
        `policy.pth`.  Loaded with `weights_only=True`: nothing from the file is executed."""
        if zipfile.is_zipfile(path):
            with zipfile.ZipFile(path) as z:
                names = z.namelist()
                if "policy.pth" in names:
                    return torch.load(io.BytesIO(z.read("policy.pth")), weights_only=True, map_location="cpu")
        return torch.load(path, weights_only=True, map_location="cpu")

    @classmethod
    def from_sb3(cls, path_or_state_dict, device="cpu"):
        sd = path_or_state_dict if isinstance(path_or_state_dict, dict) else cls.read_sb3_state_dict(path_or_state_dict)
        w0 = sd["mlp_extractor.policy_net.0.weight"]
        arch = tuple(int(sd[f"mlp_extractor.policy_net.{i}.weight"].shape[0]) for i in range(0, 64, 2)
                     if f"mlp_extractor.policy_net.{i}.weight" in sd)
        pol = cls(obs_dim=int(w0.shape[1]), act_dim=int(sd["action_net.weight"].shape[0]), net_arch=arch, ortho_init=False)
        pol.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
        return pol.to(device)

    def save_sb3_policy(self, path):
        """Write `policy.pth` with SB3's key names: `model.policy.load_state_dict(torch.load(path))` accepts it."""
        torch.save({k: v.detach().cpu().clone() for k, v in self.state_dict().items()}, path)


class RolloutBuffer:
    """Time-major rollout storage in HBM: obs [T+1, N, D] (row T = the observation after the last step), raw actions
    [T, N, A], log-probs / values / rewards / advantages / returns [T, N] f32, dones [T, N] u8 (episode ended AT step t)."""

    def __init__(self, n_steps, n_envs, obs_dim, act_dim, device):
        T, N = int(n_steps), int(n_envs)
        f = dict(dtype=torch.float32, device=device)
        self.n_steps, self.n_envs = T, N
        self.obs = torch.zeros(T + 1, N, obs_dim, **f)
        self.actions = torch.zeros(T, N, act_dim, **f)
        self.logp = torch.zeros(T, N, **f)
        self.values = torch.zeros(T, N, **f)
        self.rewards = torch.zeros(T, N, **f)
        self.dones = torch.zeros(T, N, dtype=torch.uint8, device=device)
        self.last_values = torch.zeros(N, **f)
        self.advantages = torch.zeros(T, N, **f)
        self.returns = torch.zeros(T, N, **f)

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in vars(self).values() if torch.is_tensor(t))


def compute_gae(buffer, gamma, gae_lambda):
    """advantages / returns of a rollout buffer through the HIP kernel (`amenv_gae`); device tensors only."""
    b = buffer
    if not b.rewards.is_cuda:
        raise L.AmenvError("compute_gae runs on the GPU only (amenv_gae); there is no CPU fallback")
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    rc = L.load().amenv_gae(p(b.rewards), p(b.values), p(b.dones), p(b.last_values), p(b.advantages), p(b.returns),
                            b.n_steps, b.n_envs, float(gamma), float(gae_lambda),
                            C.c_void_p(torch.cuda.current_stream(b.rewards.device).cuda_stream))
    if rc != 0:
        raise L.AmenvError(f"amenv_gae failed ({rc})")
    return b.advantages, b.returns


def gaussian_act(mean, log_std, low, high, raw_out, clipped_out, logp_out, seed, draw, env_id_offset=0):
    """Sample + log-prob + clip in one launch (`amenv_gaussian_act`); noise keyed by (seed, global env id, draw)."""
    if not mean.is_cuda:
        raise L.AmenvError("gaussian_act runs on the GPU only (amenv_gaussian_act); there is no CPU fallback")
    n, a = mean.shape
    for t in (mean, raw_out, clipped_out, logp_out, log_std, low, high):
        if not (t.is_contiguous() and t.dtype == torch.float32 and t.device == mean.device):
            raise L.AmenvError("gaussian_act: contiguous float32 tensors on one device required")
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    rc = L.load().amenv_gaussian_act(p(mean), p(log_std), p(low), p(high), p(raw_out), p(clipped_out), p(logp_out), n, a,
                                     int(seed) & 0xFFFFFFFFFFFFFFFF, int(draw) & 0xFFFFFFFF, int(env_id_offset),
                                     C.c_void_p(torch.cuda.current_stream(mean.device).cuda_stream))
    if rc != 0:
        raise L.AmenvError(f"amenv_gaussian_act failed ({rc}): act_dim must be 4 or 7")


class MinibatchStep:
    """One PPO minibatch update (SB3 `PPO.train` inner loop body): loss, backward, gradient exchange, clip, Adam.

    On the GPU the step is launch-bound in eager mode (~200 small kernels per minibatch; rocprof: 245 ms of kernels in a
    359 ms update), so after three eager calls it is captured into HIP graphs on static minibatch buffers and replayed:
    one graph for a single GPU; for several GPUs two graphs (forward/backward | clip + Adam) around the ONE eager RCCL
    all-reduce of the flat gradient buffer.  A ragged last minibatch (n % batch_size) always runs eagerly."""

    def __init__(self, policy, optimizer, *, clip_range=0.2, ent_coef=5e-4, vf_coef=0.5, max_grad_norm=0.5,
                 normalize_advantage=True, dist=None, use_graph=None, split_graphs=None, fused_loss=None, fused_mlp=None):
        if policy.flat_grad is None:
            policy.flatten_()
        self.policy, self.optimizer, self.dist = policy, optimizer, dist
        self.world = dist.get_world_size() if dist is not None else 1
        # two graphs around an eager all-reduce whenever there is a process group to talk to (tests force it with 1 rank)
        self.split = (self.world > 1) if split_graphs is None else bool(split_graphs)
        self.clip_range, self.ent_coef, self.vf_coef = float(clip_range), float(ent_coef), float(vf_coef)
        self.max_grad_norm, self.normalize_advantage = max_grad_norm, bool(normalize_advantage)
        dev = policy.flat_param.device
        capturable = all(g.get("capturable", False) for g in optimizer.param_groups) if hasattr(optimizer, "param_groups") else False
        self.use_graph = (dev.type == "cuda" and capturable) if use_graph is None else bool(use_graph)
        self.stats = torch.zeros(5, device=dev)          # policy loss, value loss, entropy loss, clip fraction, grad norm
        self._params = list(policy.parameters())         # in flat-buffer order (flatten_ walks parameters() the same way)
        assert self._params[0] is policy.log_std
        self._net_params = self._params[1:]
        # loss + its gradient in the HIP kernels (amenv_ppo_loss_grad) on the GPU; the torch expression is the host-side statement
        self.fused_loss = (dev.type == "cuda") if fused_loss is None else bool(fused_loss)
        self._fused_buf = None
        # the whole forward / loss / backward of both MLPs in ONE kernel on the fp32 matrix cores (amenv_ppo_mlp_step) where the policy
        # has the reference's architecture; the torch modules + fused loss otherwise
        ok = dev.type == "cuda" and getattr(policy, "net_arch", None) == (128, 64, 64) and (policy.obs_dim, policy.act_dim) in policy._FUSED_DIMS
        self.fused_mlp = ok if fused_mlp is None else (bool(fused_mlp) and ok)
        self._mlp_ws = None
        # clip + Adam in one launch on torch.optim.Adam's own state tensors (amenv_ppo_adam_step) where the optimiser is the plain Adam
        # on the flat buffer that `PPO` builds; anything else steps through torch
        self.fused_adam = self.fused_mlp and self._adam_is_plain(optimizer, policy)
        self._adam_hyper = self._adam_key = self._adam_ticket = None
        if self.fused_mlp and use_graph is None:
            self.use_graph = False                       # five launches per minibatch: nothing left for a graph to save
        self._static = None
        self._graphs = None
        self._eager_calls = 0

    # -- the arithmetic (shared by the eager and the captured path) --------------------------------
    def _forward_backward(self, obs, actions, old_logp, adv, ret):
        if self.fused_mlp:
            return self._forward_backward_mlp(obs, actions, old_logp, adv, ret)
        if self.fused_loss:
            return self._forward_backward_fused(obs, actions, old_logp, adv, ret)
        if self.normalize_advantage and adv.numel() > 1:
            adv = (adv - adv.mean()) / (adv.std() + 1e-8)
        values, logp, entropy = self.policy.evaluate_actions(obs, actions)
        ratio = torch.exp(logp - old_logp)
        c = self.clip_range
        pl = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1.0 - c, 1.0 + c)).mean()
        vl = torch.nn.functional.mse_loss(ret, values)
        el = -entropy.mean()
        # gradients straight into the flat buffer with ONE concatenation (autograd's per-parameter accumulate-into-.grad would
        # be 17 tiny read-modify-write kernels plus a zero fill per minibatch)
        grads = torch.autograd.grad(pl + self.ent_coef * el + self.vf_coef * vl, self._params)
        torch.cat([g.reshape(-1) for g in grads], out=self.policy.flat_grad)
        with torch.no_grad():
            self.stats[0], self.stats[1], self.stats[2] = pl.detach(), vl.detach(), el.detach()
            self.stats[3] = ((ratio.detach() - 1.0).abs() > c).float().mean()

    def _forward_backward_fused(self, obs, actions, old_logp, adv, ret):
        """Same loss, with everything between the network outputs and the backward pass in the HIP kernels of
        `amenv_ppo_loss_grad` (3 launches for ~60): autograd only walks the two MLPs, seeded with d L / d mean and d L / d value."""
        pol = self.policy
        mean, values = pol.actor(obs), pol.critic(obs)
        n, a = mean.shape
        if self._fused_buf is None or self._fused_buf[0].shape[0] != n:
            f = dict(dtype=torch.float32, device=mean.device)
            self._fused_buf = (torch.empty(n, a, **f), torch.empty(n, **f), torch.empty(a, **f),
                               torch.empty(L.load().amenv_ppo_workspace_bytes() // 8, dtype=torch.float64, device=mean.device))
        d_mean, d_value, d_log_std, ws = self._fused_buf
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        m, v = mean.detach().contiguous(), values.detach().contiguous()
        rc = L.load().amenv_ppo_loss_grad(p(m), p(v), p(pol.log_std.detach()), p(actions), p(old_logp), p(adv), p(ret), n, a, self.clip_range,
                                          self.ent_coef, self.vf_coef, 1 if self.normalize_advantage else 0, p(d_mean), p(d_value), p(d_log_std),
                                          p(self.stats), p(ws), C.c_void_p(torch.cuda.current_stream(mean.device).cuda_stream))
        if rc != 0:
            raise L.AmenvError(f"amenv_ppo_loss_grad failed ({rc})")
        grads = torch.autograd.grad([mean, values], self._net_params, grad_outputs=[d_mean, d_value])
        torch.cat([d_log_std] + [g.reshape(-1) for g in grads], out=pol.flat_grad)     # log_std is the first parameter

    @staticmethod
    def _adam_is_plain(opt, policy):
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1 or len(opt.param_groups[0]["params"]) != 1:
            return False
        g = opt.param_groups[0]
        p = g["params"][0]
        return (p.data_ptr() == policy.flat_param.data_ptr() and p.numel() == policy.flat_param.numel() and g.get("capturable", False)
                and not g.get("amsgrad") and not g.get("maximize") and g.get("weight_decay", 0) == 0 and not g.get("decoupled_weight_decay", False)
                and not isinstance(g["lr"], torch.Tensor))

    def _forward_backward_mlp(self, obs, actions, old_logp, adv, ret, index=None):
        """Forward, SB3's loss, backward and every weight gradient in ONE kernel (csrc/amenv_mlp_train.hpp): the gradient lands in the flat
        buffer, the four reported scalars in `stats`.  fp32 throughout (v_mfma_f32_32x32x2_f32), autograd is not involved.  With `index`
        (int64 [n]) the minibatch is rows `index` of the given tensors, gathered inside the kernel."""
        pol = self.policy
        if self._mlp_ws is None:
            self._mlp_ws = torch.empty(L.load().amenv_ppo_mlp_workspace_bytes() // 8 + 2, dtype=torch.float64, device=obs.device)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        obs, actions = obs.contiguous(), actions.contiguous()
        n = obs.shape[0] if index is None else index.shape[0]
        rc = L.load().amenv_ppo_mlp_step(p(pol.flat_param.detach()), pol.obs_dim, pol.act_dim, p(obs), p(actions), p(old_logp), p(adv), p(ret),
                                         None if index is None else p(index), n, self.clip_range, self.ent_coef, self.vf_coef, 1 if self.normalize_advantage else 0, p(pol.flat_grad), p(self.stats),
                                         p(self._mlp_ws), C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
        if rc != 0:
            raise L.AmenvError(f"amenv_ppo_mlp_step failed ({rc})")

    def _exchange(self):
        if self.dist is not None and (self.world > 1 or self.split):
            self.dist.all_reduce(self.policy.flat_grad)
            if self.world > 1 and not self.fused_adam:   # (the fused Adam step folds 1 / world into its gradient scale)
                self.policy.flat_grad.div_(self.world)

    def _apply_fused(self):
        """Clip + Adam as one launch on the optimiser's own state (so `optimizer.state_dict()` and checkpoints stay torch's)."""
        opt, pol = self.optimizer, self.policy
        g = opt.param_groups[0]
        leaf = g["params"][0]
        st = opt.state[leaf]
        if not st:                                       # torch.optim.Adam creates its state on the first step
            st["step"] = torch.zeros((), dtype=torch.float32, device=leaf.device)
            st["exp_avg"] = torch.zeros_like(pol.flat_param)
            st["exp_avg_sq"] = torch.zeros_like(pol.flat_param)
        key = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
               float(self.max_grad_norm) if self.max_grad_norm is not None else 0.0, 1.0 / self.world)
        if key != self._adam_key:
            if self._adam_hyper is None:
                self._adam_hyper = torch.empty(6, dtype=torch.float32, device=leaf.device)
                self._adam_ticket = torch.zeros(1, dtype=torch.int32, device=leaf.device)
            self._adam_hyper.copy_(torch.tensor(key, dtype=torch.float32))
            self._adam_key = key
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        rc = L.load().amenv_ppo_adam_step(p(pol.flat_param.detach()), p(pol.flat_grad), p(st["exp_avg"]), p(st["exp_avg_sq"]), p(st["step"]), pol.flat_param.numel(),
                                          p(self._adam_hyper), C.c_void_p(self.stats.data_ptr() + 16), p(self._adam_ticket),
                                          C.c_void_p(torch.cuda.current_stream(leaf.device).cuda_stream))
        if rc != 0:
            raise L.AmenvError(f"amenv_ppo_adam_step failed ({rc})")

    def _apply(self):
        if self.fused_adam:
            return self._apply_fused()
        g = self.policy.flat_grad
        gn = g.norm(2)
        if self.max_grad_norm is not None:
            g.mul_(torch.clamp(self.max_grad_norm / (gn + 1e-6), max=1.0))
        self.optimizer.step()
        self.stats[4] = gn

    def _eager(self, *mb):
        self._forward_backward(*mb)
        self._exchange()
        self._apply()

    def indexed(self, obs, actions, old_logp, adv, ret, index):
        """One minibatch step on rows `index` of the whole-rollout tensors (fused path only): no gathered copies, no graph."""
        self._forward_backward_mlp(obs, actions, old_logp, adv, ret, index)
        self._exchange()
        self._apply()

    # -- graph capture --------------------------------------------------------------------------
    def _capture(self, mb):
        self._static = tuple(torch.empty_like(t) for t in mb)
        for s, t in zip(self._static, mb):
            s.copy_(t)
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        if not self.split:
            with torch.cuda.graph(g1):
                self._forward_backward(*self._static)
                self._apply()
            self._graphs = (g1, None)
        else:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self._forward_backward(*self._static)
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._apply()
            self._graphs = (g1, g2)

    def __call__(self, obs, actions, old_logp, adv, ret):
        mb = (obs, actions, old_logp, adv, ret)
        full = self._static is None or obs.shape[0] == self._static[0].shape[0]
        if not self.use_graph or not full:
            return self._eager(*mb)
        if self._graphs is None:
            if self._eager_calls < 3:                    # warm-up: real updates, run eagerly (library handles, autotuning)
                self._eager_calls += 1
                return self._eager(*mb)
            try:
                self._capture(mb)                        # records only; the replay below performs this minibatch
            except Exception as e:  # noqa: BLE001 - capture is an optimisation: report and keep training eagerly
                import warnings
                warnings.warn(f"PPO minibatch graph capture failed ({type(e).__name__}: {e}); continuing without graphs")
                self.use_graph, self._static, self._graphs = False, None, None
                return self._eager(*mb)
        else:
            for s, t in zip(self._static, mb):
                s.copy_(t)
        g1, g2 = self._graphs
        g1.replay()
        if g2 is not None:
            self._exchange()
            g2.replay()


def ppo_update(policy, optimizer, obs, actions, old_logp, advantages, returns, *, batch_size, n_epochs, clip_range=0.2,
               ent_coef=5e-4, vf_coef=0.5, max_grad_norm=0.5, normalize_advantage=True, generator=None, dist=None, step=None):
    """SB3 `PPO.train()` on flattened rollout tensors (obs [n, D], actions [n, A], the rest [n]).  Device-agnostic torch;
    with `dist` (an initialised torch.distributed, RCCL on GPUs) every minibatch gradient is averaged over ranks with ONE
    all-reduce of the policy's flat gradient buffer.  `step`: a persistent `MinibatchStep` (keeps its captured graphs
    between calls).  Returns the mean losses of the last epoch (one host sync)."""
    n = obs.shape[0]
    if step is None:
        step = MinibatchStep(policy, optimizer, clip_range=clip_range, ent_coef=ent_coef, vf_coef=vf_coef, max_grad_norm=max_grad_norm,
                             normalize_advantage=normalize_advantage, dist=dist, use_graph=False)
    total = torch.zeros(5, device=obs.device)
    n_batches = 0
    for epoch in range(n_epochs):
        # one shuffle of the whole buffer per epoch (5 gathers), then every minibatch is a contiguous slice
        perm = torch.randperm(n, device=obs.device, generator=generator)
        indexed = step.fused_mlp and not step.use_graph
        if not indexed:
            obs_s, act_s, olp_s, adv_s, ret_s = obs[perm], actions[perm], old_logp[perm], advantages[perm], returns[perm]
        for start in range(0, n, batch_size):
            sl = slice(start, min(start + batch_size, n))
            if indexed:                                  # the kernel reads rows perm[sl] itself
                step.indexed(obs, actions, old_logp, advantages, returns, perm[sl])
            else:
                step(obs_s[sl], act_s[sl], olp_s[sl], adv_s[sl], ret_s[sl])
            if epoch == n_epochs - 1:   # losses are reported for the last epoch only (no host sync inside the loop)
                total += step.stats
                n_batches += 1
    s = (total / max(n_batches, 1)).tolist()
    return dict(policy_loss=s[0], value_loss=s[1], entropy_loss=s[2], clip_fraction=s[3], grad_norm=s[4])


class PPO:
    """The reference's training loop (`model = PPO(...); model.learn(...)`, v2/rl_train.py:38-56) with the rollout resident
    on the GPU.  `env` is a `GpuWaypointEnv` (auto-reset on); defaults are the reference's hyper-parameters -- with thousands
    of envs scale `n_steps` down and `batch_size` up."""

    def __init__(self, env, policy=None, learning_rate=2e-4, n_steps=2048, batch_size=128, n_epochs=12, gamma=0.995,
                 gae_lambda=0.9, clip_range=0.2, ent_coef=5e-4, vf_coef=0.5, max_grad_norm=0.5, normalize_advantage=True,
                 net_arch=(128, 64, 64), seed=0, obs_normalizer=None, bootstrap_truncated=True, dist=None, use_graph=None, fused_rollout=False, fused_mlp=None, fused_rollout_fp32_stats=True):
        if env.state_dtype != torch.float32:
            raise L.AmenvError("PPO needs the fp32 environment")
        self.env, self.dist = env, dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.device = env.device
        self.n_steps, self.batch_size, self.n_epochs = int(n_steps), int(batch_size), int(n_epochs)
        self.gamma, self.gae_lambda, self.clip_range = float(gamma), float(gae_lambda), float(clip_range)
        self.ent_coef, self.vf_coef, self.max_grad_norm = float(ent_coef), float(vf_coef), max_grad_norm
        self.normalize_advantage, self.bootstrap_truncated = bool(normalize_advantage), bool(bootstrap_truncated)
        self.seed = int(seed)
        torch.manual_seed(self.seed)   # identical initial weights on every rank (also broadcast below)
        if policy is None:
            policy = ActorCritic(env.obs_dim, env.act_dim, net_arch)
        if policy.obs_dim != env.obs_dim or policy.act_dim != env.act_dim:
            raise L.AmenvError(f"policy is {policy.obs_dim}->{policy.act_dim}, env is {env.obs_dim}->{env.act_dim}")
        self.policy = policy.to(self.device).flatten_()
        if self.world > 1:
            dist.broadcast(self.policy.flat_param, src=0)
        self._leaf = self.policy.flat_param.requires_grad_(True)
        self._leaf.grad = self.policy.flat_grad
        self.optimizer = torch.optim.Adam([self._leaf], lr=learning_rate, eps=1e-5, capturable=self.device.type == "cuda")
        self._step = MinibatchStep(self.policy, self.optimizer, clip_range=self.clip_range, ent_coef=self.ent_coef, vf_coef=self.vf_coef,
                                   max_grad_norm=self.max_grad_norm, normalize_advantage=self.normalize_advantage,
                                   dist=dist if self.world > 1 else None, use_graph=use_graph, fused_mlp=fused_mlp)
        self.obs_normalizer = obs_normalizer
        self.buffer = RolloutBuffer(self.n_steps, env.num_envs, env.obs_dim, env.act_dim, self.device)
        self._clipped = torch.zeros(env.num_envs, env.act_dim, dtype=torch.float32, device=self.device)
        self._raw_obs = torch.zeros(env.num_envs, env.obs_dim, dtype=torch.float32, device=self.device) if obs_normalizer else None
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(self.seed * 1000003 + int(getattr(env.cfg, "env_id_offset", 0)))
        self._draw = 0
        self._started = False
        # opt-in: the whole rollout (policy MLPs on the bf16 matrix cores, sampling, clip, env step) in ONE launch -- amenv_rollout_policy
        self.fused_rollout = bool(fused_rollout)
        self.fused_rollout_fp32_stats = bool(fused_rollout_fp32_stats)   # fused rollout: store the fp32 policy's log-probs / values (SB3's buffer semantics)
        if self.fused_rollout and obs_normalizer is not None:
            raise L.AmenvError("fused_rollout does not go through an observation normaliser")
        self._term_obs = self._info = None
        self.num_timesteps = 0
        self.log = []

    # ---- rollout --------------------------------------------------------------------------------
    def _first_obs(self):
        obs = self.env.reset()
        if self.obs_normalizer is not None:
            self.obs_normalizer.update(obs)
            self.obs_normalizer.normalize(obs, out=self.buffer.obs[0])
        else:
            self.buffer.obs[0].copy_(obs)
        self._started = True

    @torch.no_grad()
    def collect_rollouts(self):
        """SB3 `OnPolicyAlgorithm.collect_rollouts`: n_steps steps of every env into the buffer, then GAE."""
        b, env, pol = self.buffer, self.env, self.policy
        if self.fused_rollout:
            return self._collect_rollouts_fused()
        if not self._started:
            self._first_obs()
        else:
            b.obs[0].copy_(b.obs[self.n_steps])
        gid0 = int(env.cfg.env_id_offset)
        for t in range(self.n_steps):
            obs_t = b.obs[t]
            mean_t, value_t = pol.actor_critic(obs_t)
            b.values[t].copy_(value_t)
            gaussian_act(mean_t, pol.log_std.data, pol.action_low, pol.action_high, b.actions[t], self._clipped, b.logp[t],
                         self.seed, self._draw, gid0)
            self._draw += 1
            if self.obs_normalizer is None:
                env.step_into(self._clipped, b.obs[t + 1], b.rewards[t], b.dones[t])
                term_obs = env.terminal_obs
            else:   # VecNormalize(norm_obs=True): statistics from the raw observations, the learner sees normalised ones
                env.step_into(self._clipped, self._raw_obs, b.rewards[t], b.dones[t])
                self.obs_normalizer.update(self._raw_obs)
                self.obs_normalizer.normalize(self._raw_obs, out=b.obs[t + 1])
                term_obs = self.obs_normalizer.normalize(env.terminal_obs) if self.bootstrap_truncated else None
            if self.bootstrap_truncated:
                # SB3: reward += gamma * V(terminal_observation) where the episode was cut by the time limit only
                trunc = ((env.info_bits & (L.INFO_TERMINATED | L.INFO_TRUNCATED)) == L.INFO_TRUNCATED) & (b.dones[t] != 0)
                b.rewards[t].addcmul_(pol.critic(term_obs), trunc.to(torch.float32), value=self.gamma)
        b.last_values.copy_(pol.critic(b.obs[self.n_steps]))
        compute_gae(b, self.gamma, self.gae_lambda)
        self.num_timesteps += self.n_steps * env.num_envs * self.world
        return b

    def _collect_rollouts_fused(self):
        """collect_rollouts as ONE kernel launch (csrc/amenv_team_policy.hpp): n_steps x (actor / critic forward in bf16 on the matrix
        cores -> Gaussian sample -> clip -> env step) with state, constants and weights in registers; the buffer rows are written by the
        kernel.  What stays on the host side of the launch: the time-limit bootstrap reward += gamma V(terminal_observation) (one critic
        call over the truncated entries), V of the last observation, GAE.  The rollout policy is the bf16 rounding of the fp32 policy the
        update differentiates (means differ by ~1e-2 of their scale): PPO's clipped ratio absorbs that; log-probs are those of the
        samples under the means the kernel used."""
        b, env, pol, T = self.buffer, self.env, self.policy, self.n_steps
        if not self._started:
            env.reset()
            self._started = True
        if self._term_obs is None:
            self._term_obs = torch.zeros(T, env.num_envs, env.obs_dim, dtype=torch.float32, device=self.device)
            self._info = torch.zeros(T, env.num_envs, dtype=torch.int32, device=self.device)
        env.rollout_policy(pol.flat_param, T, self.seed, self._draw, b.obs, b.actions, b.logp, b.values, b.rewards, b.dones, self._info,
                           self._term_obs if self.bootstrap_truncated else None)
        self._draw += T
        if self.fused_rollout_fp32_stats:
            # SB3's buffer holds log pi(a|s) and V(s) of the policy the update differentiates: re-evaluate the T*N rows with the fp32 policy (the
            # fused forward kernel: one launch) so that the first epoch's ratio is exactly 1 and the value targets are the fp32 critic's; the
            # samples themselves stay those of the bf16 behaviour policy (its means are ~1e-2 of their scale away)
            flat_obs = b.obs[:T].reshape(T * env.num_envs, env.obs_dim)
            mean, value = pol.actor_critic(flat_obs)
            ls = pol.log_std.data
            zz = (b.actions.reshape(T * env.num_envs, -1) - mean) * torch.exp(-ls)
            b.logp.copy_((-0.5 * zz * zz - ls - 0.918938533204672742).sum(1).reshape(T, env.num_envs))
            b.values.copy_(value.reshape(T, env.num_envs))
        if self.bootstrap_truncated:
            trunc = ((self._info & (L.INFO_TERMINATED | L.INFO_TRUNCATED)) == L.INFO_TRUNCATED) & (b.dones != 0)
            idx = trunc.reshape(-1).nonzero().reshape(-1)
            if idx.numel():
                v = pol.critic(self._term_obs.reshape(-1, env.obs_dim).index_select(0, idx))
                b.rewards.reshape(-1).index_add_(0, idx, self.gamma * v)
        b.last_values.copy_(pol.critic(b.obs[T]))
        compute_gae(b, self.gamma, self.gae_lambda)
        self.num_timesteps += T * env.num_envs * self.world
        return b

    # ---- update ---------------------------------------------------------------------------------
    def train(self):
        """SB3 `PPO.train`: n_epochs passes over the buffer in shuffled minibatches."""
        b, T = self.buffer, self.n_steps
        n = T * b.n_envs
        return ppo_update(self.policy, self.optimizer, b.obs[:T].reshape(n, -1), b.actions.reshape(n, -1), b.logp.reshape(n),
                          b.advantages.reshape(n), b.returns.reshape(n), batch_size=self.batch_size, n_epochs=self.n_epochs,
                          generator=self._gen, step=self._step)

    def learn(self, total_timesteps, log_fn=None, save_freq=None, save_path=None, name_prefix="ppo_model"):
        """`model.learn(total_timesteps, callback=CheckpointCallback(save_freq, save_path, name_prefix))` (v2/rl_train.py:14-18,56):
        alternate rollouts and updates until the whole job has taken `total_timesteps` env steps.  Per iteration one record: losses +
        the Monitor episode statistics of the rollout.  `save_freq` counts env steps of the whole job (SB3 counts `VecEnv.step`
        calls: the reference's 12,500 calls x 8 envs = 100,000 steps); rank 0 writes `<save_path>/<name_prefix>_<steps>_steps.zip`
        at the first iteration boundary at or past every multiple."""
        target = self.num_timesteps + int(total_timesteps)
        next_save = None if not save_freq else (self.num_timesteps // int(save_freq) + 1) * int(save_freq)
        while self.num_timesteps < target:
            self.env.stats(reset=True)
            self.collect_rollouts()
            ep = self.env.stats(reset=True)
            rec = self.train()
            n_ep = max(int(ep["episodes"]), 1)
            rec.update(timesteps=self.num_timesteps, episodes=int(ep["episodes"]), ep_rew_mean=ep["return_sum"] / n_ep,
                       ep_len_mean=ep["length_sum"] / n_ep, success_rate=ep["success"] / n_ep)
            self.log.append(rec)
            if log_fn is not None:
                log_fn(rec)
            if next_save is not None and self.num_timesteps >= next_save:
                if self.dist is None or self.dist.get_rank() == 0:
                    os.makedirs(save_path or ".", exist_ok=True)
                    self.save(os.path.join(save_path or ".", f"{name_prefix}_{self.num_timesteps}_steps"))
                next_save = (self.num_timesteps // int(save_freq) + 1) * int(save_freq)
        return self

    def predict(self, obs, deterministic=True):
        if self.obs_normalizer is not None:
            obs = self.obs_normalizer.normalize(obs)
        return self.policy.predict(obs, deterministic, self._gen)

    # ---- checkpoints ----------------------------------------------------------------------------
    def save(self, path):
        """A zip with SB3's member names: `policy.pth` (SB3 keys: loadable into an SB3 `ActorCriticPolicy`),
        `policy.optimizer.pth` (this class's Adam state) and `data` (hyper-parameters as plain JSON).  It is NOT a complete
        SB3 archive (SB3's `data` holds cloudpickled objects); use `policy.pth` to move weights either way."""
        path = path if str(path).endswith(".zip") else str(path) + ".zip"
        data = {k: getattr(self, k) for k in ("n_steps", "batch_size", "n_epochs", "gamma", "gae_lambda", "clip_range", "ent_coef",
                                              "vf_coef", "max_grad_norm", "normalize_advantage", "num_timesteps", "seed")}
        data.update(format="amenv-ppo", draw=int(self._draw), learning_rate=self.optimizer.param_groups[0]["lr"], net_arch=list(self.policy.net_arch),
                    obs_dim=self.policy.obs_dim, act_dim=self.policy.act_dim)
        with zipfile.ZipFile(path, "w") as z:
            for name, obj in (("policy.pth", {k: v.detach().cpu().clone() for k, v in self.policy.state_dict().items()}),
                              ("policy.optimizer.pth", self.optimizer.state_dict())):
                buf = io.BytesIO()
                torch.save(obj, buf)
                z.writestr(name, buf.getvalue())
            z.writestr("data", json.dumps(data))
        return path

    def load_policy(self, path_or_state_dict):
        """Resume from a checkpoint's weights (`PPO.load(CHECKPOINT_PATH, ...)`, v2/rl_train.py:33-35): SB3 zip, this class's
        zip or a bare `policy.pth`.  Copies into the flat buffer in place."""
        sd = path_or_state_dict if isinstance(path_or_state_dict, dict) else ActorCritic.read_sb3_state_dict(path_or_state_dict)
        own = self.policy.state_dict()
        if set(sd) != set(own):
            raise L.AmenvError(f"checkpoint keys differ: {sorted(set(sd) ^ set(own))}")
        with torch.no_grad():
            for k, v in own.items():
                if tuple(sd[k].shape) != tuple(v.shape):
                    raise L.AmenvError(f"{k}: checkpoint shape {tuple(sd[k].shape)} != policy {tuple(v.shape)}")
                v.copy_(torch.as_tensor(sd[k]).to(v.device))
        return self

    def load(self, path):
        """Full resume from a zip written by `save()`: weights, Adam moments / step count, learning rate and the timestep counter
        (`PPO.load(CHECKPOINT_PATH, env=...)` followed by `learn`, v2/rl_train.py:33-35,56).  An SB3 archive has no optimiser state
        this class can read without unpickling: for those use `load_policy` (weights only)."""
        with zipfile.ZipFile(path if str(path).endswith(".zip") else str(path) + ".zip") as z:
            self.load_policy(torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True))
            names = z.namelist()
            data = json.loads(z.read("data")) if "data" in names else {}
            if "policy.optimizer.pth" in names and data.get("format") == "amenv-ppo":
                osd = torch.load(io.BytesIO(z.read("policy.optimizer.pth")), map_location="cpu", weights_only=True)
                cur = self.optimizer.state_dict()
                if len(osd["param_groups"]) != len(cur["param_groups"]):
                    raise L.AmenvError("optimizer state of the checkpoint does not match this optimizer")
                # copy into the live tensors: a captured update graph keeps pointing at them
                st = self.optimizer.state[self._leaf]
                src = osd["state"].get(0, {})
                if src:
                    if not st:   # Adam creates its moments lazily
                        st["step"] = torch.zeros((), dtype=torch.float32, device=self.device)
                        st["exp_avg"] = torch.zeros_like(self._leaf); st["exp_avg_sq"] = torch.zeros_like(self._leaf)
                    for k in ("step", "exp_avg", "exp_avg_sq"):
                        st[k].copy_(torch.as_tensor(src[k]).to(st[k].device))
                for g, gs in zip(self.optimizer.param_groups, osd["param_groups"]):
                    g["lr"] = gs["lr"]
            self.num_timesteps = int(data.get("num_timesteps", self.num_timesteps))
            self._draw = int(data.get("draw", self._draw))   # action-noise counter: resumed rollouts draw fresh noise
        return self


@torch.no_grad()
def clone_pid_policy(env, policy, steps=600, epochs=400, dagger_rounds=0, noise=0.15, initial_log_std=-1.0, seed=1):
    """Warm start for `PPO`: fit the actor's mean to `PidWaypointPolicy` (the reference's PID + minimum-snap baseline, HIP kernels) by behaviour
    cloning on states the PID visits (exploration noise on the executed actions, so that the data covers recoveries); with `dagger_rounds` > 0
    the STUDENT then flies while the PID labels the states it reaches, and the actor is refitted on everything collected (DAgger).  Sets
    log_std to `initial_log_std`.  Returns the final MSE.

    Why: with SB3's default initialisation PPO (the reference's hyper-parameters, v2/rl_train.py:38-53) leaves the free-fall plateau on the
    reference's quadrotor after ~35 M steps but not on the hexacopter within 150 M -- whatever its moment scaling, rotor limits, inertia,
    exploration noise or thrust bias (profiles/r03/ppo_hexa_sweep_*.json); from a cloned actor -- even one that does not yet reach a single
    waypoint itself -- it reaches > 90 % success on the hexacopter in ~55 M steps and on the hexacopter + arm (tool-point task) in 35-77 M."""
    from .baselines import PidWaypointPolicy
    pid = PidWaypointPolicy.for_env(env)
    g = torch.Generator(device=env.device).manual_seed(seed)
    X, Y = [], []

    def collect(student):
        obs = env.reset()
        done = None
        if pid.pstate is not None:
            pid.pstate[:, 13] = 1.0          # every env starts a fresh episode (new minimum-snap segment)
        for _ in range(steps):
            a = pid.predict(obs, done)
            X.append(obs.clone()); Y.append(a.clone())
            fly = policy.actor(obs) if student else a
            noisy = fly + noise * torch.randn(a.shape, device=a.device, generator=g)
            obs, _, done, _ = env.step(torch.max(torch.min(noisy, policy.action_high), policy.action_low))

    def fit():
        Xc, Yc = torch.cat(X), torch.cat(Y)
        opt = torch.optim.Adam(list(policy.mlp_extractor.policy_net.parameters()) + list(policy.action_net.parameters()), lr=1e-3)
        loss = None
        for _ in range(epochs):
            idx = torch.randint(0, Xc.shape[0], (min(16384, Xc.shape[0]),), device=Xc.device)
            with torch.enable_grad():
                loss = ((policy.actor(Xc[idx]) - Yc[idx]) ** 2).mean()
                opt.zero_grad(); loss.backward(); opt.step()
        return float(loss.detach())

    with torch.no_grad():
        collect(False)
    mse = fit()
    for _ in range(int(dagger_rounds)):
        with torch.no_grad():
            collect(True)
        mse = fit()
    with torch.no_grad():
        policy.log_std.data.fill_(float(initial_log_std))
    return mse


def evaluate_policy(model, env, n_eval_episodes=10, deterministic=True, check_every=64, max_steps=None):
    """SB3 `evaluate_policy(model, env, n_eval_episodes)` (v2/rl_train.py:60) on the batched env: every env contributes the
    same number of episodes (ceil(n / num_envs), SB3's rule for vectorised envs), returns (mean, std) of the episode returns.
    `model`: a `PPO`, an `ActorCritic` or anything with `.predict(obs, deterministic)`.  The loop stays on the device; the host
    looks at the completion counters once every `check_every` steps."""
    n = env.num_envs
    per_env = -(-int(n_eval_episodes) // n)
    obs = env.reset()
    counts = torch.zeros(n, dtype=torch.int64, device=obs.device)
    total = torch.zeros(n, dtype=torch.float64, device=obs.device)
    total_sq = torch.zeros_like(total)
    lengths = torch.zeros(n, dtype=torch.int64, device=obs.device)
    limit = max_steps if max_steps is not None else per_env * (int(env.cfg.task.max_episode_steps) + 2)
    for t in range(limit):
        obs, _, done, _ = env.step(model.predict(obs, deterministic))
        take = (done != 0) & (counts < per_env)
        r = env.ep_return.to(torch.float64)
        total += torch.where(take, r, torch.zeros_like(r))
        total_sq += torch.where(take, r * r, torch.zeros_like(r))
        lengths += torch.where(take, env.ep_len.to(torch.int64), torch.zeros_like(lengths))
        counts += take.to(torch.int64)
        if (t + 1) % check_every == 0 and bool((counts >= per_env).all()):
            break
    k = int(counts.sum())
    if k == 0:
        raise L.AmenvError("evaluate_policy: no episode finished within the step limit")
    mean = float(total.sum()) / k
    var = max(float(total_sq.sum()) / k - mean * mean, 0.0)
    return mean, math.sqrt(var)
