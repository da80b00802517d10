"""Per-parameter-block error of the fused PPO MLP step against autograd (debug aid for csrc/amenv_mlp_train.hpp)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import rl_aerial_manipulator_amd as amd  # noqa: F401
from rl_aerial_manipulator_amd.ppo import ActorCritic, MinibatchStep

D, A, n = 29, 7, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
torch.manual_seed(3)
pol = ActorCritic(D, A).cuda().flatten_()
with torch.no_grad():
    pol.log_std.data.copy_(torch.linspace(-0.7, 0.2, A))
    pol.action_net.weight.mul_(20.0)
opt = torch.optim.Adam([pol.flat_param.requires_grad_(True)], lr=1e-3)
g = torch.Generator(device="cuda").manual_seed(1)
obs = torch.randn(n, D, device="cuda", generator=g) * 0.7
with torch.no_grad():
    mean = pol.action_net(pol.mlp_extractor.policy_net(obs))
actions = mean + torch.exp(pol.log_std.detach()) * torch.randn(n, A, device="cuda", generator=g)
with torch.no_grad():
    _, logp, _ = pol.evaluate_actions(obs, actions)
old_logp = logp + 0.15 * torch.randn(n, device="cuda", generator=g)
adv = torch.randn(n, device="cuda", generator=g) * 3.0 + 0.5
ret = torch.randn(n, device="cuda", generator=g) * 2.0
outs = []
for fused in (False, True):
    step = MinibatchStep(pol, opt, clip_range=0.2, ent_coef=5e-4, vf_coef=0.5, use_graph=False, fused_loss=False, fused_mlp=fused)
    pol.flat_grad.zero_()
    step._forward_backward(obs, actions, old_logp, adv, ret)
    torch.cuda.synchronize()
    outs.append((pol.flat_grad.clone(), step.stats[:4].clone()))
(g0, s0), (g1, s1) = outs
scale = float(g0.abs().max())
print("scale", scale, "max err", float((g0 - g1).abs().max()), "stats", s0.tolist(), s1.tolist())
off = 0
for name, p_ in pol.named_parameters():
    k = p_.numel()
    if name == "flat_param":
        continue
    e = (g0[off:off + k] - g1[off:off + k]).abs()
    print(f"{name:45s} {tuple(p_.shape)!s:12s} max|g| {float(g0[off:off + k].abs().max()):.3e}  max err {float(e.max()):.3e}  argmax {int(e.argmax())}  ref {float(g0[off + int(e.argmax())]):+.4e} got {float(g1[off + int(e.argmax())]):+.4e}")
    off += k
