import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rl_aerial_manipulator_amd as amd
from rl_aerial_manipulator_amd.ppo import MinibatchStep, ActorCritic
for (D, A) in ((29, 7), (20, 4)):
    n = 65536
    pol = ActorCritic(D, A).cuda().flatten_()
    opt = torch.optim.Adam([pol.flat_param.requires_grad_(True)], lr=1e-3)
    obs = torch.randn(n, D, device="cuda"); actions = torch.randn(n, A, device="cuda"); olp = torch.randn(n, device="cuda") * 0.1 - 5
    adv = torch.randn(n, device="cuda"); ret = torch.randn(n, device="cuda")
    for fused in (False, True):
        step = MinibatchStep(pol, opt, use_graph=False, fused_mlp=fused)
        for _ in range(3): step._forward_backward(obs, actions, olp, adv, ret)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): step._forward_backward(obs, actions, olp, adv, ret)
        e1.record(); torch.cuda.synchronize()
        print(f"D={D} A={A} n={n} fused_mlp={fused}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per forward+backward")

# the forward passes alone (amenv_policy_forward_mfma through ActorCritic.forward_fused): the rollout buffer's re-evaluation, a 32768-env policy step
for n in (32768, 1048576):
    pol = ActorCritic(29, 7).cuda().flatten_()
    obs = torch.randn(n, 29, device="cuda")
    with torch.no_grad():
        for _ in range(3): pol.forward_fused(obs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): pol.forward_fused(obs)
        e1.record(); torch.cuda.synchronize()
    print(f"forward only, D=29 A=7 n={n}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (weight packing + mlp_forward_kernel)")
