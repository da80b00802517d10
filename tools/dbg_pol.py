import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rl_aerial_manipulator_amd as amd
n, T = 64, 2
env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=4, kernel="team")
dev = env.device
def run(pol):
    env.reset()
    obs = torch.zeros(T + 1, n, 29, device=dev); acts = torch.zeros(T, n, 7, device=dev)
    logp = torch.zeros(T, n, device=dev); vals = torch.zeros(T, n, device=dev); rew = torch.zeros(T, n, device=dev)
    dones = torch.zeros(T, n, dtype=torch.uint8, device=dev)
    env.rollout_policy(pol.flat_param, T, seed=1, draw0=0, obs=obs, actions=acts, logp=logp, values=vals, rewards=rew, dones=dones)
    torch.cuda.synchronize()
    with torch.no_grad():
        v32 = pol.value_net(pol.mlp_extractor.value_net(obs[0])).reshape(-1)
        m32 = pol.action_net(pol.mlp_extractor.policy_net(obs[0]))
    return obs, vals, v32, acts, m32
def zero(pol):
    with torch.no_grad():
        for p in pol.parameters(): p.zero_()
        pol.log_std.data.fill_(-20.0)   # noise off: actions = means
pol = amd.ActorCritic(29, 7).cuda().flatten_()
zero(pol)
with torch.no_grad(): pol.value_net.bias.fill_(0.7); pol.action_net.bias.copy_(torch.arange(7.0) * 0.1)
o, v, v32, a, m32 = run(pol); print("a: value bias only      kernel", v[0, :3].tolist(), "ref", v32[:3].tolist(), " actions", a[0, 0].tolist())
with torch.no_grad(): pol.value_net.weight.fill_(1.0); pol.mlp_extractor.value_net[4].bias.fill_(0.5)
o, v, v32, a, m32 = run(pol); print("b: + b3=.5, Wval=1       kernel", v[0, :3].tolist(), "ref", v32[:3].tolist())
with torch.no_grad(): pol.mlp_extractor.value_net[4].bias.zero_(); pol.mlp_extractor.value_net[4].weight.copy_(torch.eye(64) * 1.0); pol.mlp_extractor.value_net[2].bias.copy_(torch.linspace(-1, 1, 64))
o, v, v32, a, m32 = run(pol); print("c: W3=I, b2=linspace     kernel", v[0, :3].tolist(), "ref", v32[:3].tolist())
with torch.no_grad():
    pol.mlp_extractor.value_net[2].bias.zero_(); w2 = torch.zeros(64, 128); w2[torch.arange(64), torch.arange(64)] = 1.0; pol.mlp_extractor.value_net[2].weight.copy_(w2)
    pol.mlp_extractor.value_net[0].bias.copy_(torch.linspace(-1, 1, 128))
o, v, v32, a, m32 = run(pol); print("d: W2=[I 0], b1=linspace  kernel", v[0, :3].tolist(), "ref", v32[:3].tolist())
with torch.no_grad():
    pol.mlp_extractor.value_net[0].bias.zero_(); w1 = torch.zeros(128, 29); w1[torch.arange(29), torch.arange(29)] = 1.0; pol.mlp_extractor.value_net[0].weight.copy_(w1)
o, v, v32, a, m32 = run(pol); print("e: W1=[I;0]               kernel", v[0, :3].tolist(), "ref", v32[:3].tolist())
torch.manual_seed(0)
pol2 = amd.ActorCritic(29, 7).cuda().flatten_()
with torch.no_grad(): pol2.log_std.data.fill_(-20.0)
o, v, v32, a, m32 = run(pol2); print("f: random init            kernel", v[0, :3].tolist(), "ref", v32[:3].tolist(), "\n   means kernel", a[0, 0].tolist(), "\n   ref  ", m32[0].tolist())
