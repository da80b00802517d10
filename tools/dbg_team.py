import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rl_aerial_manipulator_amd as amd
n, T = 1000, 120
g = torch.Generator(device="cuda").manual_seed(11)
acts = torch.randn(T, n, 7, device="cuda", generator=g) * 0.2
acts[..., 0] += 1.0
acts[:, ::3, 0] = 0.15
acts = acts.clamp(-1, 2).contiguous()
mk = lambda: amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=3, kernel="team", max_episode_steps=70)
e1, e2 = mk(), mk()
for e in (e1, e2): e.reset()
ro = e1.rollout(acts)
for t in range(T):
    o2, r2, d2, i2 = (x.clone() for x in e2.step(acts[t]))
    for name, a, b in (("obs", ro["obs"][t], o2), ("rew", ro["reward"][t], r2), ("info", ro["info_bits"][t], i2), ("done", ro["done"][t], d2)):
        if not torch.equal(a, b):
            bad = (a != b).nonzero()
            print("t", t, name, "mismatches", len(bad), "first", bad[:6].tolist())
            for idx in bad[:4].tolist():
                row = idx[0]
                print("  row", row, "vals", a[tuple(idx)].item(), b[tuple(idx)].item(), "info ro/step", hex(int(ro["info_bits"][t][row])), hex(int(i2[row])), "prev info", hex(int(ro["info_bits"][t-1][row])) if t else None)
            sys.exit(0)
print("all equal")
