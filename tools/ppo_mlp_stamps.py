import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rl_aerial_manipulator_amd as amd
from rl_aerial_manipulator_amd.ppo import MinibatchStep, ActorCritic
D, A, n = 29, 7, 65536
pol = ActorCritic(D, A).cuda().flatten_()
opt = torch.optim.Adam([pol.flat_param.requires_grad_(True)], lr=1e-3)
obs = torch.randn(n, D, device="cuda"); actions = torch.randn(n, A, device="cuda"); olp = torch.randn(n, device="cuda") * 0.1 - 5
adv = torch.randn(n, device="cuda"); ret = torch.randn(n, device="cuda")
step = MinibatchStep(pol, opt, use_graph=False, fused_mlp=True)
for _ in range(10): step._forward_backward(obs, actions, olp, adv, ret)
torch.cuda.synchronize()
if "mlpstamps" in os.environ.get("AMENV_LIB", ""):
    import numpy as np
    ws = step._mlp_ws.view(torch.float32)
    adv_b, wt = 2 * 1024 * 2, 2 * 186 * 256   # floats: advantage partials (doubles), split-weight streams (186 KB per net)
    acc_size = 128*33 + 64*129 + 64*65 + 32*65 + 16
    blocks = 128
    part = ws[adv_b + wt: adv_b + wt + 2 * blocks * acc_size].reshape(2, blocks, acc_size).cpu().numpy()
    st = part[:, :, acc_size - 16 + 11: acc_size - 16 + 16]          # [net][block][phase]: sums over the 4 waves' lane 0
    per_wave = st.sum(1) / (blocks * 4)
    for net in (0, 1):
        print("net", net, "clocks per wave: forward %.0f  loss %.0f  weight-grad %.0f  data-grad %.0f  publish %.0f  total %.0f" % (*per_wave[net], per_wave[net].sum()))
