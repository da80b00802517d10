"""Phase durations of the MAIN wave of step_kernel_arm2w from a -DAMENV_STAMPS build:
  AMENV_LIB=tools/micro/libamenv_stamps.so python tools/stamp_arm2w.py
slots: 0 entry, 1 loads issued, 2 loads landed, 7 stores drained (absolute s_memtime); 3..6 accumulated over the 4 RHS of the step:
own share of the RHS | waiting in barrier 1 | gets + solve + puts | waiting in barrier 2."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rl_aerial_manipulator_amd as amd
env = amd.GpuWaypointEnv(4096, vehicle="hexa_arm", seed=0); env.reset()
lib = C.CDLL(amd._lib.LIB_PATH); lib.amenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(1)
ring = (torch.randn(16, 4096, 7, device="cuda", generator=g) * 0.1); ring[..., 0] += 1.0; ring[..., 4:] *= 3.0; ring = ring.clamp(-1, 2).contiguous()
rows = []
for t in range(120):
    env.step(ring[t % 16]); buf = np.zeros((64, 8), np.uint64); lib.amenv_debug_stamps(env._h, buf.ctypes.data_as(C.c_void_p)); rows.append(buf.astype(np.int64))
R = np.stack(rows[20:]).reshape(-1, 8)
tot = R[:, 7] - R[:, 0]
med = lambda a: float(np.median(a))
print(f"lifetime {med(tot):8.0f}   entry->loads landed {med(R[:,2]-R[:,0]):7.0f}")
for k, n in zip((3, 4, 5, 6), ("own share of 4 RHS", "barrier 1 (wait for helper)", "gets + solve + puts", "barrier 2")):
    print(f"  {n:28s} {med(R[:,k]):8.0f}  ({100*med(R[:,k])/med(tot):4.1f} %)")
print(f"  {'rest (RK4 axpy, task, stores)':28s} {med(tot - (R[:,2]-R[:,0]) - R[:,3]-R[:,4]-R[:,5]-R[:,6]):8.0f}")
