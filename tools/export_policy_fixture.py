"""Export the tensors of the reference's best checkpoint (v2/README.md:55: checkpoints_from_8_6M/ppo_model_2300000_steps.zip)
as a plain .npz fixture (data, 30,537 fp32 numbers) so that GPU-box tests -- where /root/reference does not exist -- can fly the
reference's own trained policy on the new environment.  Loaded with torch.load(weights_only=True): nothing is executed.

    python tools/export_policy_fixture.py            # -> tests/golden/policy_2300000.npz
"""
import io
import os
import sys
import zipfile

import numpy as np
import torch

REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
CKPT = os.path.join(REF, "initial-implementation-v2", "checkpoints_from_8_6M", "ppo_model_2300000_steps.zip")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "policy_2300000.npz")

if __name__ == "__main__":
    with zipfile.ZipFile(CKPT) as z:
        sd = torch.load(io.BytesIO(z.read("policy.pth")), weights_only=True, map_location="cpu")
        version = z.read("_stable_baselines3_version").decode().strip()
    np.savez_compressed(OUT, **{k: v.numpy() for k, v in sd.items()}, _sb3_version=np.array(version),
                        _source=np.array("initial-implementation-v2/checkpoints_from_8_6M/ppo_model_2300000_steps.zip:policy.pth"))
    print(OUT, sum(v.numel() for v in sd.values()), "parameters", file=sys.stderr)
