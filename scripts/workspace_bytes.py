import torch, sys, os
sys.path.insert(0, os.getcwd())
import robustvlm_amd as R
from robustvlm_amd.config import CONFIGS, state_dict_shapes
cfg = CONFIGS["ViT-L-14"]
g = torch.Generator().manual_seed(0)
sd = {k: (torch.randn(*s, generator=g) * 0.02).cuda() for k, s in state_dict_shapes(cfg).items()}
for p in ("bf16", "x3", "fp32", "bf16+x3fwd-first", "bf16+x3-first", "bf16+fp32-first"):
    e = R.VitEngine(cfg, sd, precision=p, max_batch=128)
    print(p, round(e.workspace_bytes() / 2**30, 1), "GiB")
    e.close()
