#!/usr/bin/env python3
"""Per-class time of ONE forward of the split-bf16 (x3) handle at the headline shape (ViT-L/14, B = 128) - what the mixed modes pay
twice per pgd() call (clean embedding + first iteration).  Usage: python scripts/x3_forward_profile.py [precision] [save]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robustvlm_amd as R  # noqa: E402
from robustvlm_amd.config import CONFIGS, state_dict_shapes  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "x3"
save = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = CONFIGS["ViT-L-14"]
g = torch.Generator().manual_seed(0)
sd = {k: (torch.randn(*s, generator=g) * 0.02) for k, s in state_dict_shapes(cfg).items()}
for k in sd:
    if k.endswith("ln_1.weight") or k.endswith("ln_2.weight") or k in ("ln_pre.weight", "ln_post.weight"):
        sd[k] = torch.ones_like(sd[k])
B = 128
eng = R.VitEngine(cfg, {k: v.cuda() for k, v in sd.items()}, precision=prec, max_batch=B)
x = torch.rand(B, 3, 224, 224, device="cuda")
for _ in range(2):
    eng.forward(x, None, False, save=bool(save))
torch.cuda.synchronize()
t0 = time.time()
n = 5
for _ in range(n):
    eng.forward(x, None, False, save=bool(save))
torch.cuda.synchronize()
print(f"{prec} forward save={save}: {(time.time() - t0) / n * 1e3:.1f} ms")
eng.set_profiling(True); eng.reset_profile()
for _ in range(n):
    eng.forward(x, None, False, save=bool(save))
torch.cuda.synchronize()
prof = eng.get_profile()
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"   {k:18s} {v['ms'] / n:8.2f} ms per forward  ({v['launches'] // n} launches)")
