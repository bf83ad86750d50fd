"""Where one training step's time goes (bench.py --mode train's step, phase by phase): HIP events around e0, the attack,
the weight-gradient forward+backward, AdamW and the weight refresh.  `python scripts/train_phases.py [--steps 3]`"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robustvlm_amd as R                                     # noqa: E402
from robustvlm_amd.trainer import AdversarialTrainer          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--model", default="ViT-L-14")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = R.CONFIGS[args.model]
sd = R.random_state_dict(cfg, seed=0, device=dev)
tr = AdversarialTrainer(cfg, sd, batch_size=args.batch, precision="bf16", attack="pgd", iterations_adv=10, device=dev)
x = torch.rand(args.batch, 3, cfg.image_size, cfg.image_size, device=dev)
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((name, e))


def wrap(obj, attr, name):
    f = getattr(obj, attr)

    def g(*a, **k):
        mark("<" + name)
        r = f(*a, **k)
        mark(">" + name)
        return r
    setattr(obj, attr, g)


wrap(tr, "model_orig", "e0_forward")
wrap(tr, "_attack", "attack")
wrap(tr.engine, "forward", "forward_save")
wrap(tr.engine, "backward_params", "backward_params")
wrap(tr.engine, "load_state_dict", "weight_refresh")
orig_adamw = tr.lib.rvlm_adamw_step


class LibProxy:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, k):
        f = getattr(self._lib, k)
        if k != "rvlm_adamw_step":
            return f

        def g(*a):
            mark("<adamw")
            r = f(*a)
            mark(">adamw")
            return r
        return g


tr.lib = LibProxy(tr.lib)
tr.train_step(x, None, global_batch=args.batch)
torch.cuda.synchronize()
acc = {}
wall = []
for _ in range(args.steps):
    marks.clear()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    tr.train_step(x, None, global_batch=args.batch)
    b.record()
    torch.cuda.synchronize()
    wall.append(a.elapsed_time(b))
    open_ = {}
    for name, e in marks:
        if name[0] == "<":
            open_.setdefault(name[1:], []).append(e)
        else:
            s = open_[name[1:]].pop()
            acc.setdefault(name[1:], []).append(s.elapsed_time(e))
n = args.steps
out = {k: round(sum(v) / n, 3) for k, v in acc.items()}
# forward_save is also called inside the attack wrapper? no: the attack goes through rvlm_pgd_run, not engine.forward
out["step_ms"] = round(sum(wall) / n, 3)
out["unaccounted_ms"] = round(out["step_ms"] - sum(v for k, v in out.items() if k != "step_ms"), 3)
print(json.dumps(out))
