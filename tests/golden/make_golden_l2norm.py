#!/usr/bin/env python3
"""Golden vectors of the L2-NORM branch of pgd() (train/pgd_train.py:38-63 with vlm_eval/attacks/utils.py:12-14,22-26),
produced by RUNNING THE REFERENCE in the build container (needs /root/reference; never runs on the GPU box).

    python tests/golden/make_golden_l2norm.py    ->  tests/golden/pgd_l2norm.npz

Contents: (a) injected-gradient sequences (NaNs, zeros, a sample whose gradient is all zero, one whose perturbation
leaves the eps ball and one that stays inside) for 1..4 iterations, both modes; (b) pgd(norm='l2') end to end on the
tiny seeded ViT with the FARE loss (the reference's own loss definitions, exec'd from its source as make_golden.py does).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG   # noqa: E402  (imports the reference, defines save / REFL / helpers; generates nothing on import)

ref_pgd, REFL = MG.ref_pgd, MG.REFL


class Inject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, G):
        ctx.save_for_backward(G)
        return v.sum() * 0.0 + 1.0

    @staticmethod
    def backward(ctx, go):
        (G,) = ctx.saved_tensors
        return G.clone(), None


def main():
    arrs = {}
    g = torch.Generator().manual_seed(202)
    shape = (5, 3, 8, 8)
    eps, step = 0.5, 0.2
    x = torch.rand(shape, generator=g)
    d0 = torch.zeros(shape).uniform_(-0.02, 0.02, generator=g)
    d0[1] *= 30.0                                   # starts outside the eps ball: renorm acts at once
    grads = []
    for i in range(4):
        gi = torch.randn(shape, generator=g)
        gi[2] = 0.0                                 # all-zero gradient for one sample: F.normalize's eps path
        gi.view(-1)[3 + i::29] = float("nan")
        gi[3] *= 1e-20
        grads.append(gi)

    def fwd(v, output_normalize=False):
        return v

    for mode in ("max", "min"):
        outs = []
        for n_it in range(1, 5):
            it = iter(range(n_it))

            def loss_fn(out, targets):
                return Inject.apply(out, grads[next(it)])

            outs.append(ref_pgd(fwd, loss_fn, x, None, "l2", eps, n_it, step, False,
                                perturbation=d0.clone().requires_grad_(True), mode=mode).numpy())
        arrs[f"ew_xadv_{mode}"] = np.stack(outs)
    arrs.update(ew_x=x.numpy(), ew_delta0=d0.numpy(), ew_grads=np.stack([t.numpy() for t in grads]),
                ew_eps=np.float64(eps), ew_stepsize=np.float64(step))

    # end to end on the tiny ViT of tiny_vit_attacks.npz (same weights / images)
    z = np.load(os.path.join(MG.OUT, "tiny_vit_attacks.npz"))
    cfg = MG.VitConfig(*[int(v) for v in z["cfg"]])
    w = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    model = MG.ClipVisionModelRef(cfg, w).eval()
    xt = torch.from_numpy(z["x"])
    with torch.no_grad():
        e0 = model(xt, False)
    wrap = REFL["ComputeLossWrapper"](e0, None, "mean", "l2", 100.)
    d0t = torch.from_numpy(z["delta0"])
    xadv = ref_pgd(model, wrap, xt, None, "l2", 1.0, 10, 0.25, False, perturbation=d0t.clone().requires_grad_(True),
                   mode="max")
    arrs["vit_xadv"] = xadv.numpy()
    with torch.no_grad():
        arrs["vit_loss_final"] = np.float32(wrap(model(xadv, False), None).item())
    arrs["vit_eps"], arrs["vit_stepsize"] = np.float64(1.0), np.float64(0.25)
    # apgd_train(norm='l2') on the same model: FARE (l2 loss, reduction none) and TeCoA (ce), 10 iterations, plus a
    # 25-iteration run whose checkpoints halve step sizes
    yt, T = torch.from_numpy(z["y"]), torch.from_numpy(z["T"])
    with torch.no_grad():
        # the frozen model_orig differs from the model under attack once training has started (a FARE attack from the
        # clean image against the model's own embedding has zero loss and zero gradient): emulate with a shifted target
        noise = torch.randn(xt.shape, generator=torch.Generator().manual_seed(77)) * 0.05
        e0n = model((xt + noise).clamp(0, 1), True)
    arrs["apgd_e0"] = e0n.numpy()
    for loss_name in ("l2", "ce"):
        wrapn = REFL["ComputeLossWrapper"](e0n, T, "none", loss_name, 100.)
        for n_iter in (10, 25):
            xa = MG.ref_apgd_train(model, xt, yt, "l2", 1.0, n_iter=n_iter, loss_fn=wrapn)
            arrs[f"apgd_{loss_name}_{n_iter}_xadv"] = xa.numpy()
            with torch.no_grad():
                arrs[f"apgd_{loss_name}_{n_iter}_loss_final"] = wrapn(model(xa, True), yt).numpy()
    MG.save("pgd_l2norm.npz", **arrs)


if __name__ == "__main__":
    main()
