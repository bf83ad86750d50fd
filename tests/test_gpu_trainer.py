"""Outer training step: weight gradients, AdamW and the step semantics against
torch autograd + torch.optim.AdamW on the oracle ViT."""
import numpy as np
import pytest
import torch

import robustvlm_amd as R
from robustvlm_amd import _lib as L
from robustvlm_amd.trainer import AdversarialTrainer, FlatParams, cosine_lr_value
from oracle import vit_ref as V
from oracle.train_ref import TrainStepRef, cosine_lr_ref
from tests.gpu_helpers import dev, cos_sim, rel_max

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)


def to_cfg(c):
    return R.VitConfig(c.image_size, c.patch, c.width, c.layers, c.heads, c.out_dim, c.act)


# width 1024 / 65 tokens / B=8: the encoder-shaped case - weight gradients take the split-K persistent-GEMM path
VIT_WIDE = V.VitConfig(64, 8, 1024, 2, 16, 64)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("cfg,B,norm", [(V.VIT_TINY, 4, False), (V.VIT_TINY2, 3, True), (VIT_WIDE, 8, True)])
def test_weight_gradients_vs_autograd(cfg, B, norm, precision):
    w = V.init_weights(cfg, seed=11)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g)
    cot = torch.randn(B, cfg.out_dim, generator=g)
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    e = V.vit_forward(cfg, wr, V.normalize_pixels(x))
    if norm:
        e = torch.nn.functional.normalize(e, dim=-1)
    (e * cot).sum().backward()
    eng = R.VitEngine(to_cfg(cfg), {k: v.to(dev()) for k, v in w.items()}, precision=precision, max_batch=B,
                      trainable=True)
    grads = FlatParams(to_cfg(cfg), None, dev())
    emb = eng.forward(x.to(dev()), None, norm, save=2)
    eng.backward_params(cot.to(dev()), grads.views, accumulate=False)
    torch.cuda.synchronize()
    worst = None
    for k, v in wr.items():
        got, ref = grads.views[k].cpu(), v.grad
        if precision == "fp32":
            r = rel_max(got, ref)
            assert r < 2e-3, f"{k}: rel {r}"
        else:
            c = cos_sim(got, ref)
            assert c > 0.97, f"{k}: cos {c}"
    # accumulate=True adds on top
    eng.forward(x.to(dev()), None, norm, save=2)
    eng.backward_params(cot.to(dev()), grads.views, accumulate=True)
    k = "transformer.resblocks.0.mlp.c_fc.weight"
    ref2 = 2 * wr[k].grad
    if precision == "fp32":
        assert rel_max(grads.views[k].cpu(), ref2) < 2e-3
    else:
        assert cos_sim(grads.views[k].cpu(), ref2) > 0.97
    eng.close()


def test_adamw_kernel_vs_torch():
    l = L.load()
    g = torch.Generator().manual_seed(3)
    n = 10007
    p = torch.randn(n, generator=g); grad = torch.randn(n, generator=g) * 0.1
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-3, weight_decay=1e-2)
    dp, dm, dv = p.to(dev()), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    for step in range(1, 4):
        pt.grad = grad.clone() * step
        opt.step()
        dg = (grad * step * 2).to(dev())          # grad_scale 0.5 undoes the x2
        L.check(l.rvlm_adamw_step(dp.data_ptr(), dg.data_ptr(), dm.data_ptr(), dv.data_ptr(), n, 1e-3, 0.9, 0.999,
                                  1e-8, 1e-2, step, 0.5, L.stream_ptr()))
        torch.cuda.synchronize()
        np.testing.assert_allclose(dp.cpu().numpy(), pt.detach().numpy(), rtol=2e-5, atol=2e-7)


def test_cosine_lr():
    for s in (0, 1, 1399, 1400, 5000, 19999):
        assert cosine_lr_value(s, 1e-5, 1400, 20000) == cosine_lr_ref(s, 1e-5, 1400, 20000)


def test_train_step_fp32_matches_oracle():
    cfg = V.VIT_TINY2
    w = V.init_weights(cfg, seed=21)
    g = torch.Generator().manual_seed(4)
    B = 4
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g)
    tr = AdversarialTrainer(to_cfg(cfg), {k: v.to(dev()) for k, v in w.items()}, batch_size=B, precision="fp32",
                            lr=1e-3, wd=1e-2, warmup=2, steps=10, loss="l2", inner_loss="l2", attack="none",
                            output_normalize=False)
    ref = TrainStepRef(cfg, w, lr=1e-3, wd=1e-2, warmup=2, steps=10, loss="l2")
    with torch.no_grad():
        e0 = V.vit_forward(cfg, w, V.normalize_pixels(x))
    # attack='none' -> data_adv = data (…clip.py:334-335); perturb the inputs a little so the loss is not 0
    xa = (x + 0.02 * torch.rand(x.shape, generator=g)).clamp(0, 1)
    for it in range(3):
        out = tr.train_step(x.to(dev()), None, data_adv=xa.to(dev()))
        loss_ref, _ = ref.step(x, xa, None, e0)
        # the trainer recomputes e0 from its frozen copy (same weights as the oracle's e0)
        assert abs(float(out["loss"]) - loss_ref) <= 2e-3 * abs(loss_ref) + 1e-7, (it, float(out["loss"]), loss_ref)
    sd = tr.state_dict()
    W = cfg.width
    for k, v in ref.w.items():
        got, want = sd[k].cpu(), v.detach()
        if k.endswith("attn.in_proj_bias"):
            # softmax is invariant to the key bias: its true gradient is 0, so both sides feed pure rounding
            # noise to Adam (which normalises it to +-lr steps) - compare the q and v thirds only
            got = torch.cat([got[:W], got[2 * W:]]); want = torch.cat([want[:W], want[2 * W:]])
        r = rel_max(got, want)
        # Adam normalises the step: elements whose gradient is rounding noise move by +-lr either way
        assert r < 1e-2, f"{k}: {r}"
    tr.engine.close(); tr.engine_orig.close()


def test_train_step_bf16_runs_and_learns():
    cfg = R.CONFIGS["ViT-B-32"]
    sd = R.random_state_dict(cfg, seed=0, device=dev())
    B = 8
    tr = AdversarialTrainer(cfg, sd, batch_size=B, precision="bf16", lr=2e-6, wd=1e-4, warmup=1, steps=100,
                            attack="pgd", iterations_adv=2)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(B, 3, 224, 224, generator=g, device=dev())
    first = tr.train_step(x, None)                        # full path once: e0, fused PGD, fwd, bwd, AdamW
    assert np.isfinite(float(first["loss"])) and float(first["loss"]) > 0
    xa = (x + (torch.rand(x.shape, generator=g, device=dev()) * 2 - 1) * (4 / 255)).clamp(0, 1)
    losses = [float(tr.train_step(x, None, data_adv=xa)["loss"]) for _ in range(5)]   # fixed adversarial batch
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    tr.engine.close(); tr.engine_orig.close()
