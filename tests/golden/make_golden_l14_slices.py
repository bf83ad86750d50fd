#!/usr/bin/env python3
"""Full-length reference slices of BASELINE configs 2 and 5 on the seeded ViT-L/14.

Run once from the repo root (build container, CPU, ~15 minutes on 8 cores):

    python tests/golden/make_golden_l14_slices.py

Needs /root/reference (read-only), so it never runs on the GPU box; only its output tests/golden/l14_slices.npz
(expected outputs; the inputs are regenerated from their seeds by the test) is committed.

What is executed from the reference (nothing is copied into this repo):
  * ``train.pgd_train.pgd`` (train/pgd_train.py:5-68) - config 2: FARE PGD, 10 steps, eps = 4/255, on the first NP images
    of the batch tests/test_gpu_fullsize.py attacks (torch.rand seed 0, delta_0 seed 1);
  * ``train.apgd_train.apgd_train`` (train/apgd_train.py:125-373) - config 3: TeCoA (CE on ``emb @ (100 T)``), 10 iterations, on
    the first NP images with the labels of the test batch (randint seed 2); written to l14_slices_c3.npz
    (``python tests/golden/make_golden_l14_slices.py c3`` makes that file alone);
  * both again on the SAME images of the tower with CLIP-like weight statistics (``... clip`` -> l14_slices_clip.npz; round 6);
  * ``autoattack.autopgd_base.APGDAttack.attack_single_run`` (autoattack/autopgd_base.py:205-451) - config 5: CE loss,
    ALL 100 iterations, on the first NA images, from a recorded start point (seed 9), labels = the model's own clean
    predictions.
The model under attack is the ViT-L/14 of oracle/vit_ref.py (open-clip-torch==2.19.0 restated, pinned against HF
transformers by tests/golden/vit_hf_*.npz) with ``init_weights(seed=3)`` - the same weights the GPU tests upload.

The attacks are per-sample, so a slice attacked alone is what the same images must become
inside the B = 128 / 256 batch on the device.  torch CPU matmuls are deterministic for a fixed thread count; the
fixture records the count it was made with (a different blocking can flip the sign of a near-zero gradient
component, which the tests' identical-pixel thresholds absorb).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import vit_ref as V  # noqa: E402
from oracle import losses_ref as Lr  # noqa: E402
from train.pgd_train import pgd as ref_pgd  # noqa: E402
from train.apgd_train import apgd_train as ref_apgd_train  # noqa: E402
from autoattack.autopgd_base import APGDAttack as RefAPGDAttack  # noqa: E402

NP, NA = 8, 2
EPS, STEP = 4 / 255, 1 / 255
THREADS = int(os.environ.get("GOLDEN_THREADS", "8"))


def config3(ref, cfg, x):
    """The reference's apgd_train() with the TeCoA loss wrapper of the trainer (...clip.py:323-333)."""
    t0 = time.time()
    y = torch.randint(0, 1000, (256,), generator=torch.Generator().manual_seed(2))
    T = torch.nn.functional.normalize(torch.randn(cfg.out_dim, 1000, generator=torch.Generator().manual_seed(3)), dim=0)
    xc, yc = x[:NP].clone(), y[:NP].clone()
    wrap = Lr.ComputeLossWrapperRef(None, T, "none", "ce", 100.)
    x_adv = ref_apgd_train(ref, xc, yc, "linf", EPS, n_iter=10, loss_fn=wrap).detach()
    with torch.no_grad():
        ce = lambda xx: Lr.compute_loss_ref("ce", ref(xx, True), yc, None, 100., T, "none")     # noqa: E731
        l_clean, l_adv = ce(xc), ce(x_adv)
    print(f"apgd_train: {time.time() - t0:.0f} s, ce {l_clean.mean():.4g} -> {l_adv.mean():.4g}", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "l14_slices_c3.npz")
    np.savez_compressed(path, n=np.int64(NP), x_adv=x_adv.numpy(), y=yc.numpy(), loss_clean=l_clean.numpy(), loss_adv=l_adv.numpy(),
                        threads=np.int64(THREADS), torch_version=np.array(torch.__version__), weights_seed=np.int64(3),
                        eps=np.float64(EPS))
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


def weights_sha(w):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(w[k].numpy().tobytes())
    return h.hexdigest()


def clip_like():
    """the reference's own pgd() (config 2) and apgd_train() (config 3) on 8 images of the ViT-L/14 with
    CLIP-LIKE weight statistics (oracle/vit_ref.py::make_clip_like: outlier residual channels, LayerNorm gains over 2.5
    decades, heavy-tailed projections, peaked attention), plus - for the per-iteration bf16 gate - the gradient signs and
    per-sample losses along the trajectory of the first NS images (oracle pgd_ref with its trace, asserted bit-equal to the
    reference's result here)."""
    from oracle import attacks_ref as A
    NS = 4
    cfg = V.VIT_L_14
    t0 = time.time()
    rec = {}
    w = V.init_weights(cfg, seed=3, clip_like=True, record=rec)
    ref = V.ClipVisionModelRef(cfg, w).eval()
    print(f"clip-like weights: {time.time() - t0:.0f} s, sha256 {weights_sha(w)[:16]}", flush=True)
    x = torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    d0 = (torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1) * EPS
    y = torch.randint(0, 1000, (256,), generator=torch.Generator().manual_seed(2))
    T = torch.nn.functional.normalize(torch.randn(cfg.out_dim, 1000, generator=torch.Generator().manual_seed(3)), dim=0)
    out = dict(threads=np.int64(THREADS), torch_version=np.array(torch.__version__), weights_seed=np.int64(3),
               weights_sha256=np.array(weights_sha(w)), weights_calib=np.array(rec["calib"], dtype=np.int64),
               weights_fingerprint=V.weights_fingerprint(w), eps=np.float64(EPS), stepsize=np.float64(STEP))
    t0 = time.time()
    xc, dc = x[:NP].clone(), d0[:NP].clone()
    with torch.no_grad():
        e0 = ref(xc, False)
    wrap = Lr.ComputeLossWrapperRef(e0, None, "mean", "l2", 100.)
    x_adv = ref_pgd(ref, wrap, xc, None, "linf", EPS, 10, STEP, False, perturbation=dc.clone().requires_grad_(True),
                    mode="max").detach()
    with torch.no_grad():
        loss_end = ((ref(x_adv, False) - e0) ** 2).sum(1)
        loss_start = ((ref(xc + dc, False) - e0) ** 2).sum(1)
    print(f"pgd: {time.time() - t0:.0f} s, loss {loss_start.mean():.4g} -> {loss_end.mean():.4g}", flush=True)
    out.update(pgd_n=np.int64(NP), pgd_x_adv=x_adv.numpy(), pgd_e0=e0.numpy(), pgd_loss_end=loss_end.numpy(),
               pgd_loss_start=loss_start.numpy())
    # trajectory of the first NS images: gradient signs at every iterate (int8) and the per-sample loss there
    t0 = time.time()
    trace = []
    x_or = A.pgd_ref(ref, Lr.ComputeLossWrapperRef(e0[:NS], None, "mean", "l2", 100.), xc[:NS], None, "linf", EPS, 10, STEP,
                     False, perturbation=dc[:NS].clone(), mode="max", trace=trace)
    assert torch.equal(x_or, x_adv[:NS]), "oracle pgd_ref and the reference's pgd() must agree bit for bit on the slice"
    signs = np.stack([np.sign(t["grad"]).astype(np.int8) for t in trace])
    xn = xc[:NS].numpy().astype(np.float32)
    delta, vel = dc[:NS].numpy().astype(np.float32).copy(), np.zeros_like(xn)
    losses, gnorm = [], []
    for t in trace:
        with torch.no_grad():
            losses.append(((ref(torch.from_numpy(xn + delta), False) - e0[:NS]) ** 2).sum(1).numpy())
        gnorm.append(np.sqrt((t["grad"].astype(np.float64) ** 2).reshape(NS, -1).sum(1)))
        delta, vel = A.pgd_linf_update_ref(xn, t["grad"], delta, vel, EPS, STEP, 0.9, "max")
    assert np.array_equal(xn + delta, x_or.numpy())
    print(f"trajectory: {time.time() - t0:.0f} s, per-iteration loss {[float(l.mean()) for l in losses]}", flush=True)
    out.update(traj_n=np.int64(NS), traj_grad_sign=signs, traj_loss=np.stack(losses), traj_grad_norm=np.stack(gnorm))
    # config 3: the reference's apgd_train with the TeCoA wrapper
    t0 = time.time()
    yc = y[:NP].clone()
    wrap3 = Lr.ComputeLossWrapperRef(None, T, "none", "ce", 100.)
    x3 = ref_apgd_train(ref, xc, yc, "linf", EPS, n_iter=10, loss_fn=wrap3).detach()
    with torch.no_grad():
        ce = lambda xx: Lr.compute_loss_ref("ce", ref(xx, True), yc, None, 100., T, "none")     # noqa: E731
        l_clean, l_adv = ce(xc), ce(x3)
    print(f"apgd_train: {time.time() - t0:.0f} s, ce {l_clean.mean():.4g} -> {l_adv.mean():.4g}", flush=True)
    out.update(c3_x_adv=x3.numpy(), c3_y=yc.numpy(), c3_loss_clean=l_clean.numpy(), c3_loss_adv=l_adv.numpy())
    path = os.path.join(ROOT, "tests", "golden", "l14_slices_clip.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


def main():
    torch.set_num_threads(THREADS)
    if len(sys.argv) > 1 and sys.argv[1] == "clip":
        clip_like()
        return
    cfg = V.VIT_L_14
    w = V.init_weights(cfg, seed=3)
    ref = V.ClipVisionModelRef(cfg, w).eval()
    if len(sys.argv) > 1 and sys.argv[1] == "c3":
        config3(ref, cfg, torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(0)))
        return
    # the batch of tests/test_gpu_fullsize.py::setup (the whole 256-image tensors are drawn, then sliced)
    x = torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    d0 = (torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1) * EPS
    T = torch.nn.functional.normalize(torch.randn(cfg.out_dim, 1000, generator=torch.Generator().manual_seed(3)), dim=0)
    out = dict(threads=np.int64(THREADS), torch_version=np.array(torch.__version__), weights_seed=np.int64(3),
               eps=np.float64(EPS), stepsize=np.float64(STEP))

    # ---- config 2: the reference's pgd() -----------------------------------------------------------------------
    t0 = time.time()
    xc, dc = x[:NP].clone(), d0[:NP].clone()
    with torch.no_grad():
        e0 = ref(xc, False)
    wrap = Lr.ComputeLossWrapperRef(e0, None, "mean", "l2", 100.)
    x_adv = ref_pgd(ref, wrap, xc, None, "linf", EPS, 10, STEP, False, perturbation=dc.clone().requires_grad_(True),
                    mode="max").detach()
    with torch.no_grad():
        loss_end = ((ref(x_adv, False) - e0) ** 2).sum(1)
        loss_start = ((ref(xc + dc, False) - e0) ** 2).sum(1)
    print(f"pgd: {time.time() - t0:.0f} s, loss {loss_start.mean():.4g} -> {loss_end.mean():.4g}", flush=True)
    out.update(pgd_n=np.int64(NP), pgd_x_adv=x_adv.numpy(), pgd_e0=e0.numpy(), pgd_loss_end=loss_end.numpy(),
               pgd_loss_start=loss_start.numpy())

    # ---- config 5: the reference's APGDAttack, all 100 iterations ------------------------------------------------
    t0 = time.time()
    clf = V.ClassificationModelRef(cfg, w, T).eval()
    xa = x[:NA].clone()
    with torch.no_grad():
        logits = clf(xa)
        ya = logits.argmax(1)
    start = (xa + EPS * (2 * torch.rand(xa.shape, generator=torch.Generator().manual_seed(9)) - 1)).clamp(0, 1)
    atk = RefAPGDAttack(clf, n_iter=100, norm="Linf", n_restarts=1, eps=EPS, seed=0, loss="ce", device="cpu")
    atk.init_hyperparam(xa)
    x_best, acc, loss_best, x_best_adv = atk.attack_single_run(xa, ya, x_init=start)
    print(f"apgd-ce 100: {time.time() - t0:.0f} s, loss_best {loss_best.tolist()}, acc {acc.tolist()}", flush=True)
    top2 = logits.topk(2, dim=1).values
    out.update(apgd_n=np.int64(NA), apgd_y=ya.numpy(), apgd_clean_margin=(top2[:, 0] - top2[:, 1]).numpy(),
               apgd_x_best=x_best.detach().numpy(), apgd_loss_best=loss_best.detach().numpy(),
               apgd_acc=acc.numpy(), apgd_x_best_adv=x_best_adv.detach().numpy())
    path = os.path.join(ROOT, "tests", "golden", "l14_slices.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")
    config3(ref, cfg, x)


if __name__ == "__main__":
    main()
