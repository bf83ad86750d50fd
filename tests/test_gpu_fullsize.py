"""BASELINE.json configs 2, 3 and 5 at their FULL sizes on the GPU (ViT-L/14, bf16, B = 128 / 128 / 256), checked

* through size-independent properties of the attack (eps-ball, image range, determinism, loss goes up, robust
  accuracy can only go down, sharding invariance), and
* against the CPU oracle (oracle/attacks_ref.py over oracle/vit_ref.py, fp32) on a 4-image slice of the SAME batch:
  the attack is per-sample, so the slice's result inside the full batch must agree with the oracle's run of the slice
  alone.  Two different encoders (bf16 MFMA vs fp32 CPU) flip the sign of near-zero gradient components (SURVEY
  Appendix D.12), so the layered bar is: identical-pixel fraction, final per-sample loss within tolerance; the
  fp32 mode of the engine, run on the same slice, has to agree with the oracle much more tightly.

Since round 5 the FULL-LENGTH slices of configs 2, 3 and 5 come from tests/golden/l14_slices[_c3].npz: the reference's
own ``pgd`` (8 images, 10 steps), ``apgd_train`` (8 images, 10 iterations) and ``APGDAttack.attack_single_run`` (2 images,
all 100 iterations) run on the seeded ViT-L/14 in the build container (tests/golden/make_golden_l14_slices.py) - nothing
is attacked on the CPU of the GPU box for them any more (the per-iteration trajectory test still runs the oracle there).

One ViT-L/14 engine pair (bf16 max_batch 256, fp32 max_batch 8) is shared by the module; the oracle runs with 32
host threads and costs ~20 s per attack of 4 images (the trajectory test).
"""
import os
import time

import numpy as np
import pytest
import torch

import robustvlm_amd as R
from oracle import vit_ref as V
from oracle import attacks_ref as A
from oracle import losses_ref as Lr
from tests.gpu_helpers import dev, record

pytestmark = pytest.mark.gpu

EPS, STEP = 4 / 255, 1 / 255
EPS_F = float(np.float32(EPS))
NS = 4          # images of the on-box oracle slices (config 3, trajectory)
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "l14_slices.npz"))
NP = int(GOLD["pgd_n"])      # images of the reference's full-length pgd() slice
GOLD3 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "l14_slices_c3.npz"))


def to_cfg(c):
    return R.VitConfig(c.image_size, c.patch, c.width, c.layers, c.heads, c.out_dim, c.act)


@pytest.fixture(scope="module")
def setup():
    torch.set_num_threads(32)
    cfg = V.VIT_L_14
    w = V.init_weights(cfg, seed=3)
    wd = {k: v.to(dev()) for k, v in w.items()}
    eng = R.VitEngine(to_cfg(cfg), wd, precision="bf16", max_batch=256)
    eng32 = R.VitEngine(to_cfg(cfg), wd, precision="fp32", max_batch=NP)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(256, 3, 224, 224, generator=g)                       # BASELINE: torch.rand images, seed 0
    d0 = (torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1) * EPS
    y = torch.randint(0, 1000, (256,), generator=torch.Generator().manual_seed(2))
    T = torch.nn.functional.normalize(torch.randn(cfg.out_dim, 1000, generator=torch.Generator().manual_seed(3)), dim=0)
    ref = V.ClipVisionModelRef(cfg, w).eval()
    yield dict(cfg=cfg, w=w, eng=eng, eng32=eng32, x=x, d0=d0, y=y, T=T, ref=ref)
    eng.close(); eng32.close()
    torch.set_num_threads(8)


def ball_and_range(x_adv, x):
    d = x_adv - x
    # (x + delta) - x is evaluated in fp32: one ulp of the O(1) pixel values on top of float32(eps)
    assert float(d.abs().max()) <= EPS_F + 1.2e-7, float(d.abs().max())
    assert float(x_adv.min()) >= 0.0 and float(x_adv.max()) <= 1.0
    return d


def test_config2_fare_pgd_b128(setup):
    """configs[1]: FARE PGD 10-step eps=4/255 on ViT-L/14 bf16, batch 128 - the shape bench.py times."""
    s = setup
    B = 128
    x, d0 = s["x"][:B].to(dev()), s["d0"][:B].to(dev())
    model = R.ClipVisionModel(s["eng"]).eval()
    e0 = model(x, False)
    wrap = R.ComputeLossWrapper(e0, None, "mean", "l2", 100.)
    run = lambda: R.pgd(model, wrap, x, None, "linf", EPS, 10, STEP, False, perturbation=d0.clone(), mode="max")  # noqa: E731
    xa = run()
    assert torch.equal(xa, run()), "not deterministic"
    d = ball_and_range(xa, x)
    assert float((d.abs() >= EPS_F * 0.999).float().mean()) > 0.3          # the attack uses its budget
    with torch.no_grad():
        l_start = ((model(x + d0, False) - e0) ** 2).sum(1)
        l_end = ((model(xa, False) - e0) ** 2).sum(1)
    assert bool((l_end > l_start).all()) and float(l_end.mean()) > 2 * float(l_start.mean())
    # sharding invariance (what the data-parallel path relies on): the first 16 images attacked alone
    h = 16
    xs = R.pgd(model, R.ComputeLossWrapper(e0[:h], None, "mean", "l2", 100.), x[:h], None, "linf", EPS, 10, STEP, False,
               perturbation=d0[:h].clone(), mode="max")
    same_shard = float((xs == xa[:h]).float().mean())
    # the reference's own pgd() on the first NP images (tests/golden/l14_slices.npz: fp32 CPU, run in the build container),
    # and the engine's fp32 mode on the same slice
    x_or = torch.from_numpy(GOLD["pgd_x_adv"])
    e0c = torch.from_numpy(GOLD["pgd_e0"])
    l_or = torch.from_numpy(GOLD["pgd_loss_end"])
    m32 = R.ClipVisionModel(s["eng32"]).eval()
    e0_32 = m32(x[:NP], False)
    assert rel(e0_32.cpu(), e0c) < 1e-4                                   # fp32 embeddings within 1e-4 relative (north_star)
    x32 = R.pgd(m32, R.ComputeLossWrapper(e0_32, None, "mean", "l2", 100.), x[:NP], None, "linf", EPS, 10, STEP, False,
                perturbation=d0[:NP].clone(), mode="max").cpu()
    same_bf16 = float((xa[:NP].cpu() == x_or).float().mean())
    same_fp32 = float((x32 == x_or).float().mean())
    with torch.no_grad():
        l_bf = ((s["ref"](xa[:NS].cpu(), False) - e0c[:NS]) ** 2).sum(1)   # judged by the oracle encoder, like the fixture
    loss_ratio = float((l_bf / l_or[:NS]).mean())
    record("config2_fare_pgd_b128", same_pixels_bf16_vs_reference=same_bf16, same_pixels_fp32_vs_reference=same_fp32,
           same_pixels_shard16_vs_b128=same_shard, loss_ratio_bf16_over_reference=loss_ratio,
           loss_end_over_start=float(l_end.mean()) / float(l_start.mean()), slice_images=NP)
    # measured (profiles/r02_parity_metrics.jsonl, 4-image oracle slice): shard 0.986, fp32 0.9969, bf16 0.743 (ten
    # iterations compound the sign flips of near-zero gradient components), loss ratio 0.998
    assert same_shard > 0.97, same_shard
    assert same_fp32 > 0.99, same_fp32
    assert same_bf16 > 0.70, same_bf16
    assert 0.97 < loss_ratio < 1.03, loss_ratio


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def test_config2_mixed_precision_first_iteration_fp32(setup):
    """precision='bf16+fp32-first' (VERDICT r4 item 4): the clean embedding and the FIRST iteration in the reference's
    own precision (fp32, on the matrix pipe), iterations 2..10 in bf16 - against the reference's pgd() on the same 8
    images.  The bf16 path loses its first step to rounding noise (FARE's first cotangent is a difference of nearly
    equal embeddings: sign agreement 0.82 at iteration 0); with one fp32 iteration the trajectory stays with the
    reference's far longer."""
    s = setup
    wd = {k: v.to(dev()) for k, v in s["w"].items()}
    eng = R.VitEngine(to_cfg(s["cfg"]), wd, precision="bf16+fp32-first", max_batch=NP)
    try:
        model = R.ClipVisionModel(eng).eval()
        x, d0 = s["x"][:NP].to(dev()), s["d0"][:NP].to(dev())
        e0 = model(x, False)                                               # gradient-free forward: the fp32 handle
        assert rel(e0.cpu(), torch.from_numpy(GOLD["pgd_e0"])) < 1e-4
        run = lambda: R.pgd(model, R.ComputeLossWrapper(e0, None, "mean", "l2", 100.), x, None, "linf", EPS, 10, STEP,   # noqa: E731
                            False, perturbation=d0.clone(), mode="max")
        xa = run()
        assert torch.equal(xa, run()), "not deterministic"
        ball_and_range(xa, x)
        x_or = torch.from_numpy(GOLD["pgd_x_adv"])
        same_mixed = float((xa.cpu() == x_or).float().mean())
        # the plain bf16 engine on the same 8 images, for the comparison
        model16 = R.ClipVisionModel(s["eng"]).eval()
        e16 = model16(x, False)
        x16 = R.pgd(model16, R.ComputeLossWrapper(e16, None, "mean", "l2", 100.), x, None, "linf", EPS, 10, STEP, False,
                    perturbation=d0.clone(), mode="max")
        same_bf16 = float((x16.cpu() == x_or).float().mean())
        # first-iteration gradient signs: one iteration of each engine from the same start
        one = lambda m, e: R.pgd(m, R.ComputeLossWrapper(e, None, "mean", "l2", 100.), x, None, "linf", EPS, 1, STEP,   # noqa: E731
                                 False, perturbation=d0.clone(), mode="max")
        x1_mixed, x1_bf16 = one(model, e0), one(model16, e16)
        x1_32 = one(R.ClipVisionModel(s["eng32"]).eval(), s["eng32"].forward(x, None, False, save=False))
        first_mixed = float((x1_mixed == x1_32).float().mean())
        first_bf16 = float((x1_bf16 == x1_32).float().mean())
        record("config2_mixed_precision", same_pixels_mixed_vs_reference=same_mixed, same_pixels_bf16_vs_reference=same_bf16,
               first_step_same_pixels_mixed_vs_fp32=first_mixed, first_step_same_pixels_bf16_vs_fp32=first_bf16)
        assert first_mixed == 1.0, first_mixed                 # the first iteration IS the fp32 engine's
        assert same_mixed > same_bf16 + 0.05, (same_mixed, same_bf16)
    finally:
        eng.close()


def test_config3_tecoa_apgd_b128(setup):
    """configs[2]: TeCoA (supervised CE) apgd_train 10-step on ViT-L/14, batch 128."""
    s = setup
    B = 128
    x, y, T = s["x"][:B].to(dev()), s["y"][:B].to(dev()), s["T"].to(dev())
    model = R.ClipVisionModel(s["eng"]).eval()
    wrap = R.ComputeLossWrapper(None, T, "none", "ce", 100.)
    run = lambda: R.apgd_train(model, x, y, "linf", EPS, n_iter=10, loss_fn=wrap)      # noqa: E731
    xa = run()
    assert torch.equal(xa, run()), "not deterministic"
    ball_and_range(xa, x)
    with torch.no_grad():
        l_clean = R.compute_loss("ce", model(x, True), y, None, 100., T, "none")
        l_adv = R.compute_loss("ce", model(xa, True), y, None, 100., T, "none")
    assert float(l_adv.mean()) > float(l_clean.mean())
    # the reference's own apgd_train() on the first 8 images (tests/golden/l14_slices_c3.npz, run in the build container)
    n3 = int(GOLD3["n"])
    yc = s["y"][:n3]
    assert torch.equal(yc, torch.from_numpy(GOLD3["y"]))
    x_or = torch.from_numpy(GOLD3["x_adv"])
    m32 = R.ClipVisionModel(s["eng32"]).eval()
    x32 = R.apgd_train(m32, x[:n3], y[:n3], "linf", EPS, n_iter=10, loss_fn=wrap).cpu()
    same_bf16 = float((xa[:n3].cpu() == x_or).float().mean())
    same_fp32 = float((x32 == x_or).float().mean())
    with torch.no_grad():
        ce = lambda xx: Lr.compute_loss_ref("ce", s["ref"](xx, True), yc[:NS], None, 100., s["T"], "none")   # noqa: E731
        loss_ratio = float((ce(xa[:NS].cpu()) / torch.from_numpy(GOLD3["loss_adv"])[:NS]).mean())     # judged by the oracle encoder
    record("config3_tecoa_apgd_b128", same_pixels_bf16_vs_reference=same_bf16, same_pixels_fp32_vs_reference=same_fp32,
           loss_ratio_bf16_over_reference=loss_ratio, loss_adv_over_clean=float(l_adv.mean()) / float(l_clean.mean()),
           slice_images=n3)
    # measured (4-image oracle slice of rounds 2-4): fp32 0.9999, bf16 0.9795-0.983, loss ratio 0.9999
    assert same_fp32 > 0.999, same_fp32
    assert same_bf16 > 0.95, same_bf16
    assert 0.98 < loss_ratio < 1.02, loss_ratio


def test_config5_apgd_ce_100_b256(setup):
    """configs[4]: APGDAttack CE, 100 iterations, ViT-L/14 + zero-shot head with 1000 classes, batch 256; the slice
    checked element by element is the reference's own 100-iteration run on 2 images (committed fixture)."""
    s = setup
    B = 256
    x, y, T = s["x"].to(dev()), s["y"].to(dev()), s["T"].to(dev())
    clf = R.ClassificationModel(s["eng"], T).eval()
    with torch.no_grad():
        # labels = the model's own clean predictions (random-init weights know no ImageNet): every image starts
        # "robust", so all 256 go through the 100 iterations, as in the timed configuration
        y = clf(x).argmax(1)
    att = R.APGDAttack(clf, n_iter=100, norm="Linf", n_restarts=1, eps=EPS, seed=0, loss="ce", device=dev())
    xa = att.perturb(x, y)
    ball_and_range(xa, x)
    with torch.no_grad():
        acc_clean = (clf(x).argmax(1) == y).float().mean().item()
        acc_adv = (clf(xa).argmax(1) == y).float().mean().item()
        l_clean = R.ce(clf(x), y, "none")
        l_adv = R.ce(clf(xa), y, "none")
    assert acc_clean == 1.0 and acc_adv <= acc_clean
    # perturb() returns x for the points that stay robust and an adversarial point otherwise: the loss never drops
    assert bool((l_adv >= l_clean - 1e-3).all())
    att2 = R.APGDAttack(clf, n_iter=100, norm="Linf", n_restarts=1, eps=EPS, seed=0, loss="ce", device=dev())
    assert torch.equal(att2.perturb(x[:32], y[:32]), R.APGDAttack(clf, n_iter=100, norm="Linf", n_restarts=1, eps=EPS, seed=0,
                                                                  loss="ce", device=dev()).perturb(x[:32], y[:32]))
    # the reference's APGDAttack.attack_single_run over ALL 100 iterations on the first 2 images (tests/golden/l14_slices.npz),
    # the SAME start point on both sides; labels = the reference model's clean predictions, which must be the bf16 engine's
    n = int(GOLD["apgd_n"])
    yc = torch.from_numpy(GOLD["apgd_y"])
    assert torch.equal(y[:n].cpu(), yc), (y[:n].cpu(), yc, GOLD["apgd_clean_margin"])
    xc = s["x"][:n]
    start = (xc + EPS * (2 * torch.rand(xc.shape, generator=torch.Generator().manual_seed(9)) - 1)).clamp(0, 1)
    g = R.APGDAttack(clf, n_iter=100, norm="Linf", n_restarts=1, eps=EPS, seed=0, loss="ce", device=dev())
    xb, acc_b, lb, xba = g.attack_single_run(x[:n], y[:n], x_init=start.to(dev()))
    xb_or, lb_or = torch.from_numpy(GOLD["apgd_x_best"]), torch.from_numpy(GOLD["apgd_loss_best"])
    same = float((xb.cpu() == xb_or).float().mean())
    loss_ratio = float((lb.cpu() / lb_or).mean())
    same_acc = bool((acc_b.cpu().bool() == torch.from_numpy(GOLD["apgd_acc"]).bool()).all())
    record("config5_apgd_ce_100_b256", acc_clean=acc_clean, acc_adv=acc_adv, same_pixels_bf16_vs_reference_100it=same,
           loss_best_ratio_bf16_over_reference_100it=loss_ratio, same_acc_flags_100it=same_acc,
           loss_adv_over_clean=float(l_adv.mean()) / float(l_clean.mean()))
    # (20-iteration oracle slice of rounds 2-4: identical pixels 0.990, loss_best ratio 0.99995)
    assert same > 0.95, same
    assert 0.98 < loss_ratio < 1.02, loss_ratio
    assert same_acc


def test_l2_threat_model_at_full_size(setup):
    """The L2 entry points (pgd / apgd_train / APGDAttack with norm = l2: the --norm l2 of the reference's trainer and of
    clip_robustbench.py) on ViT-L/14 bf16 at B = 32: size-independent properties - the L2 ball, the image range, determinism,
    loss up / accuracy not up, and the per-sample property the data-parallel path relies on (a batch run as two halves
    lands on the same points up to the encoder's batch-shape dependence)."""
    s = setup
    B, eps = 32, 3.0          # (a usual ImageNet L2 radius; the L-inf ball of 4/255 reaches |delta|_2 = 6.1 on 224 x 224 x 3)
    x, d0 = s["x"][:B].to(dev()), s["d0"][:B].to(dev())
    y, T = s["y"][:B].to(dev()), s["T"].to(dev())
    model = R.ClipVisionModel(s["eng"]).eval()

    def in_ball(xa):
        assert float((xa - x).flatten(1).norm(dim=1).max()) <= eps * (1 + 1e-5)
        assert float(xa.min()) >= 0.0 and float(xa.max()) <= 1.0
    with torch.no_grad():
        e0 = model(x, False)
    wrap = R.ComputeLossWrapper(e0, None, "mean", "l2", 100.)
    run = lambda lo, hi, w: R.pgd(model, w, x[lo:hi], None, "l2", eps, 10, eps / 4, False, perturbation=d0[lo:hi], mode="max")   # noqa: E731
    xa = run(0, B, wrap)
    in_ball(xa)
    assert torch.equal(xa, run(0, B, wrap))
    with torch.no_grad():
        l_start = ((model(x + d0, False) - e0) ** 2).sum(1)
        l_end = ((model(xa, False) - e0) ** 2).sum(1)
    assert bool((l_end > l_start).all())
    halves = torch.cat([run(0, B // 2, R.ComputeLossWrapper(e0[:B // 2], None, "mean", "l2", 100.)),
                        run(B // 2, B, R.ComputeLossWrapper(e0[B // 2:], None, "mean", "l2", 100.))])
    # images whose token rows sit in full 256-row GEMM tiles in both runs come out bit-identical; the one or two whose rows fall into
    # the remainder-row phase (a different K split) start from a different bf16 rounding of FARE's noise-limited first gradient
    rel_each = (halves - xa).flatten(1).norm(dim=1) / eps
    rel_pgd, same_pgd = float(rel_each.max()), float((rel_each < 1e-3).float().mean())
    # TeCoA apgd_train(norm='l2')
    wce = R.ComputeLossWrapper(None, T, "none", "ce", 100.)
    xt = R.apgd_train(model, x, y, "l2", eps, n_iter=10, loss_fn=wce)
    in_ball(xt)
    assert torch.equal(xt, R.apgd_train(model, x, y, "l2", eps, n_iter=10, loss_fn=wce))
    with torch.no_grad():
        assert float(wce(model(xt, True), y).mean()) > float(wce(model(x, True), y).mean())
    # APGDAttack(norm='L2') on the zero-shot head, labels = the clean predictions
    clf = R.ClassificationModel(s["eng"], T).eval()
    with torch.no_grad():
        yc = clf(x).argmax(1)
    kw = dict(n_iter=20, norm="L2", n_restarts=1, eps=eps, seed=0, loss="ce", device=dev())
    xe = R.APGDAttack(clf, **kw).perturb(x, yc)
    in_ball(xe)
    assert torch.equal(xe, R.APGDAttack(clf, **kw).perturb(x, yc))
    with torch.no_grad():
        acc_adv = float((clf(xe).argmax(1) == yc).float().mean())
        assert bool((R.ce(clf(xe), yc, "none") >= R.ce(clf(x), yc, "none") - 1e-3).all())
    record("l2_threat_model_b32", pgd_halves_vs_whole_rel_l2_max=rel_pgd, pgd_halves_vs_whole_images_equal=same_pgd,
           apgdattack_l2_acc_adv=acc_adv,
           pgd_loss_end_over_start=float(l_end.mean() / l_start.mean()))
    assert same_pgd >= 0.8 and rel_pgd < 1.0, (same_pgd, rel_pgd)
    assert acc_adv < 1.0


def _oracle_pgd_trajectory(ref, xc, dc, e0c, iterations=10):
    """oracle pgd_ref on a slice with its per-iteration gradients, and the iterates delta_0 .. delta_{I-1} it evaluated
    them at (replayed from the traced gradients with the oracle's own update)."""
    trace = []
    x_or = A.pgd_ref(ref, Lr.ComputeLossWrapperRef(e0c, None, "mean", "l2", 100.), xc, None, "linf", EPS, iterations, STEP,
                     False, perturbation=dc.clone(), mode="max", trace=trace)
    xn = xc.numpy().astype(np.float32)
    delta, vel = dc.numpy().astype(np.float32).copy(), np.zeros_like(xn)
    deltas = []
    for t in trace:
        deltas.append(delta.copy())
        delta, vel = A.pgd_linf_update_ref(xn, t["grad"], delta, vel, EPS, STEP, 0.9, "max")
    assert np.array_equal(xn + delta, x_or.numpy()), "trajectory replay must reproduce the oracle's result"
    return x_or, trace, deltas


def test_config2_gradient_signs_along_the_oracle_trajectory_b128(setup):
    """VERDICT r2 weak 1(b,c): the ENGINE AS A WHOLE under the B = 128 production dispatch (M = 32 896: persistent GEMM +
    strip phase, fused attention, class-token tail), not only its GEMM shapes one by one.  The oracle attacks the first
    NS images; at each of its iterates delta_k the HIP path evaluates forward + loss + input gradient of the FULL batch
    (images NS.. carry their own delta_0) and the slice's embeddings / gradients are compared with the oracle's at the
    same point.  sign(g) is all the L-inf update uses, so per-iteration sign agreement is the bf16 gate that the
    end-to-end identical-pixel fraction (ten compounding iterations) cannot be."""
    s = setup
    B = 128
    x, d0 = s["x"][:B].to(dev()), s["d0"][:B].to(dev())
    xc, dc = s["x"][:NS], s["d0"][:NS]
    with torch.no_grad():
        e0c = s["ref"](xc, False)
    _, trace, deltas = _oracle_pgd_trajectory(s["ref"], xc, dc, e0c)
    eng = s["eng"]
    e0 = eng.forward(x, None, False, save=False)
    assert cos_sim_rows(e0[:NS].cpu(), e0c) > 0.9999
    signs, coss, losses = [], [], []
    for k in (0, 1, 2, 4, 9):
        d = d0.clone()
        d[:NS] = torch.from_numpy(deltas[k]).to(dev())
        emb, per, _, g = eng.fwd_inputgrad(x, d, "l2", "mean", e0, None, False)
        torch.cuda.synchronize()
        gk, gr = g[:NS].cpu().numpy(), trace[k]["grad"]
        signs.append(float(np.mean(np.sign(gk) == np.sign(gr))))
        coss.append(float((gk.astype(np.float64) * gr).sum() / (np.linalg.norm(gk.astype(np.float64)) * np.linalg.norm(gr.astype(np.float64)))))
        with torch.no_grad():
            per_or = ((s["ref"](xc + torch.from_numpy(deltas[k]), False) - e0c) ** 2).sum(1)
        losses.append(float((per[:NS].cpu() / per_or).mean()))
    record("config2_gradient_signs_along_the_oracle_trajectory_b128",
           **{f"sign_agree_it{k}": v for k, v in zip((0, 1, 2, 4, 9), signs)},
           **{f"grad_cos_it{k}": v for k, v in zip((0, 1, 2, 4, 9), coss)},
           **{f"loss_ratio_it{k}": v for k, v in zip((0, 1, 2, 4, 9), losses)})
    # Measured (profiles/r03_parity_metrics.jsonl): sign agreement 0.820 / 0.980 / 0.990 / 0.994 / 0.996 at iterations
    # 0 / 1 / 2 / 4 / 9, gradient cosine 0.846 / 0.998 / 0.9995 / 0.9998 / 0.9999.  ITERATION 0 IS NOISE-LIMITED IN bf16, by
    # the loss and not by the kernels: FARE's loss is ||phi(x + d) - phi(x)||^2 and at the random start the two embeddings
    # differ by ~1e-2 of their norm (the loss grows 1 100-fold over the ten iterations), which is only ~3x the 2e-3
    # relative error of ANY bf16 encoder - the cotangent 2 (phi(x + d) - phi(x)) is then one part rounding noise in three.
    # One iteration later the difference has grown tenfold and the bf16 gradient agrees with the fp32 oracle's to 0.998.
    # The fp32 mode of the engine on the same slice (below) is the tight check of the path at iteration 0.
    assert signs[0] >= 0.78 and coss[0] > 0.80, (signs, coss)
    assert signs[1] >= 0.97 and coss[1] > 0.995, (signs, coss)
    assert min(signs[2:]) >= 0.985 and min(coss[2:]) > 0.999, (signs, coss)
    assert 0.9 < losses[0] < 1.8 and all(0.985 < r < 1.01 for r in losses[1:]), losses
    e0_32 = s["eng32"].forward(x[:NS], None, False, save=False)
    _, _, _, g32 = s["eng32"].fwd_inputgrad(x[:NS], d0[:NS], "l2", "mean", e0_32, None, False)
    sign32 = float(np.mean(np.sign(g32.cpu().numpy()) == np.sign(trace[0]["grad"])))
    record("config2_gradient_signs_along_the_oracle_trajectory_b128", sign_agree_it0_fp32_mode=sign32)
    assert sign32 > 0.999, sign32


def cos_sim_rows(a, b):
    a, b = a.double(), b.double()
    return float(((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))).min())


def test_config1_fare_pgd_b32_b8():
    """configs[0] (the reference's own CPU-runnable case) THROUGH THE HIP PATH as an attack: FARE PGD 10-step eps=4/255
    on ViT-B/32, batch 8, torch.rand images (seed 0), delta0 ~ U(-eps, eps) (seed 1) - exactly the problem bench.py's
    cpu_baseline times - against oracle pgd_ref (train/pgd_train.py:5-68).  The engine's fp32 mode is the tight check
    (identical pixels >= 0.99, final loss within 1 %); the bf16 mode is held to the layered bar."""
    torch.set_num_threads(32)
    cfg = V.VIT_B_32
    w = V.init_weights(cfg, seed=0)
    ref = V.ClipVisionModelRef(cfg, w).eval()
    B = 8
    x = torch.rand(B, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    d0 = torch.zeros_like(x).uniform_(-EPS, EPS, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        e0c = ref(x, False)
    x_or, trace, _ = _oracle_pgd_trajectory(ref, x, d0, e0c)
    with torch.no_grad():
        l_or = ((ref(x_or, False) - e0c) ** 2).sum(1)
    wd = {k: v.to(dev()) for k, v in w.items()}
    out = {}
    for prec in ("fp32", "bf16"):
        eng = R.VitEngine(to_cfg(cfg), wd, precision=prec, max_batch=B)
        model = R.ClipVisionModel(eng).eval()
        xd = x.to(dev())
        e0 = model(xd, False)
        wrap = R.ComputeLossWrapper(e0, None, "mean", "l2", 100.)
        xa = R.pgd(model, wrap, xd, None, "linf", EPS, 10, STEP, False, perturbation=d0.to(dev()), mode="max")
        assert torch.equal(xa, R.pgd(model, wrap, xd, None, "linf", EPS, 10, STEP, False, perturbation=d0.to(dev()), mode="max"))
        ball_and_range(xa, xd)
        # first-iteration gradient at the oracle's own start point
        _, _, _, g0 = eng.fwd_inputgrad(xd, d0.to(dev()), "l2", "mean", e0, None, False)
        sign0 = float(np.mean(np.sign(g0.cpu().numpy()) == np.sign(trace[0]["grad"])))
        with torch.no_grad():
            l_hip = ((ref(xa.cpu(), False) - e0c) ** 2).sum(1)         # both judged by the oracle encoder
        out[prec] = dict(same=float((xa.cpu() == x_or).float().mean()), loss_ratio=float((l_hip / l_or).mean()),
                         sign0=sign0, emb_rel=float((e0.cpu() - e0c).abs().max() / e0c.abs().max()))
        eng.close()
    record("config1_fare_pgd_b32_b8", **{f"{k}_{p}": v for p, d in out.items() for k, v in d.items()})
    torch.set_num_threads(8)
    assert out["fp32"]["emb_rel"] < 1e-4, out                      # north_star: fp32 embeddings within 1e-4 relative
    assert out["fp32"]["same"] >= 0.99 and 0.99 < out["fp32"]["loss_ratio"] < 1.01, out
    assert out["fp32"]["sign0"] > 0.999, out
    # bf16 (measured: first-iteration sign agreement 0.897, identical pixels 0.786, loss ratio 0.991): the first iteration of
    # FARE is noise-limited in bf16 (see test_config2_gradient_signs_along_the_oracle_trajectory_b128), the end result is not
    assert out["bf16"]["sign0"] > 0.85 and out["bf16"]["same"] > 0.70 and 0.97 < out["bf16"]["loss_ratio"] < 1.03, out


# ---- CLIP-LIKE WEIGHT STATISTICS -----------------------------------------------------------------------
# Every full-size number above is on i.i.d. Gaussian weights with LayerNorm gains 1 +- 0.1.  The towers the reference fine-tunes
# (train/adversarial_training_clip.py:95-103: OpenAI ViT-L/14) have outlier residual channels ("massive activations", 30-100x
# the ordinary channels), LayerNorm gains over two and a half decades and heavy-tailed projections with peaked attention.
# oracle/vit_ref.py::make_clip_like imposes those statistics on the seeded draw; tests/golden/l14_slices_clip.npz holds the
# REFERENCE's own pgd() / apgd_train() on 8 images of that tower (build container) and, for the first 4 images, the gradient
# signs and per-sample losses at every iterate of the reference's trajectory.
GOLDC = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "l14_slices_clip.npz"))


@pytest.fixture(scope="module")
def setup_clip():
    import hashlib
    torch.set_num_threads(32)
    cfg = V.VIT_L_14
    rec = {}
    w = V.init_weights(cfg, seed=3, clip_like=True, record=rec)
    # the same weights as the build container's (where the reference ran): the calibration took the same decisions (exact) and
    # every tensor has the same sums to 1e-6; bit equality of all 304 M floats across hosts is recorded, not demanded
    assert np.array_equal(np.array(rec["calib"]), GOLDC["weights_calib"]), "make_clip_like calibrated differently on this host"
    fp, fp0 = V.weights_fingerprint(w), GOLDC["weights_fingerprint"]
    assert np.allclose(fp, fp0, rtol=1e-6, atol=1e-6 * np.abs(fp0[:, 1:]).max()), np.abs(fp - fp0).max()
    sha = hashlib.sha256()
    for k in sorted(w):
        sha.update(w[k].numpy().tobytes())
    record("clip_like_weights", bit_identical_to_the_build_container=float(sha.hexdigest() == str(GOLDC["weights_sha256"])))
    wd = {k: v.to(dev()) for k, v in w.items()}
    n = int(GOLDC["pgd_n"])
    eng = R.VitEngine(to_cfg(cfg), wd, precision="bf16", max_batch=128)
    eng32 = R.VitEngine(to_cfg(cfg), wd, precision="fp32", max_batch=n)
    x = torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(0))[:128]
    d0 = ((torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1) * EPS)[:128]
    y = torch.randint(0, 1000, (256,), generator=torch.Generator().manual_seed(2))[:128]
    T = torch.nn.functional.normalize(torch.randn(cfg.out_dim, 1000, generator=torch.Generator().manual_seed(3)), dim=0)
    ref = V.ClipVisionModelRef(cfg, w).eval()
    yield dict(cfg=cfg, eng=eng, eng32=eng32, x=x, d0=d0, y=y, T=T, ref=ref, n=n)
    eng.close(); eng32.close()
    torch.set_num_threads(8)


def test_clip_like_config2_vs_reference(setup_clip):
    """configs[1] on the CLIP-like tower, B = 128 under the production dispatch, against the reference's own pgd() on the first 8
    images: fp32 mode identical pixels > 0.99 and embeddings within 1e-4; bf16 end to end held to the layered bar."""
    s = setup_clip
    n = s["n"]
    x, d0 = s["x"].to(dev()), s["d0"].to(dev())
    model = R.ClipVisionModel(s["eng"]).eval()
    e0 = model(x, False)
    xa = R.pgd(model, R.ComputeLossWrapper(e0, None, "mean", "l2", 100.), x, None, "linf", EPS, 10, STEP, False,
               perturbation=d0.clone(), mode="max")
    ball_and_range(xa, x)
    x_or, e0c, l_or = (torch.from_numpy(GOLDC[k]) for k in ("pgd_x_adv", "pgd_e0", "pgd_loss_end"))
    m32 = R.ClipVisionModel(s["eng32"]).eval()
    e0_32 = m32(x[:n], False)
    emb_rel32 = rel(e0_32.cpu(), e0c)
    emb_rel16 = rel(e0[:n].cpu(), e0c)
    x32 = R.pgd(m32, R.ComputeLossWrapper(e0_32, None, "mean", "l2", 100.), x[:n], None, "linf", EPS, 10, STEP, False,
                perturbation=d0[:n].clone(), mode="max").cpu()
    same_fp32 = float((x32 == x_or).float().mean())
    same_bf16 = float((xa[:n].cpu() == x_or).float().mean())
    with torch.no_grad():
        l_bf = ((s["ref"](xa[:NS].cpu(), False) - e0c[:NS]) ** 2).sum(1)       # judged by the oracle encoder, like the fixture
        l_32 = ((s["ref"](x32[:NS], False) - e0c[:NS]) ** 2).sum(1)
    ratio16, ratio32 = float((l_bf / l_or[:NS]).mean()), float((l_32 / l_or[:NS]).mean())
    record("clip_like_config2_fare_pgd_b128", same_pixels_fp32_vs_reference=same_fp32, same_pixels_bf16_vs_reference=same_bf16,
           emb_rel_fp32=emb_rel32, emb_rel_bf16=emb_rel16, loss_ratio_bf16_over_reference=ratio16,
           loss_ratio_fp32_over_reference=ratio32, loss_end_over_start_reference=float(GOLDC["pgd_loss_end"].mean() / GOLDC["pgd_loss_start"].mean()))
    assert emb_rel32 < 1e-4, emb_rel32                       # north_star: fp32 embeddings within 1e-4 relative
    assert same_fp32 > 0.99, same_fp32
    assert 0.99 < ratio32 < 1.01, ratio32
    assert 0.97 < ratio16 < 1.03, ratio16
    assert same_bf16 > 0.70, same_bf16


def test_clip_like_gradient_signs_along_the_reference_trajectory(setup_clip):
    """The per-iteration bf16 gate on the CLIP-like tower: at every iterate of the REFERENCE's trajectory (replayed from the
    fixture's gradient signs with the oracle's bit-exact update) the HIP path evaluates forward + FARE loss + input gradient of
    the full B = 128 batch; the slice's gradient signs and losses are compared with the reference's at the same point."""
    s = setup_clip
    ns = int(GOLDC["traj_n"])
    x, d0 = s["x"].to(dev()), s["d0"].to(dev())
    signs_ref, loss_ref = GOLDC["traj_grad_sign"], GOLDC["traj_loss"]
    xn = s["x"][:ns].numpy().astype(np.float32)
    delta, vel = s["d0"][:ns].numpy().astype(np.float32).copy(), np.zeros_like(xn)
    eng = s["eng"]
    e0 = eng.forward(x, None, False, save=False)
    e0_32 = s["eng32"].forward(x[:ns], None, False, save=False)
    sign16, sign32, lr16, lr32 = [], [], [], []
    for k in range(signs_ref.shape[0]):
        d = d0.clone()
        d[:ns] = torch.from_numpy(delta).to(dev())
        _, per, _, g = eng.fwd_inputgrad(x, d, "l2", "mean", e0, None, False)
        _, per32, _, g32 = s["eng32"].fwd_inputgrad(x[:ns], d[:ns], "l2", "mean", e0_32, None, False)
        torch.cuda.synchronize()
        nz = signs_ref[k] != 0
        sign16.append(float(np.mean((np.sign(g[:ns].cpu().numpy()) == signs_ref[k])[nz])))
        sign32.append(float(np.mean((np.sign(g32.cpu().numpy()) == signs_ref[k])[nz])))
        lr16.append(float((per[:ns].cpu().numpy() / loss_ref[k]).mean()))
        lr32.append(float((per32.cpu().numpy() / loss_ref[k]).mean()))
        # the reference's own step from its own gradient signs (sign(g) is all the L-inf update uses)
        delta, vel = A.pgd_linf_update_ref(xn, signs_ref[k].astype(np.float32), delta, vel, EPS, STEP, 0.9, "max")
    assert np.array_equal(xn + delta, GOLDC["pgd_x_adv"][:ns]), "the replay must land on the reference's result"
    record("clip_like_gradient_signs_along_the_reference_trajectory",
           **{f"sign_agree_bf16_it{k}": v for k, v in enumerate(sign16)}, **{f"sign_agree_fp32_it{k}": v for k, v in enumerate(sign32)},
           **{f"loss_ratio_bf16_it{k}": v for k, v in enumerate(lr16)}, **{f"loss_ratio_fp32_it{k}": v for k, v in enumerate(lr32)})
    assert min(sign32) > 0.999 and all(0.999 < r < 1.001 for r in lr32), (sign32, lr32)
    # bf16 per-iteration sign agreement (it >= 1) >= 0.97, loss within 3 %
    assert min(sign16[1:]) >= 0.97, sign16
    assert all(0.97 < r < 1.03 for r in lr16[1:]), lr16


def test_clip_like_config3_vs_reference(setup_clip):
    """configs[2] (TeCoA apgd_train, 10 iterations) on the CLIP-like tower at B = 128 against the reference's own apgd_train on
    the first 8 images."""
    s = setup_clip
    n = s["n"]
    x, y, T = s["x"].to(dev()), s["y"].to(dev()), s["T"].to(dev())
    assert torch.equal(s["y"][:n], torch.from_numpy(GOLDC["c3_y"]))
    wrap = R.ComputeLossWrapper(None, T, "none", "ce", 100.)
    xa = R.apgd_train(R.ClipVisionModel(s["eng"]).eval(), x, y, "linf", EPS, n_iter=10, loss_fn=wrap)
    ball_and_range(xa, x)
    x32 = R.apgd_train(R.ClipVisionModel(s["eng32"]).eval(), x[:n], y[:n], "linf", EPS, n_iter=10, loss_fn=wrap).cpu()
    x_or = torch.from_numpy(GOLDC["c3_x_adv"])
    same_bf16, same_fp32 = float((xa[:n].cpu() == x_or).float().mean()), float((x32 == x_or).float().mean())
    with torch.no_grad():
        ce = lambda xx: Lr.compute_loss_ref("ce", s["ref"](xx, True), s["y"][:NS], None, 100., s["T"], "none")   # noqa: E731
        loss_ratio = float((ce(xa[:NS].cpu()) / torch.from_numpy(GOLDC["c3_loss_adv"])[:NS]).mean())
    record("clip_like_config3_tecoa_apgd_b128", same_pixels_bf16_vs_reference=same_bf16, same_pixels_fp32_vs_reference=same_fp32,
           loss_ratio_bf16_over_reference=loss_ratio)
    assert same_fp32 > 0.99, same_fp32
    assert same_bf16 > 0.90, same_bf16
    assert 0.97 < loss_ratio < 1.03, loss_ratio


def test_config2_x3_first_iteration(setup):
    """precision='bf16+x3-first' (VERDICT r5 item 4): the mixed mode with the first iteration and the clean embedding on the
    split-bf16 handle (fp32 storage, 3 bf16 MFMA products per linear) instead of the fp32 matrix pipe.  Bar: the first step agrees
    with the fp32 engine's on >= 0.995 of the pixels (emulation: 0.9996 gradient signs), the 10-step result with the reference's
    own pgd() on the 8-image slice as closely as the fp32-first mode does (0.928; bf16 alone 0.729)."""
    s = setup
    wd = {k: v.to(dev()) for k, v in s["w"].items()}
    eng = R.VitEngine(to_cfg(s["cfg"]), wd, precision="bf16+x3-first", max_batch=NP)
    eng3 = R.VitEngine(to_cfg(s["cfg"]), wd, precision="x3", max_batch=NP)
    try:
        x, d0 = s["x"][:NP].to(dev()), s["d0"][:NP].to(dev())
        model = R.ClipVisionModel(eng).eval()
        e0 = model(x, False)
        e0c = torch.from_numpy(GOLD["pgd_e0"])
        emb_rel = rel(e0.cpu(), e0c)
        assert emb_rel < 1e-4, emb_rel                                 # north_star's fp32 bar, met by the split-bf16 linears
        run = lambda m, e, n: R.pgd(m, R.ComputeLossWrapper(e, None, "mean", "l2", 100.), x, None, "linf", EPS, n, STEP, False,   # noqa: E731
                                    perturbation=d0.clone(), mode="max")
        xa = run(model, e0, 10)
        assert torch.equal(xa, run(model, e0, 10)), "not deterministic"
        ball_and_range(xa, x)
        x_or = torch.from_numpy(GOLD["pgd_x_adv"])
        same_mixed = float((xa.cpu() == x_or).float().mean())
        m32 = R.ClipVisionModel(s["eng32"]).eval()
        e32 = s["eng32"].forward(x, None, False, save=False)
        first_x3 = float((run(model, e0, 1) == run(m32, e32, 1)).float().mean())
        # the x3 engine alone, all ten iterations: the reference's trajectory to fp32-like accuracy
        m3 = R.ClipVisionModel(eng3).eval()
        same_x3 = float((run(m3, m3(x, False), 10).cpu() == x_or).float().mean())
        _, _, _, g3 = eng3.fwd_inputgrad(x, d0, "l2", "mean", eng3.forward(x, None, False, save=False), None, False)
        _, _, _, g32 = s["eng32"].fwd_inputgrad(x, d0, "l2", "mean", e32, None, False)
        sign0 = float((torch.sign(g3) == torch.sign(g32)).float().mean())
        record("config2_x3_first", same_pixels_mixed_x3_vs_reference=same_mixed, same_pixels_x3_engine_vs_reference=same_x3,
               first_step_same_pixels_x3_vs_fp32=first_x3, sign_agree_it0_x3_vs_fp32_engine=sign0, emb_rel_x3_vs_reference=emb_rel)
        assert sign0 > 0.995, sign0
        assert first_x3 > 0.995, first_x3
        assert same_x3 > 0.97, same_x3
        assert same_mixed > 0.90, same_mixed
    finally:
        eng.close(); eng3.close()


def test_config2_x3_forward_bf16_backward_first_iteration(setup):
    """precision='bf16+x3fwd-first' (round 6): like 'bf16+x3-first', but the first iteration's input gradient runs on the bf16
    backward kernels from the x3 forward's saved tensors (rvlm_pgd_run_mixed_fwd).  Emulation (oracle/split_bf16_emulation.py,
    arm x3fwd-bf16bwd-flash): 0.998 first-step gradient signs.  Bars: first step >= 0.99 of the fp32 engine's pixels, the
    ten-step result as close to the reference's own pgd() as the all-x3 first iteration gets (0.927) within 0.02."""
    s = setup
    wd = {k: v.to(dev()) for k, v in s["w"].items()}
    eng = R.VitEngine(to_cfg(s["cfg"]), wd, precision="bf16+x3fwd-first", max_batch=NP)
    try:
        x, d0 = s["x"][:NP].to(dev()), s["d0"][:NP].to(dev())
        model = R.ClipVisionModel(eng).eval()
        e0 = model(x, False)
        assert rel(e0.cpu(), torch.from_numpy(GOLD["pgd_e0"])) < 1e-4
        run = lambda m, e, n: R.pgd(m, R.ComputeLossWrapper(e, None, "mean", "l2", 100.), x, None, "linf", EPS, n, STEP, False,   # noqa: E731
                                    perturbation=d0.clone(), mode="max")
        xa = run(model, e0, 10)
        assert torch.equal(xa, run(model, e0, 10)), "not deterministic"
        ball_and_range(xa, x)
        same_mixed = float((xa.cpu() == torch.from_numpy(GOLD["pgd_x_adv"])).float().mean())
        m32 = R.ClipVisionModel(s["eng32"]).eval()
        e32 = s["eng32"].forward(x, None, False, save=False)
        first = float((run(model, e0, 1) == run(m32, e32, 1)).float().mean())
        _, gh = eng.handoff_inputgrad(x, d0, ref=e0)
        _, _, _, g32 = s["eng32"].fwd_inputgrad(x, d0, "l2", "mean", e32, None, False)
        sign0 = float((torch.sign(gh) == torch.sign(g32)).float().mean())
        record("config2_x3fwd_first", same_pixels_mixed_x3fwd_vs_reference=same_mixed, first_step_same_pixels_x3fwd_vs_fp32=first,
               sign_agree_it0_x3fwd_vs_fp32_engine=sign0)
        assert sign0 > 0.99, sign0
        assert first > 0.99, first
        assert same_mixed > 0.90, same_mixed
    finally:
        eng.close()
    # 'bf16+x3fwd': the same split (split-bf16 forward, bf16 backward) in EVERY iteration
    eng = R.VitEngine(to_cfg(s["cfg"]), wd, precision="bf16+x3fwd", max_batch=NP)
    try:
        model = R.ClipVisionModel(eng).eval()
        xa = run(model, model(x, False), 10)
        ball_and_range(xa, x)
        same_all = float((xa.cpu() == torch.from_numpy(GOLD["pgd_x_adv"])).float().mean())
        record("config2_x3fwd_all", same_pixels_x3fwd_every_iteration_vs_reference=same_all)
        assert same_all > same_mixed, (same_all, same_mixed)
    finally:
        eng.close()


def test_clip_like_x3_vs_reference(setup_clip):
    """The split-bf16 precision on the CLIP-like tower (outlier channels 30-100x, LayerNorm gains over 2.5 decades): the x3 engine
    alone against the reference's pgd() on the 8-image slice, and its first-iteration gradient signs against the reference's."""
    s = setup_clip
    n = s["n"]
    w = V.init_weights(s["cfg"], seed=3, clip_like=True)
    eng3 = R.VitEngine(to_cfg(s["cfg"]), {k: v.to(dev()) for k, v in w.items()}, precision="x3", max_batch=n)
    try:
        x, d0 = s["x"][:n].to(dev()), s["d0"][:n].to(dev())
        m3 = R.ClipVisionModel(eng3).eval()
        e0 = m3(x, False)
        emb_rel = rel(e0.cpu(), torch.from_numpy(GOLDC["pgd_e0"]))
        xa = R.pgd(m3, R.ComputeLossWrapper(e0, None, "mean", "l2", 100.), x, None, "linf", EPS, 10, STEP, False,
                   perturbation=d0.clone(), mode="max")
        same = float((xa.cpu() == torch.from_numpy(GOLDC["pgd_x_adv"])).float().mean())
        ns = int(GOLDC["traj_n"])
        _, per, _, g = eng3.fwd_inputgrad(x, d0, "l2", "mean", e0, None, False)
        sg = GOLDC["traj_grad_sign"][0]
        sign0 = float(np.mean((np.sign(g[:ns].cpu().numpy()) == sg)[sg != 0]))
        record("clip_like_x3", same_pixels_x3_vs_reference=same, sign_agree_it0_x3_vs_reference=sign0, emb_rel_x3_vs_reference=emb_rel,
               loss_ratio_it0=float((per[:ns].cpu().numpy() / GOLDC["traj_loss"][0]).mean()))
        assert emb_rel < 1e-4, emb_rel
        assert sign0 > 0.995, sign0
        assert same > 0.97, same
    finally:
        eng3.close()


def test_clip_like_x3_forward_bf16_backward(setup_clip):
    """The handoff (precision='bf16+x3fwd-first' / 'bf16+x3fwd') on the CLIP-like tower: first-iteration gradient signs against the
    REFERENCE's own (emulation: 0.9955 with the flash-style backward), ten-step pgd() against the reference's."""
    s = setup_clip
    n = s["n"]
    w = V.init_weights(s["cfg"], seed=3, clip_like=True)
    wd = {k: v.to(dev()) for k, v in w.items()}
    out = {}
    for prec in ("bf16+x3fwd-first", "bf16+x3fwd"):
        eng = R.VitEngine(to_cfg(s["cfg"]), wd, precision=prec, max_batch=n)
        try:
            x, d0 = s["x"][:n].to(dev()), s["d0"][:n].to(dev())
            model = R.ClipVisionModel(eng).eval()
            e0 = model(x, False)
            assert rel(e0.cpu(), torch.from_numpy(GOLDC["pgd_e0"])) < 1e-4
            xa = R.pgd(model, R.ComputeLossWrapper(e0, None, "mean", "l2", 100.), x, None, "linf", EPS, 10, STEP, False,
                       perturbation=d0.clone(), mode="max")
            out[prec] = float((xa.cpu() == torch.from_numpy(GOLDC["pgd_x_adv"])).float().mean())
            if prec == "bf16+x3fwd-first":
                ns = int(GOLDC["traj_n"])
                _, g = eng.handoff_inputgrad(x, d0, ref=e0)
                sg = GOLDC["traj_grad_sign"][0]
                sign0 = float(np.mean((np.sign(g[:ns].cpu().numpy()) == sg)[sg != 0]))
        finally:
            eng.close()
    record("clip_like_x3fwd", sign_agree_it0_x3fwd_vs_reference=sign0, same_pixels_x3fwd_first_vs_reference=out["bf16+x3fwd-first"],
           same_pixels_x3fwd_every_iteration_vs_reference=out["bf16+x3fwd"])
    assert sign0 > 0.99, sign0
    assert out["bf16+x3fwd-first"] > 0.85, out
    assert out["bf16+x3fwd"] > 0.95, out


def test_config2_faithful_iterations_curve(setup):
    """VitEngine(precision='bf16+x3fwd-first', faithful_iterations=k): the first k iterations with a split-bf16 forward (bf16 backward),
    the rest bf16 - identical pixels with the reference's pgd() as a function of k (k = 1 is the '-first' mode, 10 'bf16+x3fwd')."""
    s = setup
    wd = {k: v.to(dev()) for k, v in s["w"].items()}
    x, d0 = s["x"][:NP].to(dev()), s["d0"][:NP].to(dev())
    x_or = torch.from_numpy(GOLD["pgd_x_adv"])
    curve = {}
    for k in (0, 1, 2, 3, 5, 10):
        eng = R.VitEngine(to_cfg(s["cfg"]), wd, precision="bf16+x3fwd-first", max_batch=NP, faithful_iterations=k)
        try:
            model = R.ClipVisionModel(eng).eval()
            e0 = model(x, False)
            torch.cuda.synchronize()
            t0 = time.time()
            xa = R.pgd(model, R.ComputeLossWrapper(e0, None, "mean", "l2", 100.), x, None, "linf", EPS, 10, STEP, False,
                       perturbation=d0.clone(), mode="max")
            torch.cuda.synchronize()
            curve[k] = (float((xa.cpu() == x_or).float().mean()), (time.time() - t0) * 1e3)
        finally:
            eng.close()
    record("config2_faithful_iterations_curve", **{f"same_pixels_k{k}": v[0] for k, v in curve.items()},
           **{f"ms_k{k}": v[1] for k, v in curve.items()})
    assert curve[1][0] > curve[0][0] + 0.1 and curve[10][0] > curve[3][0] > curve[1][0] - 0.01, curve
