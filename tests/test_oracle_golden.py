"""The oracle (oracle/*.py) against the reference's own outputs (tests/golden/*.npz).

These are the pins that make the oracle trustworthy: bit-equality for the Linf index/sign/clamp
arithmetic and the APGD controller, <=1e-5 relative for the fp32 ViT against HF transformers.
CPU only.
"""
import numpy as np
import pytest
import torch

from oracle import attacks_ref as A
from oracle import losses_ref as Lr
from oracle import vit_ref as V
from tests.helpers import (load_golden, weights_digest, cfg_from_array, weights_from_golden,
                           rel_err, SmallNet, InjectGrad)

torch.set_num_threads(4)


@pytest.mark.parametrize("mode", ["max", "min"])
def test_pgd_linf_elementwise_bit_exact(mode):
    z = load_golden(f"pgd_linf_elementwise_{mode}.npz")
    x, d0, grads = z["x"], z["delta0"], z["grads"]
    eps, step = float(z["eps"]), float(z["stepsize"])
    delta, vel = d0.copy(), np.zeros_like(d0)
    for i in range(4):
        delta, vel = A.pgd_linf_update_ref(x, grads[i], delta, vel, eps, step, 0.9, mode)
        assert np.array_equal((x + delta).astype(np.float32), z["xadv"][i]), f"step {i}"
    # and through the full loop restatement with an injected-gradient loss
    xt = torch.from_numpy(x)
    it = iter(range(4))
    out = A.pgd_ref(lambda v, output_normalize=False: v,
                    lambda o, t: InjectGrad.apply(o, torch.from_numpy(grads[next(it)])),
                    xt, None, "linf", eps, 4, step, False,
                    perturbation=torch.from_numpy(d0.copy()), mode=mode)
    assert np.array_equal(out.numpy(), z["xadv"][3])


@pytest.mark.parametrize("n_iter", [10, 50, 100])
def test_apgd_train_controller_bit_exact(n_iter):
    z = load_golden(f"apgd_train_smallnet_{n_iter}.npz")
    net = SmallNet(torch.from_numpy(z["w1"]), torch.from_numpy(z["w2"])).eval()
    ce = lambda lg, yy: torch.nn.functional.cross_entropy(lg, yy, reduction="none")  # noqa: E731
    out = A.apgd_train_ref(net, torch.from_numpy(z["x"]), torch.from_numpy(z["y"]), "linf",
                           float(z["eps"]), n_iter=n_iter, loss_fn=ce)
    its = np.stack([t.numpy() for t in net.seen])
    assert its.shape == z["iterates"].shape
    for i in range(its.shape[0]):
        assert np.array_equal(its[i], z["iterates"][i]), f"iterate {i} differs"
    assert np.array_equal(out.numpy(), z["x_best_adv"])


def test_apgd_schedule_constants():
    # n_iter=10 -> (2,1,1); 50 -> (11,3,1); 100 -> (22,6,3)
    assert A.apgd_schedule(10) == (2, 1, 1)
    assert A.apgd_schedule(50) == (11, 3, 1)
    assert A.apgd_schedule(100) == (22, 6, 3)


def test_losses_match_reference():
    z = load_golden("losses.npz")
    emb, e0, T, y = (torch.from_numpy(z[k]) for k in ("emb", "e0", "T", "y"))
    for loss in ("l2", "ce"):
        for red in ("mean", "none"):
            e = emb.clone().requires_grad_(True)
            val = Lr.compute_loss_ref(loss, e, y, e0, 100., T, red)
            (g,) = torch.autograd.grad(val.sum(), e)
            assert np.array_equal(val.detach().numpy(), z[f"{loss}_{red}"])
            assert np.array_equal(g.numpy(), z[f"{loss}_{red}_grad"])
    assert Lr.compute_acc_ref(emb @ (100. * T), y) == float(z["acc"])
    with pytest.raises(AssertionError):
        Lr.l2_ref(emb[:1], e0[:1])          # batch size 1 is illegal (…clip.py:513)
    with pytest.raises(ValueError):
        Lr.compute_loss_ref("dlr", emb, y, e0, 100., T)


@pytest.mark.parametrize("name", ["tiny2", "tiny2gelu", "b32"])
def test_vit_matches_hf(name):
    z = load_golden(f"vit_hf_{name}.npz")
    cfg = cfg_from_array(z["cfg"], str(z["act"]))
    w = V.init_weights(cfg, seed=int(z["weights_seed"]))
    assert weights_digest(w) == str(z["weights_sha256"]), "seeded weight generator drifted"
    x = torch.from_numpy(z["x"])
    xn = V.normalize_pixels(x).requires_grad_(True)
    emb = V.vit_forward(cfg, w, xn)
    (gx,) = torch.autograd.grad((emb * torch.from_numpy(z["cot"])).sum(), xn)
    assert rel_err(emb.detach().numpy(), z["emb"]) < 1e-5
    assert rel_err(gx.numpy(), z["grad_xn"]) < 1e-4
    agree = np.mean(np.sign(gx.numpy()) == np.sign(z["grad_xn"]))
    assert agree > 0.999


def _tiny():
    z = load_golden("tiny_vit_attacks.npz")
    cfg = cfg_from_array(z["cfg"])
    w = weights_from_golden(z)
    assert weights_digest(w) == str(z["weights_sha256"])
    w2 = V.init_weights(cfg, seed=int(z["weights_seed"]))
    assert weights_digest(w2) == str(z["weights_sha256"]), "seeded weight generator drifted"
    return z, cfg, w


@pytest.mark.parametrize("loss_name,on", [("l2", False), ("ce", True)])
def test_tiny_vit_pgd_end_to_end_bit_exact(loss_name, on):
    z, cfg, w = _tiny()
    model = V.ClipVisionModelRef(cfg, w).eval()
    x, y, T = (torch.from_numpy(z[k]) for k in ("x", "y", "T"))
    with torch.no_grad():
        e0 = model(x, on)
    assert np.array_equal(e0.numpy(), z[f"e0_norm{int(on)}"])
    wrap = Lr.ComputeLossWrapperRef(e0, T, "mean", loss_name, 100.)
    out = A.pgd_ref(model, wrap, x, y, "linf", float(z["eps"]), 10, float(z["stepsize"]), on,
                    perturbation=torch.from_numpy(z["delta0"].copy()), mode="max")
    assert np.array_equal(out.numpy(), z[f"pgd_{loss_name}_xadv"])


@pytest.mark.parametrize("loss_name", ["l2", "ce"])
def test_tiny_vit_apgd_end_to_end_bit_exact(loss_name):
    z, cfg, w = _tiny()
    model = V.ClipVisionModelRef(cfg, w).eval()
    x, y, T = (torch.from_numpy(z[k]) for k in ("x", "y", "T"))
    with torch.no_grad():
        e0 = model(x, True)
    wrap = Lr.ComputeLossWrapperRef(e0, T, "none", loss_name, 100.)
    out = A.apgd_train_ref(model, x, y, "linf", float(z["eps"]), n_iter=10, loss_fn=wrap)
    assert np.array_equal(out.numpy(), z[f"apgd_{loss_name}_xadv"])


@pytest.mark.parametrize("r", [1, 2])
def test_autopgd_bit_exact(r):
    z = load_golden(f"autopgd_tiny_r{r}.npz")
    cfg = V.VIT_TINY
    w = V.init_weights(cfg, seed=int(z["weights_seed"]))
    assert weights_digest(w) == str(z["weights_sha256"])
    clf = V.ClassificationModelRef(cfg, w, torch.from_numpy(z["T"]), 100.0).eval()
    seen = []

    def predict(v):
        seen.append(v.detach().clone())
        return clf(v)

    atk = A.APGDAttackRef(predict, n_iter=int(z["n_iter"]), norm="Linf", n_restarts=r,
                          eps=float(z["eps"]), seed=0, loss="ce", alpha=2.0, use_rs=True)
    adv = atk.perturb(torch.from_numpy(z["x"]), torch.from_numpy(z["y"]))
    assert np.array_equal(seen[1].numpy(), z["first_start"])
    assert len(seen) == int(z["n_model_calls"])
    assert np.array_equal(adv.numpy(), z["adv"])
    # never-attacked sample (started misclassified) comes back untouched
    assert np.array_equal(adv.numpy()[0], z["x"][0])


@pytest.mark.parametrize("tag", ["rho050", "rho090"])
def test_autopgd_rho_bit_exact(tag):
    """APGDAttack's `rho` (oscillation threshold, autopgd_base.py:111,137,415-416) away from 0.75, at a radius where
    it changes the result."""
    z = load_golden(f"autopgd_tiny_{tag}.npz")
    cfg = V.VIT_TINY
    w = V.init_weights(cfg, seed=int(z["weights_seed"]))
    clf = V.ClassificationModelRef(cfg, w, torch.from_numpy(z["T"]), 100.0).eval()
    kw = dict(n_iter=int(z["n_iter"]), norm="Linf", n_restarts=1, eps=float(z["eps"]), seed=0, loss="ce", use_rs=True)
    adv = A.APGDAttackRef(clf, rho=float(z["rho"]), **kw).perturb(torch.from_numpy(z["x"]), torch.from_numpy(z["y"]))
    assert np.array_equal(adv.numpy(), z["adv"])
    other = A.APGDAttackRef(clf, rho=0.75, **kw).perturb(torch.from_numpy(z["x"]), torch.from_numpy(z["y"]))
    assert not np.array_equal(other.numpy(), z["adv"])          # the parameter matters on this input


def test_autopgd_and_autoattack_l2_bit_exact():
    """The L2 norm of APGDAttack / AutoAttack (``clip_robustbench.py --norm l2``; autopgd_base.py:184-185,215-218,343-351):
    gaussian random start on the L2 sphere, the L2 step in APGDAttack's spelling, two restarts; then the two APGD stages
    of AutoAttack(version='custom').  Against what the reference produced (tests/golden/autopgd_tiny_l2.npz, g10)."""
    from oracle import autoattack_ref as AA
    z = load_golden("autopgd_tiny_l2.npz")
    cfg = V.VIT_TINY
    w = V.init_weights(cfg, seed=int(z["weights_seed"]))
    assert weights_digest(w) == str(z["weights_sha256"])
    clf = V.ClassificationModelRef(cfg, w, torch.from_numpy(z["T"]), 100.0).eval()
    seen = []

    def predict(v):
        seen.append(v.detach().clone())
        return clf(v)

    x = torch.from_numpy(z["x"])
    atk = A.APGDAttackRef(predict, n_iter=int(z["n_iter"]), norm="L2", n_restarts=2, eps=float(z["eps"]), seed=0,
                          loss="ce", use_rs=True)
    adv = atk.perturb(x, torch.from_numpy(z["y"]))
    assert np.array_equal(seen[1].numpy(), z["first_start"])
    assert len(seen) == int(z["n_model_calls"])
    assert np.array_equal(adv.numpy(), z["adv"])
    assert 0 < z["robust"].sum() < len(z["robust"])                 # the radius separates the points
    assert float((adv - x).flatten(1).norm(dim=1).max()) <= float(z["eps"]) * (1 + 1e-5)
    seen.clear()
    aa = AA.AutoAttackRef(predict, norm="L2", eps=float(z["eps"]), seed=0, version="custom",
                          attacks_to_run=["apgd-ce", "apgd-t"], iterations_apgd=int(z["aa_n_iter"]), use_rs=True)
    aa.apgd.n_restarts = 1
    aa.apgd_targeted.n_target_classes = int(z["aa_n_target_classes"])
    xa, ya = aa.run_standard_evaluation(x, torch.from_numpy(z["aa_y"]), bs=int(z["aa_bs"]), return_labels=True)
    assert len(seen) == int(z["aa_n_model_calls"])
    assert np.array_equal(xa.numpy(), z["aa_x_adv"]) and np.array_equal(ya.numpy(), z["aa_y_adv"])


# ------------------------------------------------------------------ section 8(f) rank 3: AutoAttack orchestration
def test_dlr_losses_bit_exact():
    z = load_golden("dlr_losses.npz")
    obj = A.APGDAttackRef(lambda v: v, eps=0.1, n_iter=5, loss="dlr")
    for name, tgt in (("dlr", None), ("dlr_targeted", torch.from_numpy(z["y_target"]))):
        lg = torch.from_numpy(z["logits"]).clone().requires_grad_(True)
        obj.y_target = tgt
        fn = obj.dlr_loss if tgt is None else obj.dlr_loss_targeted
        loss = fn(lg, torch.from_numpy(z["y"]))
        (g,) = torch.autograd.grad(loss.sum(), lg)
        assert np.array_equal(loss.detach().numpy(), z[name])
        assert np.array_equal(g.numpy(), z[name + "_grad"])


@pytest.mark.parametrize("tag", ["e3", "e5"])
def test_autoattack_orchestration_bit_exact(tag):
    from oracle import autoattack_ref as AA
    z = load_golden("autoattack_tiny.npz")
    cfg = V.VIT_TINY
    w = V.init_weights(cfg, seed=int(z["weights_seed"]))
    assert weights_digest(w) == str(z["weights_sha256"])
    clf = V.ClassificationModelRef(cfg, w, torch.from_numpy(z["T"]), 100.0).eval()
    calls = []

    def predict(v):
        calls.append(tuple(v.shape))
        return clf(v)

    x, y, eps = torch.from_numpy(z["x"]), torch.from_numpy(z["y"]), float(z[tag + "_eps"])
    atk = AA.APGDAttackTargetedRef(predict, n_iter=int(z["n_iter"]), norm="Linf", n_restarts=1, eps=eps, seed=0,
                                   n_target_classes=int(z["n_target_classes"]), alpha=2.0, use_rs=True)
    adv_t = atk.perturb(x.clone(), y.clone())
    assert len(calls) == int(z[tag + "_n_model_calls_targeted"])
    assert np.array_equal(adv_t.numpy(), z[tag + "_adv_targeted"])
    calls.clear()
    aa = AA.AutoAttackRef(predict, norm="Linf", eps=eps, seed=0, version="custom", attacks_to_run=["apgd-ce", "apgd-t"],
                          alpha=2.0, iterations_apgd=int(z["n_iter"]), use_rs=True)
    aa.apgd.n_restarts = 1
    aa.apgd_targeted.n_target_classes = int(z["n_target_classes"])
    x_adv, y_adv = aa.run_standard_evaluation(x.clone(), y.clone(), bs=int(z["bs"]), return_labels=True)
    assert len(calls) == int(z[tag + "_n_model_calls_aa"])
    assert np.array_equal(x_adv.numpy(), z[tag + "_x_adv"])
    assert np.array_equal(y_adv.numpy(), z[tag + "_y_adv"])
    with torch.no_grad():
        assert np.array_equal((clf(x_adv).max(1)[1] == y).numpy(), z[tag + "_robust"])


@pytest.mark.parametrize("tag", ["m", "c", "a"])
def test_square_attack_bit_exact(tag):
    """oracle/square_ref.py against what the reference's SquareAttack produced (tests/golden/square_tiny.npz):
    perturb() with two restarts (adversarial images, number of model calls) and one single run on every sample (the
    per-query accept / reject trajectory: x_best and the query counters)."""
    from oracle.square_ref import SquareAttackRef
    z = load_golden("square_tiny.npz")
    cfg = V.VIT_TINY
    w = V.init_weights(cfg, seed=int(z["weights_seed"]))
    assert weights_digest(w) == str(z["weights_sha256"])
    clf = V.ClassificationModelRef(cfg, w, torch.from_numpy(z["T"]), 100.0).eval()
    calls = []

    def predict(v):
        calls.append(tuple(v.shape))
        with torch.no_grad():
            return clf(v)

    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    loss, resc = {"m": ("margin", True), "c": ("ce", True), "a": ("margin", False)}[tag]
    atk = SquareAttackRef(predict, norm="Linf", n_queries=int(z[tag + "_n_queries"]), eps=float(z[tag + "_eps"]),
                          p_init=.8, n_restarts=2, seed=5, loss=loss, resc_schedule=resc)
    adv = atk.perturb(x.clone(), y.clone())
    assert len(calls) == int(z[tag + "_n_model_calls"])
    assert np.array_equal(adv.numpy(), z[tag + "_adv"])
    torch.random.manual_seed(11)
    nq, xb = atk.attack_single_run(x.clone(), y.clone())
    assert np.array_equal(nq.numpy(), z[tag + "_run_queries"])
    assert np.array_equal(xb.numpy(), z[tag + "_run_x_best"])


def test_square_through_autoattack_bit_exact():
    from oracle import autoattack_ref as AA
    z = load_golden("square_tiny.npz")
    cfg = V.VIT_TINY
    w = V.init_weights(cfg, seed=int(z["weights_seed"]))
    clf = V.ClassificationModelRef(cfg, w, torch.from_numpy(z["T"]), 100.0).eval()
    calls = []

    def predict(v):
        calls.append(tuple(v.shape))
        with torch.no_grad():
            return clf(v)

    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    aa = AA.AutoAttackRef(predict, norm="Linf", eps=float(z["aa_eps"]), seed=0, version="custom",
                          attacks_to_run=["square"])
    aa.square.n_queries = int(z["aa_n_queries"])
    x_adv, y_adv = aa.run_standard_evaluation(x.clone(), y.clone(), bs=4, return_labels=True)
    assert len(calls) == int(z["aa_n_model_calls"])
    assert np.array_equal(x_adv.numpy(), z["aa_x_adv"])
    assert np.array_equal(y_adv.numpy(), z["aa_y_adv"])


# ------------------------------------------------------------------ section 8(f) rank 4: input transform
def test_preprocess_oracle_bit_exact_vs_pillow_golden():
    from oracle import preprocess_ref as P
    z = load_golden("preprocess_pil.npz")
    for i in range(int(z["n"])):
        got = P.preprocess_ref(z[f"img{i}"], int(z[f"size{i}"]))
        want = z[f"crop{i}"].transpose(2, 0, 1).astype(np.float32) / np.float32(255)
        assert got.dtype == np.float32 and np.array_equal(got, want), f"case {i}"
    assert P.resize_size(375, 500, 224) == (224, 298) and P.resize_size(500, 333, 224) == (336, 224)
    assert P.center_crop_box(224, 298, 224, 224) == (0, 37) and P.center_crop_box(336, 224, 224, 224) == (56, 0)


# ---- L2-norm branch of pgd (vlm_eval/attacks/utils.py:12-14,22-26), tests/golden/pgd_l2norm.npz (make_golden_l2norm.py)
@pytest.mark.parametrize("mode", ["max", "min"])
def test_pgd_l2norm_elementwise_bit_exact(mode):
    from tests.helpers import InjectGrad
    z = load_golden("pgd_l2norm.npz")
    x, d0 = torch.from_numpy(z["ew_x"]), torch.from_numpy(z["ew_delta0"])
    grads = [torch.from_numpy(g) for g in z["ew_grads"]]
    for n_it in range(1, 5):
        it = iter(range(n_it))
        out = A.pgd_ref(lambda v, output_normalize=False: v, lambda o, t: InjectGrad.apply(o, grads[next(it)]), x, None,
                        "l2", float(z["ew_eps"]), n_it, float(z["ew_stepsize"]), False, perturbation=d0.clone(), mode=mode)
        assert np.array_equal(out.numpy(), z[f"ew_xadv_{mode}"][n_it - 1]), n_it
    # the eps ball is the L2 ball, per sample
    dn = (torch.from_numpy(z[f"ew_xadv_{mode}"][3]) - x).flatten(1).norm(dim=1)
    assert float(dn.max()) <= float(z["ew_eps"]) * (1 + 1e-6)


def test_pgd_l2norm_tiny_vit_bit_exact():
    z, cfg, w = _tiny()
    g = load_golden("pgd_l2norm.npz")
    model = V.ClipVisionModelRef(cfg, w).eval()
    x = torch.from_numpy(z["x"])
    with torch.no_grad():
        e0 = model(x, False)
    wrap = Lr.ComputeLossWrapperRef(e0, None, "mean", "l2", 100.)
    out = A.pgd_ref(model, wrap, x, None, "l2", float(g["vit_eps"]), 10, float(g["vit_stepsize"]), False,
                    perturbation=torch.from_numpy(z["delta0"].copy()), mode="max")
    assert np.array_equal(out.numpy(), g["vit_xadv"])


@pytest.mark.parametrize("loss_name,n_iter", [("l2", 10), ("l2", 25), ("ce", 10), ("ce", 25)])
def test_apgd_train_l2norm_tiny_vit_bit_exact(loss_name, n_iter):
    """apgd_train(norm='l2') (train/apgd_train.py:231-254) with FARE / TeCoA losses: the oracle's L2 step + the shared
    controller against the reference's x_best_adv, 10 iterations and 25 (several step-size checkpoints)."""
    z, cfg, w = _tiny()
    g = load_golden("pgd_l2norm.npz")
    model = V.ClipVisionModelRef(cfg, w).eval()
    x, y, T = (torch.from_numpy(z[k]) for k in ("x", "y", "T"))
    wrap = Lr.ComputeLossWrapperRef(torch.from_numpy(g["apgd_e0"]), T, "none", loss_name, 100.)
    out = A.apgd_train_ref(model, x, y, "l2", 1.0, n_iter=n_iter, loss_fn=wrap)
    assert np.array_equal(out.numpy(), g[f"apgd_{loss_name}_{n_iter}_xadv"])
    assert float((out - x).flatten(1).norm(dim=1).max()) <= 1.0 + 1e-5


# --------------------------------------------------------------------------------------------------------------------
# round 4: the optimizer step against the reference's OWN train_one_epoch (tests/golden/make_golden_train.py)
# --------------------------------------------------------------------------------------------------------------------
TRAIN_CASES = {   # the args fields of each recorded run (same table as make_golden_train.py::CASES)
    "fare_pgd": dict(loss="l2", loss_clean="l2", clean_weight=0.0, trades=False, output_normalize=False),
    "tecoa_pgd": dict(loss="ce", loss_clean="l2", clean_weight=0.0, trades=False, output_normalize=True),
    "none_cw": dict(loss="ce", loss_clean="l2", clean_weight=0.5, trades=False, output_normalize=True),
    "apgd_trades_cw": dict(loss="l2", loss_clean="l2", clean_weight=0.3, trades=True, output_normalize=True),
}


@pytest.mark.parametrize("name", sorted(TRAIN_CASES))
def test_train_step_oracle_vs_reference_train_one_epoch(name):
    """oracle/train_ref.py against what the reference's train_one_epoch (…clip.py:276-486) did on the same batches:
    learning rate USED by every optimizer step, loss / loss-total / cos-sim-clean / cos-sim / acc / racc, and every
    parameter after the step, bit for bit (same torch, same ops in the same order)."""
    from oracle.train_ref import TrainStepRef
    z = load_golden("train_step_tiny.npz")
    assert sorted(TRAIN_CASES) == [str(c) for c in z["cases"]]
    cfg = cfg_from_array(z["cfg"])
    w = weights_from_golden(z)
    c = TRAIN_CASES[name]
    T = torch.from_numpy(z["T"])
    ref = TrainStepRef(cfg, w, lr=float(z["lr"]), wd=float(z["wd"]), warmup=int(z["warmup"]), steps=int(z["steps"]),
                       T=T, **c)
    model_orig = V.ClipVisionModelRef(cfg, w)
    keys = list(w)
    n_steps = z["x"].shape[0]
    for s in range(n_steps):
        x, y = torch.from_numpy(z["x"][s]), torch.from_numpy(z["y"][s])
        xa = torch.from_numpy(z[f"{name}::x_adv"][s]) if f"{name}::x_adv" in z.files else x
        with torch.no_grad():
            e0 = model_orig(x, c["output_normalize"])
        lr_used = ref.opt.param_groups[0]["lr"]
        loss, _ = ref.step(x, xa, y, e0)
        m = ref.last_metrics
        assert lr_used == float(z[f"{name}::lr_used"][s]), (s, lr_used)
        assert ref.opt.param_groups[0]["lr"] == float(z[f"{name}::lr_after"][s])
        for key, got in (("loss", loss), ("loss_total", m["loss_total"]), ("cos_sim_clean", m["cos_sim_clean"]),
                         ("cos_sim", m["cos_sim"]), ("acc", m["acc"]), ("racc", m["racc"])):
            assert got == float(z[f"{name}::{key}"][s]), (s, key, got, float(z[f"{name}::{key}"][s]))
        wsum = np.array([float(ref.w[k].detach().double().sum()) for k in keys])
        wsq = np.array([float((ref.w[k].detach().double() ** 2).sum()) for k in keys])
        assert np.array_equal(wsum, z[f"{name}::wsum"][s]) and np.array_equal(wsq, z[f"{name}::wsq"][s]), s
        if f"{name}::w{s + 1}::{keys[0]}" in z.files:
            for k in keys:
                assert np.array_equal(ref.w[k].detach().numpy(), z[f"{name}::w{s + 1}::{k}"]), (s, k)


@pytest.mark.parametrize("name", ["fare_pgd", "none_cw"])
def test_eval_oracle_vs_reference_train_one_epoch(name):
    """The validation the reference runs after its first optimizer step (…clip.py:389-424) on the weights of that step:
    the 50-step supervised APGD batch bit for bit, acc / racc / cos-sim equal."""
    from oracle.train_ref import eval_ref
    z = load_golden("train_step_tiny.npz")
    cfg = cfg_from_array(z["cfg"])
    c = TRAIN_CASES[name]
    if f"{name}::w1::class_embedding" in z.files:
        w1 = {k[len(f"{name}::w1::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{name}::w1::")}
    else:                              # only the last step's parameters are stored in full: replay step 1 with the oracle
        from oracle.train_ref import TrainStepRef
        w = weights_from_golden(z)
        T = torch.from_numpy(z["T"])
        ref = TrainStepRef(cfg, w, lr=float(z["lr"]), wd=float(z["wd"]), warmup=int(z["warmup"]),
                           steps=int(z["steps"]), T=T, **c)
        x, y = torch.from_numpy(z["x"][0]), torch.from_numpy(z["y"][0])
        xa = torch.from_numpy(z[f"{name}::x_adv"][0]) if f"{name}::x_adv" in z.files else x
        with torch.no_grad():
            e0 = V.ClipVisionModelRef(cfg, w)(x, c["output_normalize"])
        ref.step(x, xa, y, e0)
        w1 = {k: v.detach() for k, v in ref.w.items()}
    acc, racc, cs, adv = eval_ref(cfg, w1, torch.from_numpy(z["x_eval"]), torch.from_numpy(z["y_eval"]),
                                  torch.from_numpy(z["T"]), float(z["eps"]), c["clean_weight"])
    assert np.array_equal(adv.numpy(), z[f"{name}::x_adv_eval"])
    want = z[f"{name}::eval"]
    assert acc == want[0] and racc == want[1] and cs == want[2], (acc, racc, cs, want)


def test_clip_like_tower_is_the_fixtures_and_is_clip_like():
    """oracle/vit_ref.py::make_clip_like (round 6): deterministic across hosts (the sha256 the reference-run fixture
    tests/golden/l14_slices_clip.npz recorded), the oracle reproduces the reference's clean embeddings of that fixture, and the
    tower has the statistics it is named after - outlier residual channels 30-100x the ordinary RMS at the input of the first
    block, LayerNorm gains over more than two decades, peaked attention."""
    import hashlib
    import os
    import torch.nn.functional as F
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "l14_slices_clip.npz"))
    cfg = V.VIT_L_14
    torch.set_num_threads(8)
    rec = {}
    w = V.init_weights(cfg, seed=3, clip_like=True, record=rec)
    assert np.array_equal(np.array(rec["calib"]), g["weights_calib"])
    fp, fp0 = V.weights_fingerprint(w), g["weights_fingerprint"]
    assert np.allclose(fp, fp0, rtol=1e-6, atol=1e-6 * np.abs(fp0[:, 1:]).max())
    sha = hashlib.sha256()
    for k in sorted(w):
        sha.update(w[k].numpy().tobytes())
    if torch.get_num_threads() == int(g["threads"]):       # the host that made the fixture: the very same bits
        assert sha.hexdigest() == str(g["weights_sha256"])
    x = torch.rand(256, 3, 224, 224, generator=torch.Generator().manual_seed(0))[:2]
    with torch.no_grad():
        e, tok = V.vit_forward(cfg, w, V.normalize_pixels(x), return_tokens=True)
        # (thread count / blocking may differ from the generating host: fp32 summation order, not bits)
        assert float((e - torch.from_numpy(g["pgd_e0"][:2])).abs().max() / e.abs().max()) < 2e-5
        W = cfg.width
        t = F.conv2d(V.normalize_pixels(x), w["conv1.weight"], stride=cfg.patch).reshape(2, W, -1).permute(0, 2, 1)
        t = torch.cat([w["class_embedding"].expand(2, 1, W), t], 1) + w["positional_embedding"]
        t = F.layer_norm(t, (W,), w["ln_pre.weight"], w["ln_pre.bias"], 1e-5)
        mag = t.abs().mean(dim=(0, 1))
        top = mag.topk(V.CLIP_LIKE_OUTLIERS).values
        rms = float(t[..., mag < top.min()].pow(2).mean().sqrt())
        assert 25 < float(top.min()) / rms and float(top.max()) / rms < 120, (top, rms)
        gains = torch.cat([w[k] for k in w if "ln_" in k and k.endswith(".weight")])
        assert float(gains.max() / gains.min()) > 150
        p = "transformer.resblocks.0."
        h = F.layer_norm(t, (W,), w[p + "ln_1.weight"], w[p + "ln_1.bias"], 1e-5)
        q, k, _ = F.linear(h, w[p + "attn.in_proj_weight"], w[p + "attn.in_proj_bias"]).split(W, -1)
        q, k = (z.reshape(2, -1, cfg.heads, 64).transpose(1, 2) for z in (q, k))
        pmax = torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1).amax(-1).mean()
        assert float(pmax) > 10.0 / cfg.tokens          # far from uniform attention (1 / 257)
