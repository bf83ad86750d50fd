"""Bit-exact parity of the L-inf attack kernels with the oracle (C restatement) and the reference's
own outputs (golden vectors): sign / momentum / step / project / clamp index arithmetic."""
import numpy as np
import pytest
import torch

from oracle import linf_c
from robustvlm_amd import _lib as L
import robustvlm_amd as R
from tests.gpu_helpers import dev, st, lib
from tests.helpers import load_golden, SmallNet, InjectGrad

pytestmark = pytest.mark.gpu
F32 = np.float32


def _cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.mark.parametrize("mode", ["max", "min"])
def test_pgd_update_kernel_vs_golden(mode):
    l = lib()
    z = load_golden(f"pgd_linf_elementwise_{mode}.npz")
    x = _cu(z["x"]); delta = _cu(z["delta0"]); vel = torch.zeros_like(x)
    flags = torch.zeros(1, dtype=torch.int32, device=dev())
    xadv = torch.zeros_like(x)
    for i in range(4):
        g = _cu(z["grads"][i])
        L.check(l.rvlm_pgd_linf_update(x.data_ptr(), g.data_ptr(), delta.data_ptr(), vel.data_ptr(), x.numel(),
                                       float(z["eps"]), float(z["stepsize"]), 0.9, int(mode == "max"),
                                       xadv.data_ptr(), flags.data_ptr(), st()))
        torch.cuda.synchronize()
        assert np.array_equal(xadv.cpu().numpy(), z["xadv"][i]), f"step {i}"
    assert int(flags.item()) == L.FLAG_NAN_GRAD      # NaNs were present (zeroed), nothing else tripped


@pytest.mark.parametrize("n", [1, 63, 4096, 150528 * 3 + 1])
def test_pgd_update_kernel_vs_c_oracle_random(n):
    l = lib()
    rng = np.random.default_rng(n)
    x = rng.random(n, dtype=F32)
    g = rng.standard_normal(n).astype(F32)
    g[rng.random(n) < 0.1] = 0.0
    g[rng.random(n) < 0.01] = np.nan
    delta = rng.uniform(-4 / 255, 4 / 255, n).astype(F32)
    vel = np.sign(rng.standard_normal(n)).astype(F32)
    eps, step = 4 / 255, 1 / 255
    dx, dg, dd, dv = _cu(x), _cu(g), _cu(delta), _cu(vel)
    for it in range(3):
        linf_c.pgd_linf_update(x, g, delta, vel, eps, step, 0.9, "max")
        L.check(l.rvlm_pgd_linf_update(dx.data_ptr(), dg.data_ptr(), dd.data_ptr(), dv.data_ptr(), n, eps, step,
                                       0.9, 1, None, None, st()))
        torch.cuda.synchronize()
        assert np.array_equal(dd.cpu().numpy(), delta) and np.array_equal(dv.cpu().numpy(), vel)
    assert np.abs(delta).max() <= F32(eps)


def test_range_flags():
    l = lib()
    x = torch.rand(1000, device=dev())
    flags = torch.zeros(1, dtype=torch.int32, device=dev())
    L.check(l.rvlm_check_image_range(x.data_ptr(), 1000, flags.data_ptr(), st()))
    assert int(flags.item()) == 0
    x[17] = 1.5
    L.check(l.rvlm_check_image_range(x.data_ptr(), 1000, flags.data_ptr(), st()))
    assert int(flags.item()) == L.FLAG_INPUT_RANGE


@pytest.mark.parametrize("n_iter", [10, 50, 100])
def test_apgd_kernels_vs_golden(n_iter):
    """HIP step/controller/select kernels replaying the reference's own loss / argmax / gradient traces
    (apgd_train_smallnet_*.npz): every x_adv iterate and the returned x_best_adv must be bit-identical
    to what train/apgd_train.py produced.  (Replay instead of re-evaluating the little network: the
    losses are compared at the last ulp and CPU BLAS rounding differs from host to host.)"""
    l = lib()
    z = load_golden(f"apgd_train_smallnet_{n_iter}.npz")
    x = np.ascontiguousarray(z["x"]); y = z["y"]
    B = x.shape[0]; npix = x[0].size; eps = float(z["eps"])
    from robustvlm_amd.apgd_train import apgd_schedule
    k, n_iter_min, size_decr = apgd_schedule(n_iter)
    dx = _cu(x); x_adv = _cu(np.clip(x, F32(0), F32(1))); x_old = x_adv.clone(); x_best = x_adv.clone()
    x_best_adv = x_adv.clone()
    grad = _cu(z["grads"][0]); grad_best = grad.clone()
    loss_best = _cu(z["losses"][0]); loss_best_lc = loss_best.clone(); reduced_lc = torch.ones(B, device=dev())
    step = torch.full((B,), float(F32(2.0 * eps)), device=dev())
    acc = _cu((z["argmax"][0] == y).astype(np.uint8))
    loss_steps = torch.zeros(n_iter, B, device=dev())
    f0 = torch.zeros(B, dtype=torch.uint8, device=dev()); f1 = f0.clone(); f2 = f0.clone()
    counter3 = 0
    assert np.array_equal(x_adv.cpu().numpy(), z["iterates"][0])
    for i in range(n_iter):
        L.check(l.rvlm_apgd_linf_step(dx.data_ptr(), x_adv.data_ptr(), x_old.data_ptr(), grad.data_ptr(),
                                      step.data_ptr(), 0.75 if i > 0 else 1.0, eps, npix, B, st()))
        torch.cuda.synchronize()
        assert np.array_equal(x_adv.cpu().numpy(), z["iterates"][i + 1]), f"iterate {i + 1}"
        if i < n_iter - 1:                                   # apgd_train.py:293-295 skips the last backward
            grad.copy_(_cu(z["grads"][i + 1]))
        pred = _cu((z["argmax"][i + 1] == y).astype(np.uint8))
        dl = _cu(z["losses"][i + 1])
        counter3 += 1
        do_check = int(counter3 == k)
        L.check(l.rvlm_apgd_controller(i, B, n_iter, k, do_check, dl.data_ptr(), pred.data_ptr(),
                                       loss_steps.data_ptr(), loss_best.data_ptr(), loss_best_lc.data_ptr(),
                                       reduced_lc.data_ptr(), step.data_ptr(), acc.data_ptr(), f0.data_ptr(),
                                       f1.data_ptr(), f2.data_ptr(), st()))
        L.check(l.rvlm_apgd_select(x_adv.data_ptr(), grad.data_ptr(), x_best.data_ptr(), grad_best.data_ptr(),
                                   x_best_adv.data_ptr(), f0.data_ptr(), f1.data_ptr(), f2.data_ptr(), npix, B,
                                   st()))
        if do_check:
            counter3 = 0
            k = max(k - size_decr, n_iter_min)
    torch.cuda.synchronize()
    assert np.array_equal(x_best_adv.cpu().numpy(), z["x_best_adv"])


def test_random_start_kernel():
    l = lib()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(3, 3, 8, 8, generator=g)
    t = 2 * torch.rand(x.shape, generator=g) - 1
    eps = 4 / 255
    tmax = t.abs().view(3, -1).max(1)[0].view(-1, 1, 1, 1)
    ref = x + eps * torch.ones_like(x) * (t / (tmax + 1e-12))
    out = torch.zeros_like(x, device=dev())
    xd, td = x.to(dev()), t.to(dev())          # keep the device copies alive across the launch
    L.check(l.rvlm_linf_random_start(xd.data_ptr(), td.data_ptr(), eps, 192, 3, out.data_ptr(), st()))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref.numpy())


@pytest.mark.parametrize("mode", ["max", "min"])
def test_generic_pgd_injected_gradients_bit_exact(mode):
    """The public pgd() (generic route: torch autograd + HIP update kernel) against the reference's
    own output for prescribed gradients."""
    z = load_golden(f"pgd_linf_elementwise_{mode}.npz")
    grads = [_cu(z["grads"][i]) for i in range(4)]
    it = iter(range(4))
    out = R.pgd(lambda v, output_normalize=False: v, lambda o, t: InjectGrad.apply(o, grads[next(it)]),
                _cu(z["x"]), None, "linf", float(z["eps"]), 4, float(z["stepsize"]), False,
                perturbation=_cu(z["delta0"]).requires_grad_(True), mode=mode)
    assert np.array_equal(out.cpu().numpy(), z["xadv"][3])


def test_pgd_error_behaviour():
    x = torch.rand(2, 3, 4, 4, device=dev())
    f = lambda v, output_normalize=False: v.flatten(1)  # noqa: E731
    lf = lambda o, t: o.sum()  # noqa: E731
    with pytest.raises(ValueError):
        R.pgd(f, lf, x, None, "linf", 4 / 255, 1, 1 / 255, False, mode="sideways")
    with pytest.raises(NotImplementedError):
        R.pgd(f, lf, x, None, "l7", 4 / 255, 1, 1 / 255, False, mode="max")
    with pytest.raises(AssertionError):
        R.pgd(f, lf, x + 2.0, None, "linf", 4 / 255, 1, 1 / 255, False, mode="max")
    with pytest.raises(L.RvlmError):
        R.pgd(f, lf, x.cpu(), None, "linf", 4 / 255, 1, 1 / 255, False, mode="max")   # no CPU fallback


@pytest.mark.parametrize("n_iter", [10, 50, 100])
def test_apgd_kernels_vs_c_oracle_state_by_state(n_iter):
    """Synthetic losses / predictions / gradients: every state array of the HIP controller+select+step
    kernels must equal the C oracle's after every iteration (pinpoints the first divergence)."""
    l = lib()
    rng = np.random.default_rng(n_iter)
    B, npix, eps = 7, 3 * 8 * 8, 8 / 255
    x = rng.random((B, npix), dtype=F32)
    from robustvlm_amd.apgd_train import apgd_schedule
    k, n_iter_min, size_decr = apgd_schedule(n_iter)
    x_adv = x.copy(); x_old = x.copy(); x_best = x.copy(); x_best_adv = x.copy()
    grad = rng.standard_normal((B, npix)).astype(F32); grad_best = grad.copy()
    loss0 = rng.random(B, dtype=F32)
    pred0 = (rng.random(B) < 0.7).astype(np.uint8)
    stc = linf_c.ApgdStateC(n_iter, loss0, np.full(B, F32(2.0 * eps), F32), pred0)
    d = {n: _cu(a) for n, a in dict(x=x, x_adv=x_adv, x_old=x_old, x_best=x_best, x_best_adv=x_best_adv, grad=grad,
                                   grad_best=grad_best, loss_best=loss0, loss_best_lc=loss0,
                                   reduced_lc=np.ones(B, F32), step=np.full(B, F32(2.0 * eps), F32), acc=pred0).items()}
    d["loss_steps"] = torch.zeros(n_iter, B, device=dev())
    for f in ("f0", "f1", "f2"):
        d[f] = torch.zeros(B, dtype=torch.uint8, device=dev())
    counter3 = 0
    loss = loss0.copy()
    for i in range(n_iter):
        a = 0.75 if i > 0 else 1.0
        linf_c.apgd_linf_step(x, x_adv, x_old, grad, stc.step, a, eps)
        L.check(l.rvlm_apgd_linf_step(d["x"].data_ptr(), d["x_adv"].data_ptr(), d["x_old"].data_ptr(),
                                      d["grad"].data_ptr(), d["step"].data_ptr(), a, eps, npix, B, st()))
        # slowly saturating noisy loss so that improvements, plateaus and oscillations all occur
        loss = (loss + rng.standard_normal(B).astype(F32) * F32(0.3 / (1 + i)) + F32(0.02)).astype(F32)
        if i % 7 == 3:
            loss[i % B] = stc.loss_best[i % B]           # exact tie with the best
        pred = (rng.random(B) < 0.6).astype(np.uint8)
        if i < n_iter - 1:
            g = rng.standard_normal((B, npix)).astype(F32)
            grad[...] = g
            d["grad"].copy_(_cu(g))
        dl, dp = _cu(loss), _cu(pred)
        counter3 += 1
        do_check = int(counter3 == k)
        stc.update(i, loss, pred, x_adv, grad, x_best, grad_best, x_best_adv)
        L.check(l.rvlm_apgd_controller(i, B, n_iter, k, do_check, dl.data_ptr(), dp.data_ptr(),
                                       d["loss_steps"].data_ptr(), d["loss_best"].data_ptr(),
                                       d["loss_best_lc"].data_ptr(), d["reduced_lc"].data_ptr(), d["step"].data_ptr(),
                                       d["acc"].data_ptr(), d["f0"].data_ptr(), d["f1"].data_ptr(), d["f2"].data_ptr(), st()))
        L.check(l.rvlm_apgd_select(d["x_adv"].data_ptr(), d["grad"].data_ptr(), d["x_best"].data_ptr(),
                                   d["grad_best"].data_ptr(), d["x_best_adv"].data_ptr(), d["f0"].data_ptr(),
                                   d["f1"].data_ptr(), d["f2"].data_ptr(), npix, B, st()))
        torch.cuda.synchronize()
        if do_check:
            counter3 = 0
            k = max(k - size_decr, n_iter_min)
        ref = dict(x_adv=x_adv, x_old=x_old, x_best=x_best, x_best_adv=x_best_adv, grad=grad, grad_best=grad_best,
                   loss_best=stc.loss_best, loss_best_lc=stc.loss_best_last_check, reduced_lc=stc.reduced_last_check,
                   step=stc.step, acc=stc.acc, f0=stc.f_notpred, f1=stc.f_improved, f2=stc.f_reduced,
                   loss_steps=stc.loss_steps)
        for name, r in ref.items():
            got = d[name].cpu().numpy().reshape(r.shape)
            if not np.array_equal(got, r):
                bad = np.argwhere(got != r)[:5]
                raise AssertionError(f"iteration {i} (do_check={do_check}, k={k}): {name} differs at {bad.tolist()} "
                                     f"got {got[tuple(bad[0])]} want {r[tuple(bad[0])]}")


# ------------------------------------------------------------------ section 8(f) rank 4: input front end
def test_preprocess_kernel_bit_exact_vs_pillow_golden():
    """csrc/preprocess.hip against Pillow's own output (tests/golden/preprocess_pil.npz) and, on larger / odd shapes,
    against the oracle restatement (itself pinned bit-exactly to Pillow): every pixel identical."""
    import robustvlm_amd as R
    from oracle import preprocess_ref as P
    from tests.helpers import load_golden
    z = load_golden("preprocess_pil.npz")
    tf = {}
    for i in range(int(z["n"])):
        size = int(z[f"size{i}"])
        tf.setdefault(size, R.ResizeCenterCropToTensor(size))
        got = tf[size](torch.from_numpy(z[f"img{i}"]).cuda()).cpu().numpy()
        want = z[f"crop{i}"].transpose(2, 0, 1).astype(np.float32) / np.float32(255)
        assert np.array_equal(got, want), f"golden case {i}"
    rng = np.random.default_rng(3)
    t224 = tf.setdefault(224, R.ResizeCenterCropToTensor(224))
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((375, 500), (500, 333), (224, 224), (100, 150),
                                                                          (1200, 1600), (225, 1000), (375, 500))]
    batch = t224.batch([torch.from_numpy(a).cuda() for a in imgs]).cpu().numpy()
    for a, b in zip(imgs, batch):
        assert np.array_equal(b, P.preprocess_ref(a, 224)), a.shape
    # the batch is ONE launch (rvlm_preproc_run_batch: blockIdx.z = image, every image its own shape / tables / tile
    # height); it must equal the one-image calls bit for bit, also when the staging buffers are reused and regrown
    dimgs = [torch.from_numpy(a).cuda() for a in imgs]
    single = torch.stack([t224(d) for d in dimgs])
    assert torch.equal(torch.from_numpy(batch).cuda(), single)
    big = dimgs * 19                                             # 133 images, 6 distinct shapes
    again = t224.batch(big)
    assert torch.equal(again[:7], single) and torch.equal(again[-7:], single)
    assert torch.equal(t224.batch(dimgs[:2]), single[:2])
    with pytest.raises(ValueError):
        t224.batch([])
    with pytest.raises(ValueError):
        t224(torch.zeros(4, 4, 3).cuda())
    small = R.ResizeCenterCropToTensor(32, max_input_dim=64)
    with pytest.raises(NotImplementedError):
        small(torch.zeros(1000, 1000, 3, dtype=torch.uint8).cuda())   # 31x down-scaling needs more taps than provisioned
    with pytest.raises(NotImplementedError):
        small.batch([torch.zeros(40, 40, 3, dtype=torch.uint8).cuda(), torch.zeros(1000, 1000, 3, dtype=torch.uint8).cuda()])


# ---- L2-norm branch of pgd as a device kernel (rvlm_pgd_l2_update), against the reference's own outputs
@pytest.mark.parametrize("mode", ["max", "min"])
def test_pgd_l2_update_kernel_vs_golden(mode):
    """Per-sample normalise / momentum / renorm / clamp (train/pgd_train.py:38-63 with utils.py:12-14,22-26).  The norms
    are fp32 sums in the kernel's own order: equal to the reference to fp32 rounding of three norms per step (tolerance
    2e-6 absolute on pixel values in [0,1]; the all-zero-gradient sample and the NaN entries must behave identically)."""
    l = lib()
    z = load_golden("pgd_l2norm.npz")
    x = _cu(z["ew_x"]); delta = _cu(z["ew_delta0"]); vel = torch.zeros_like(x)
    B, npix = x.shape[0], x[0].numel()
    flags = torch.zeros(1, dtype=torch.int32, device=dev())
    xadv = torch.zeros_like(x)
    eps, step = float(z["ew_eps"]), float(z["ew_stepsize"])
    for i in range(4):
        g = _cu(z["ew_grads"][i])
        L.check(l.rvlm_pgd_l2_update(x.data_ptr(), g.data_ptr(), delta.data_ptr(), vel.data_ptr(), npix, B, eps, step,
                                     0.9, int(mode == "max"), xadv.data_ptr(), flags.data_ptr(), st()))
        torch.cuda.synchronize()
        got, want = xadv.cpu().numpy(), z[f"ew_xadv_{mode}"][i]
        assert np.abs(got - want).max() < 2e-6, (i, np.abs(got - want).max())
        assert np.array_equal(got[2], want[2])                   # zero gradient, zero velocity: the start point, clamped
    assert int(flags.item()) == L.FLAG_NAN_GRAD
    dn = (xadv - x).flatten(1).norm(dim=1)
    assert float(dn.max()) <= eps * (1 + 1e-5)
    # deterministic
    d2 = _cu(z["ew_delta0"]); v2 = torch.zeros_like(x); x2 = torch.zeros_like(x)
    for i in range(4):
        L.check(l.rvlm_pgd_l2_update(x.data_ptr(), _cu(z["ew_grads"][i]).data_ptr(), d2.data_ptr(), v2.data_ptr(), npix, B,
                                     eps, step, 0.9, int(mode == "max"), x2.data_ptr(), None, st()))
    torch.cuda.synchronize()
    assert torch.equal(x2, xadv)


def test_pgd_l2norm_fused_and_generic_vs_reference_golden():
    """pgd(norm='l2') through the fused route (rvlm_pgd_run_norm) and the generic route (autograd + rvlm_pgd_l2_update)
    on the fp32 engine against the reference's x_adv on the same tiny ViT."""
    from oracle import vit_ref as V
    from tests.helpers import cfg_from_array, weights_from_golden
    zt = load_golden("tiny_vit_attacks.npz")
    g = load_golden("pgd_l2norm.npz")
    cfg = cfg_from_array(zt["cfg"])
    w = weights_from_golden(zt)
    eng = R.VitEngine(R.VitConfig(cfg.image_size, cfg.patch, cfg.width, cfg.layers, cfg.heads, cfg.out_dim, cfg.act),
                      {k: v.to(dev()) for k, v in w.items()}, precision="fp32", max_batch=8)
    model = R.ClipVisionModel(eng).eval()
    x, d0 = _cu(zt["x"]), _cu(zt["delta0"])
    e0 = model(x, False)
    wrap = R.ComputeLossWrapper(e0, None, "mean", "l2", 100.)
    eps, step = float(g["vit_eps"]), float(g["vit_stepsize"])
    fused = R.pgd(model, wrap, x, None, "l2", eps, 10, step, False, perturbation=d0.clone(), mode="max")
    generic = R.pgd(lambda v, output_normalize: model(v, output_normalize), wrap, x, None, "l2", eps, 10, step, False,
                    perturbation=d0.clone().requires_grad_(True), mode="max")
    want = g["vit_xadv"]
    for name, got in (("fused", fused), ("generic", generic)):
        diff = np.abs(got.cpu().numpy() - want).max()
        assert diff < 1e-4, (name, diff)                          # fp32 encoders differ at 1e-6 relative; 10 steps
        assert float((got - x).flatten(1).norm(dim=1).max()) <= eps * (1 + 1e-5)
    lf = float(wrap(model(fused, False), None))
    assert abs(lf - float(g["vit_loss_final"])) <= 1e-3 * abs(float(g["vit_loss_final"]))
    eng.close()


def test_apgd_l2_step_kernel_vs_oracle():
    """One APGD L2 step (apgd_train.py:231-254) on random data with mixed per-sample step sizes, first step (a = 1) and a
    later one (a = 0.75), against the oracle's torch restatement; tolerance = fp32 rounding of the three norms."""
    from oracle import attacks_ref as A
    l = lib()
    rng = np.random.default_rng(5)
    B, shape = 6, (6, 3, 16, 16)
    x = rng.random(shape, dtype=F32)
    x_adv = np.clip(x + rng.normal(0, 0.02, shape).astype(F32), 0, 1).astype(F32)
    x_old = np.clip(x + rng.normal(0, 0.02, shape).astype(F32), 0, 1).astype(F32)
    grad = rng.standard_normal(shape).astype(F32)
    grad[3] = 0.0                                               # zero gradient: the 1e-12 guards
    step = np.array([1.0, 0.5, 0.25, 2.0, 0.125, 1.0], dtype=F32)
    eps = 0.5
    for a in (1.0, 0.75):
        want, want_old = A.apgd_l2_step_ref(x, x_adv, x_old, grad, step.reshape(B, 1, 1, 1), a, eps)
        dx, da, do, dg, ds = _cu(x), _cu(x_adv), _cu(x_old), _cu(grad), _cu(step)
        L.check(l.rvlm_apgd_l2_step(dx.data_ptr(), da.data_ptr(), do.data_ptr(), dg.data_ptr(), ds.data_ptr(), a, eps,
                                    x[0].size, B, st()))
        torch.cuda.synchronize()
        assert np.abs(da.cpu().numpy() - want).max() < 2e-6
        assert np.array_equal(do.cpu().numpy(), want_old)
        assert float((da - dx).flatten(1).norm(dim=1).max()) <= eps * (1 + 1e-5)


@pytest.mark.parametrize("loss_name", ["l2", "ce"])
def test_apgd_train_l2norm_fused_and_generic_vs_reference_golden(loss_name):
    """apgd_train(norm='l2') through rvlm_apgd_run_norm (fused) and through autograd + rvlm_apgd_l2_step (generic) on the
    fp32 engine against the reference's x_best_adv on the same tiny ViT."""
    from tests.helpers import cfg_from_array, weights_from_golden
    zt = load_golden("tiny_vit_attacks.npz")
    g = load_golden("pgd_l2norm.npz")
    cfg = cfg_from_array(zt["cfg"])
    w = weights_from_golden(zt)
    eng = R.VitEngine(R.VitConfig(cfg.image_size, cfg.patch, cfg.width, cfg.layers, cfg.heads, cfg.out_dim, cfg.act),
                      {k: v.to(dev()) for k, v in w.items()}, precision="fp32", max_batch=8)
    model = R.ClipVisionModel(eng).eval()
    x, y, T, e0 = _cu(zt["x"]), _cu(zt["y"]), _cu(zt["T"]), _cu(g["apgd_e0"])
    wrap = R.ComputeLossWrapper(e0, T, "none", loss_name, 100.)
    fused = R.apgd_train(model, x, y, "l2", 1.0, n_iter=10, loss_fn=wrap)

    class Plain(torch.nn.Module):          # not a ClipVisionModel: takes the generic route
        def forward(self, v, output_normalize=True):
            return model(v, output_normalize)
    generic = R.apgd_train(Plain().eval(), x, y, "l2", 1.0, n_iter=10, loss_fn=wrap)
    want = g[f"apgd_{loss_name}_10_xadv"]
    for name, got in (("fused", fused), ("generic", generic)):
        diff = np.abs(got.cpu().numpy() - want).max()
        assert diff < 2e-4, (name, diff)
        assert float((got - x).flatten(1).norm(dim=1).max()) <= 1.0 + 1e-5
    with torch.no_grad():
        lf = wrap(model(fused, True), y).cpu().numpy()
    np.testing.assert_allclose(lf, g[f"apgd_{loss_name}_10_loss_final"], rtol=2e-3, atol=1e-6)
    eng.close()


def test_standalone_attack_utils_vs_torch():
    """robustvlm_amd.project_perturbation / normalize_grad (vlm_eval/attacks/utils.py:8-26) are device kernels of librvlm:
    L-inf branches bit-equal to torch.clamp / torch.sign (NaN, signed zeros), L2 branches equal to torch.renorm /
    F.normalize to fp32 rounding of the per-sample norm."""
    g = torch.Generator(device="cuda").manual_seed(3)
    t = torch.randn(5, 3, 16, 16, generator=g, device=dev())
    t.view(-1)[::17] = 0.0
    t.view(-1)[5::29] = -0.0
    t.view(-1)[7::31] = float("nan")
    eps = 0.3
    assert torch.equal(R.normalize_grad(t, "linf"), t.sign())
    got, want = R.project_perturbation(t, eps, "linf"), torch.clamp(t, -eps, eps)
    assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(want, nan=7.0))
    u = torch.randn(5, 3, 16, 16, generator=g, device=dev())
    u[1] *= 0.001                                          # inside the ball: untouched by renorm
    u[2] = 0.0                                             # zero row: F.normalize's eps path
    for name, got, want in (("normalize", R.normalize_grad(u, "l2"),
                             torch.nn.functional.normalize(u.view(5, -1), p=2, dim=1).view_as(u)),
                            ("renorm", R.project_perturbation(u, eps, 2), torch.renorm(u, p=2, dim=0, maxnorm=eps))):
        assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max() + 1e-30) + 1e-9, name
    assert torch.equal(R.project_perturbation(u, eps, "l2")[1], u[1])
    with pytest.raises(NotImplementedError):
        R.normalize_grad(u, "l1")
    with pytest.raises(L.RvlmError):
        R.project_perturbation(u.cpu(), eps, "linf")


# ------------------------------------------------------------------ round 4: APGDAttack's `rho`
@pytest.mark.parametrize("rho", [0.3, 0.5, 0.75, 0.9])
def test_apgd_controller_rho_vs_oracle(rho):
    """rvlm_apgd_controller_rho against the numpy controller (oracle/attacks_ref.py::ApgdController, whose rho path is
    pinned by tests/golden/autopgd_tiny_rho*.npz): oscillation threshold `k * rho` evaluated like check_oscillation
    (autopgd_base.py:170-175); per-sample step sizes, bests and reduce flags bit for bit over 50 iterations."""
    from oracle.attacks_ref import ApgdController, apgd_schedule
    l = lib()
    rng = np.random.default_rng(7)
    B, n_iter, npix = 96, 50, 8
    loss0 = rng.random(B, dtype=F32)
    ctl = ApgdController(n_iter, loss0, np.full(B, F32(0.03), F32), rho=rho)
    d = dict(loss_steps=torch.zeros(n_iter, B, device=dev()), loss_best=_cu(loss0), loss_best_lc=_cu(loss0),
             reduced_lc=torch.ones(B, device=dev()), step=torch.full((B,), 0.03, device=dev()),
             acc=torch.ones(B, dtype=torch.uint8, device=dev()))
    for f in ("f0", "f1", "f2"):
        d[f] = torch.zeros(B, dtype=torch.uint8, device=dev())
    pred = torch.ones(B, dtype=torch.uint8, device=dev())
    xa = rng.random((B, npix), dtype=F32); g = xa.copy(); xb = xa.copy(); gb = xa.copy()
    loss = loss0.copy()
    k, n_iter_min, size_decr = apgd_schedule(n_iter)
    counter3, n_red = 0, 0
    for i in range(n_iter):
        # noisy, slowly rising losses: the count of increases over a window lands on both sides of k * rho
        loss = (loss + rng.standard_normal(B).astype(F32) * F32(0.2) + F32(0.03)).astype(F32)
        counter3 += 1
        do_check = int(counter3 == k)
        red = ctl.update(i, loss, xa, g, xb, gb)
        L.check(l.rvlm_apgd_controller_rho(i, B, n_iter, k, do_check, float(rho), _cu(loss).data_ptr(), pred.data_ptr(),
                                           d["loss_steps"].data_ptr(), d["loss_best"].data_ptr(),
                                           d["loss_best_lc"].data_ptr(), d["reduced_lc"].data_ptr(),
                                           d["step"].data_ptr(), d["acc"].data_ptr(), d["f0"].data_ptr(),
                                           d["f1"].data_ptr(), d["f2"].data_ptr(), st()))
        torch.cuda.synchronize()
        if do_check:
            counter3 = 0
            k = max(k - size_decr, n_iter_min)
            n_red += int(d["f2"].sum())
        assert k == ctl.k
        for name, want in (("step", ctl.step), ("loss_best", ctl.loss_best), ("reduced_lc", ctl.reduced_last_check),
                           ("loss_best_lc", ctl.loss_best_last_check)):
            assert np.array_equal(d[name].cpu().numpy(), want), (i, name)
    assert n_red > 0


@pytest.mark.parametrize("tag", ["rho050", "rho090"])
def test_apgdattack_rho_vs_reference_golden(tag):
    """APGDAttack(rho != .75) through the fused device loop (rvlm_vit_set_apgd_rho) against the reference's own result."""
    from oracle import vit_ref as V
    from tests.test_gpu_engine import make_engine
    z = load_golden(f"autopgd_tiny_{tag}.npz")
    cfg = V.VIT_TINY
    w = V.init_weights(cfg, seed=int(z["weights_seed"]))
    eng = make_engine(cfg, w, "fp32")
    clf = R.ClassificationModel(eng, torch.from_numpy(z["T"]).to(dev())).eval()
    x, y = torch.from_numpy(z["x"]).to(dev()), torch.from_numpy(z["y"]).to(dev())
    kw = dict(n_iter=int(z["n_iter"]), norm="Linf", n_restarts=1, eps=float(z["eps"]), seed=0, loss="ce", use_rs=True)
    adv = R.APGDAttack(clf, rho=float(z["rho"]), **kw).perturb(x, y).cpu().numpy()
    other = R.APGDAttack(clf, rho=0.75, **kw).perturb(x, y).cpu().numpy()
    same, same_other = np.mean(adv == z["adv"]), np.mean(other == z["adv"])
    # fp32 engine vs the CPU reference flips a few near-zero gradient signs either way; the run with the RIGHT threshold
    # is still the closer one (by a wide margin at rho = 0.5, where 38 % of the reference's pixels depend on it)
    assert same > 0.85 and same > same_other + (0.03 if tag == "rho050" else 0.0), (same, same_other)
    eng.close()


def test_apgd_train_ignores_a_rho_left_on_the_handle():
    """rvlm_vit_set_apgd_rho is handle state; train/apgd_train.py hard-codes 0.75 (:117,334), so a C host that ran
    APGDAttack(rho = 0.5) and then apgd_train on the same handle must still get the 0.75 schedule."""
    import ctypes as C
    from oracle import vit_ref as V
    from tests.test_gpu_engine import make_engine
    z = load_golden("autopgd_tiny_rho050.npz")
    cfg = V.VIT_TINY
    w = V.init_weights(cfg, seed=int(z["weights_seed"]))
    eng = make_engine(cfg, w, "fp32")
    model = R.ClipVisionModel(eng).eval()
    x, y = torch.from_numpy(z["x"]).to(dev()), torch.from_numpy(z["y"]).to(dev())
    T = torch.from_numpy(z["T"]).to(dev())
    wrap = R.ComputeLossWrapper(None, T, "none", "ce", 100.)
    run = lambda: R.apgd_train(model, x, y, "linf", float(z["eps"]), n_iter=30, loss_fn=wrap)     # noqa: E731
    want = run()
    L.check(eng.lib.rvlm_vit_set_apgd_rho(eng._h, C.c_double(0.5)))       # behind the Python mirror's back, like a C host
    assert torch.equal(run(), want)
    L.check(eng.lib.rvlm_vit_set_apgd_rho(eng._h, C.c_double(0.75)))
    eng.close()
