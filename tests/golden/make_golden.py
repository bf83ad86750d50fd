#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build container.

Run once from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (read-only) and therefore never runs on the GPU box; only its outputs
(small .npz files = inputs + expected outputs) are committed.

What is executed from the reference (nothing is copied into this repo):
  * train.pgd_train.pgd, train.apgd_train.apgd_train, vlm_eval.attacks.utils.*,
    autoattack.autopgd_base.APGDAttack / APGDAttack_targeted, autoattack.AutoAttack, autoattack.square.SquareAttack
                                                - imported from /root/reference
  * l2 / ce / compute_loss / ComputeLossWrapper / compute_acc of
    train/adversarial_training_clip.py          - that module needs torchvision/open_clip/wandb to
    import, so the function/class definitions are pulled out of the file with ``ast`` at run time
    and exec'd here (the reference's own code, run as is).
  * the ViT the reference wraps is third-party (open-clip-torch==2.19.0, not in the tree): its
    arithmetic is pinned against HF transformers' CLIPVisionModelWithProjection (independent
    implementation of the same architecture), loaded with the same seeded weights.
"""
import ast
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import vit_ref  # noqa: E402
from oracle.vit_ref import VitConfig, ClipVisionModelRef, init_weights  # noqa: E402

from train.pgd_train import pgd as ref_pgd  # noqa: E402
from train.apgd_train import apgd_train as ref_apgd_train  # noqa: E402
from autoattack.autopgd_base import APGDAttack as RefAPGDAttack  # noqa: E402

torch.set_num_threads(4)
torch.use_deterministic_algorithms(True)


def extract_reference_losses():
    """exec the loss definitions of train/adversarial_training_clip.py without importing it."""
    path = os.path.join(REF, "train", "adversarial_training_clip.py")
    tree = ast.parse(open(path).read())
    want = {"l2", "ce", "compute_loss", "ComputeLossWrapper", "compute_acc"}
    nodes = [n for n in tree.body
             if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in want]
    assert {n.name for n in nodes} == want
    ns = {"torch": torch, "F": torch.nn.functional}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, "exec"), ns)
    return ns


REFL = extract_reference_losses()


def weights_digest(w: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(w[k].numpy().tobytes())
    return h.hexdigest()


def save(name, **arrs):
    p = os.path.join(OUT, name)
    np.savez_compressed(p, **arrs)
    print(f"wrote {name}: {os.path.getsize(p)/1024:.1f} KiB")


# ------------------------------------------------------------------ G1: pgd Linf elementwise
def g1_pgd_elementwise():
    g = torch.Generator().manual_seed(101)
    shape = (4, 3, 8, 8)
    eps, step = 4 / 255, 1 / 255
    x = torch.rand(shape, generator=g)
    x.view(-1)[:16] = torch.tensor([0., 1., 0., 1., 0.5, 0.999, 0.001, 1., 0., 0., 1., 1.,
                                    4 / 255, 1 - 4 / 255, 2 / 255, 1 - 2 / 255])
    d0 = torch.zeros(shape).uniform_(-eps, eps, generator=g)
    grads = []
    for i in range(4):
        gi = torch.randn(shape, generator=g)
        gi.view(-1)[i::7] = 0.0           # exact zeros -> momentum keeps the previous sign
        gi.view(-1)[3 + i::29] = float("nan")
        gi.view(-1)[5::31] = -0.0
        gi.view(-1)[11::37] *= 1e-30
        grads.append(gi)
    class Inject(torch.autograd.Function):
        """scalar 'loss' whose gradient w.r.t. its input is exactly the prescribed tensor G
        (NaNs and signed zeros included)."""

        @staticmethod
        def forward(ctx, v, G):
            ctx.save_for_backward(G)
            return v.sum() * 0.0 + 1.0

        @staticmethod
        def backward(ctx, go):
            (G,) = ctx.saved_tensors
            return G.clone(), None

    def fwd(v, output_normalize=False):
        return v

    for mode in ("max", "min"):
        deltas = []
        for n_it in range(1, 5):
            it = iter(range(n_it))

            def loss_fn(out, targets):
                return Inject.apply(out, grads[next(it)])

            xadv = ref_pgd(fwd, loss_fn, x, None, "linf", eps, n_it, step, False,
                           perturbation=d0.clone().requires_grad_(True), mode=mode)
            deltas.append((xadv - x).numpy())
        save(f"pgd_linf_elementwise_{mode}.npz", x=x.numpy(), delta0=d0.numpy(),
             grads=np.stack([t.numpy() for t in grads]),
             xadv_minus_x=np.stack(deltas),
             xadv=np.stack([(x + torch.from_numpy(d)).numpy() for d in deltas]),
             eps=np.float64(eps), stepsize=np.float64(step))


# ------------------------------------------------------------------ G2: apgd controller traces
class SmallNet(torch.nn.Module):
    """tanh-MLP; records every input it sees, the gradient that flows back to it, and its argmax."""

    def __init__(self, seed, d_in, n_cls=10, hidden=32, sharp=6.0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w1 = torch.randn(d_in, hidden, generator=g) * sharp / d_in ** 0.5
        self.w2 = torch.randn(hidden, n_cls, generator=g) * sharp / hidden ** 0.5
        self.seen, self.grads, self.argmax = [], [], []

    def forward(self, x, output_normalize=True):
        self.seen.append(x.detach().clone())
        if x.requires_grad:
            x.register_hook(lambda g: self.grads.append(g.detach().clone()))
        out = torch.tanh(x.flatten(1) @ self.w1) @ self.w2
        self.argmax.append(out.detach().max(1)[1].clone())
        return out


def g2_apgd_controller():
    for n_iter in (10, 50, 100):
        g = torch.Generator().manual_seed(200 + n_iter)
        shape = (6, 3, 8, 8)
        x = torch.rand(shape, generator=g)
        y = torch.randint(0, 10, (6,), generator=g)
        net = SmallNet(7 + n_iter, 3 * 8 * 8).eval()
        losses = []

        def ce(lg, yy):
            l = torch.nn.functional.cross_entropy(lg, yy, reduction="none")
            losses.append(l.detach().clone())
            return l

        out = ref_apgd_train(net, x, y, "linf", 8 / 255, n_iter=n_iter, loss_fn=ce)
        # traces (model input k -> loss k, argmax k, and - except for the last call - gradient k) let the
        # device kernels be replayed without re-evaluating the network (CPU BLAS rounding differs by host)
        save(f"apgd_train_smallnet_{n_iter}.npz", x=x.numpy(), y=y.numpy(),
             w1=net.w1.numpy(), w2=net.w2.numpy(), eps=np.float64(8 / 255),
             iterates=np.stack([t.numpy() for t in net.seen]), x_best_adv=out.numpy(),
             losses=np.stack([t.numpy() for t in losses]),
             argmax=np.stack([t.numpy() for t in net.argmax]),
             grads=np.stack([t.numpy() for t in net.grads]))


# ------------------------------------------------------------------ G3: end to end on a tiny ViT
def g3_tiny_vit_attacks():
    cfg = vit_ref.VIT_TINY
    w = init_weights(cfg, seed=3)
    model = ClipVisionModelRef(cfg, w).eval()
    g = torch.Generator().manual_seed(33)
    B = 4
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    T = torch.randn(cfg.out_dim, 10, generator=g)
    T = T / T.norm(dim=0, keepdim=True)
    eps, step = 4 / 255, 1 / 255
    d0 = torch.zeros_like(x).uniform_(-eps, eps, generator=g)
    arrs = dict(x=x.numpy(), y=y.numpy(), T=T.numpy(), delta0=d0.numpy(),
                eps=np.float64(eps), stepsize=np.float64(step),
                cfg=np.array([cfg.image_size, cfg.patch, cfg.width, cfg.layers, cfg.heads,
                              cfg.out_dim]),
                weights_seed=np.int64(3), weights_sha256=np.array(weights_digest(w)))
    for k, v in w.items():
        arrs["w::" + k] = v.numpy()
    with torch.no_grad():
        for on in (False, True):
            arrs[f"e0_norm{int(on)}"] = model(x, on).numpy()
    # FARE: pgd + l2/mean, output_normalize False (README.md:282)
    for loss_name, on in (("l2", False), ("ce", True)):
        with torch.no_grad():
            e0 = model(x, on)
        wrap = REFL["ComputeLossWrapper"](e0, T, "mean", loss_name, 100.)
        xadv = ref_pgd(model, wrap, x, y, "linf", eps, 10, step, on,
                       perturbation=d0.clone().requires_grad_(True), mode="max")
        arrs[f"pgd_{loss_name}_xadv"] = xadv.numpy()
        with torch.no_grad():
            arrs[f"pgd_{loss_name}_loss_final"] = np.float32(wrap(model(xadv, on), y).item())
        # apgd: reduction none, output_normalize hard-wired True (apgd_train.py:181,288)
        with torch.no_grad():
            e0n = model(x, True)
        wrapn = REFL["ComputeLossWrapper"](e0n, T, "none", loss_name, 100.)
        xadv = ref_apgd_train(model, x, y, "linf", eps, n_iter=10, loss_fn=wrapn)
        arrs[f"apgd_{loss_name}_xadv"] = xadv.numpy()
        with torch.no_grad():
            arrs[f"apgd_{loss_name}_loss_final"] = wrapn(model(xadv, True), y).numpy()
    save("tiny_vit_attacks.npz", **arrs)


# ------------------------------------------------------------------ G4: APGDAttack on tiny ViT + head
def g4_autopgd():
    cfg = vit_ref.VIT_TINY
    w = init_weights(cfg, seed=3)
    g = torch.Generator().manual_seed(44)
    B = 6
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g)
    T = torch.randn(cfg.out_dim, 10, generator=g)
    T = T / T.norm(dim=0, keepdim=True)
    clf = vit_ref.ClassificationModelRef(cfg, w, T, 100.0).eval()
    with torch.no_grad():
        y = clf(x).max(1)[1]          # clean predictions -> all samples start "correct"
    y[0] = (y[0] + 1) % 10            # one sample starts misclassified (never attacked)
    seen = []

    def predict(v):
        seen.append(v.detach().clone())
        return clf(v)

    for n_restarts in (1, 2):
        seen.clear()
        atk = RefAPGDAttack(predict, n_iter=12, norm="Linf", n_restarts=n_restarts, eps=4 / 255,
                            seed=0, loss="ce", device="cpu", alpha=2.0, use_rs=True)
        adv = atk.perturb(x.clone(), y.clone())
        save(f"autopgd_tiny_r{n_restarts}.npz", x=x.numpy(), y=y.numpy(), T=T.numpy(),
             adv=adv.detach().numpy(), eps=np.float64(4 / 255), n_iter=np.int64(12),
             first_start=seen[1].numpy(),       # seen[0] = clean pass, seen[1] = clamped random start
             n_model_calls=np.int64(len(seen)), weights_seed=np.int64(3),
             weights_sha256=np.array(weights_digest(w)))
    # `rho` (the oscillation threshold, autopgd_base.py:111,137,415-416) away from its default: more iterations so that
    # several checkpoints fall inside the run, two thresholds on either side of 0.75, and a radius at which the losses
    # do oscillate (at 4/255 they rise monotonically and rho changes nothing; at 0.15: 38 % / 8 % of the pixels of the
    # result differ from the rho = 0.75 run)
    eps_rho = 0.15
    for rho in (0.5, 0.9):
        seen.clear()
        atk = RefAPGDAttack(predict, n_iter=30, norm="Linf", n_restarts=1, eps=eps_rho, seed=0, loss="ce",
                            device="cpu", rho=rho, use_rs=True)
        adv = atk.perturb(x.clone(), y.clone())
        save(f"autopgd_tiny_rho{int(rho * 100):03d}.npz", x=x.numpy(), y=y.numpy(), T=T.numpy(),
             adv=adv.detach().numpy(), eps=np.float64(eps_rho), n_iter=np.int64(30), rho=np.float64(rho),
             first_start=seen[1].numpy(), n_model_calls=np.int64(len(seen)), weights_seed=np.int64(3),
             weights_sha256=np.array(weights_digest(w)))


# ------------------------------------------------------------------ G5: ViT vs HF transformers
def openclip_to_hf(w, cfg):
    sd = {}
    e = "vision_model.embeddings."
    sd[e + "patch_embedding.weight"] = w["conv1.weight"]
    sd[e + "class_embedding"] = w["class_embedding"]
    sd[e + "position_embedding.weight"] = w["positional_embedding"]
    sd["vision_model.pre_layrnorm.weight"] = w["ln_pre.weight"]
    sd["vision_model.pre_layrnorm.bias"] = w["ln_pre.bias"]
    sd["vision_model.post_layernorm.weight"] = w["ln_post.weight"]
    sd["vision_model.post_layernorm.bias"] = w["ln_post.bias"]
    sd["visual_projection.weight"] = w["proj"].t().contiguous()
    W = cfg.width
    for i in range(cfg.layers):
        s = f"vision_model.encoder.layers.{i}."
        p = f"transformer.resblocks.{i}."
        sd[s + "layer_norm1.weight"] = w[p + "ln_1.weight"]
        sd[s + "layer_norm1.bias"] = w[p + "ln_1.bias"]
        sd[s + "layer_norm2.weight"] = w[p + "ln_2.weight"]
        sd[s + "layer_norm2.bias"] = w[p + "ln_2.bias"]
        for j, n in enumerate("qkv"):
            sd[s + f"self_attn.{n}_proj.weight"] = w[p + "attn.in_proj_weight"][j * W:(j + 1) * W]
            sd[s + f"self_attn.{n}_proj.bias"] = w[p + "attn.in_proj_bias"][j * W:(j + 1) * W]
        sd[s + "self_attn.out_proj.weight"] = w[p + "attn.out_proj.weight"]
        sd[s + "self_attn.out_proj.bias"] = w[p + "attn.out_proj.bias"]
        sd[s + "mlp.fc1.weight"] = w[p + "mlp.c_fc.weight"]
        sd[s + "mlp.fc1.bias"] = w[p + "mlp.c_fc.bias"]
        sd[s + "mlp.fc2.weight"] = w[p + "mlp.c_proj.weight"]
        sd[s + "mlp.fc2.bias"] = w[p + "mlp.c_proj.bias"]
    return sd


def g5_vit_vs_hf():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    for name, cfg, B, seed in (("tiny2", vit_ref.VIT_TINY2, 2, 5),
                               ("tiny2gelu", VitConfig(96, 16, 128, 3, 2, 48, "gelu"), 2, 5),
                               ("b32", vit_ref.VIT_B_32, 1, 6)):
        w = init_weights(cfg, seed=seed)
        hcfg = CLIPVisionConfig(hidden_size=cfg.width, intermediate_size=cfg.mlp,
                                num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                                image_size=cfg.image_size, patch_size=cfg.patch,
                                projection_dim=cfg.out_dim, hidden_act=cfg.act,
                                layer_norm_eps=1e-5, attn_implementation="eager")
        hf = CLIPVisionModelWithProjection(hcfg).eval()
        missing = hf.load_state_dict(openclip_to_hf(w, cfg), strict=False)
        assert not missing.unexpected_keys, missing
        assert all("position_ids" in k for k in missing.missing_keys), missing
        g = torch.Generator().manual_seed(55)
        x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g)
        xn = vit_ref.normalize_pixels(x).requires_grad_(True)
        emb = hf(pixel_values=xn).image_embeds
        cot = torch.randn(emb.shape, generator=g)
        (gx,) = torch.autograd.grad((emb * cot).sum(), xn)
        save(f"vit_hf_{name}.npz", x=x.numpy(), emb=emb.detach().numpy(), cot=cot.numpy(),
             grad_xn=gx.numpy(), weights_seed=np.int64(seed),
             weights_sha256=np.array(weights_digest(w)),
             cfg=np.array([cfg.image_size, cfg.patch, cfg.width, cfg.layers, cfg.heads,
                           cfg.out_dim]), act=np.array(cfg.act))


# ------------------------------------------------------------------ G6: losses
def g6_losses():
    g = torch.Generator().manual_seed(66)
    B, D, C = 5, 32, 10
    emb = torch.randn(B, D, generator=g)
    e0 = torch.randn(B, D, generator=g)
    T = torch.randn(D, C, generator=g)
    T = T / T.norm(dim=0, keepdim=True)
    y = torch.randint(0, C, (B,), generator=g)
    arrs = dict(emb=emb.numpy(), e0=e0.numpy(), T=T.numpy(), y=y.numpy())
    for loss in ("l2", "ce"):
        for red in ("mean", "none"):
            e = emb.clone().requires_grad_(True)
            val = REFL["compute_loss"](loss, e, y, e0, 100., T, red)
            (ge,) = torch.autograd.grad(val.sum(), e)
            arrs[f"{loss}_{red}"] = val.detach().numpy()
            arrs[f"{loss}_{red}_grad"] = ge.numpy()
    logits = emb @ (100. * T)
    arrs["acc"] = np.float64(REFL["compute_acc"](logits, y))
    save("losses.npz", **arrs)


# ------------------------------------------------------------------ G7: AutoAttack orchestration (section 8(f) rank 3)
def g7_autoattack():
    """DLR losses on random logits (values + autograd gradients), APGDAttack_targeted.perturb and
    AutoAttack(version='custom', ['apgd-ce', 'apgd-t']).run_standard_evaluation on the tiny ViT + 10-class head."""
    from autoattack.autopgd_base import APGDAttack_targeted as RefTargeted
    from autoattack import AutoAttack as RefAutoAttack
    g = torch.Generator().manual_seed(77)
    # (a) loss functions
    B, C = 9, 12
    logits = torch.randn(B, C, generator=g) * 3
    y = torch.randint(0, C, (B,), generator=g)
    y[:3] = logits[:3].argmax(1)                       # some rows correctly classified (ind = 1 branch)
    yt = torch.randint(0, C, (B,), generator=g)
    yt = torch.where(yt == y, (yt + 1) % C, yt)
    obj = RefTargeted(lambda v: v, eps=0.1, n_iter=5, device="cpu")
    res = {}
    lg = logits.clone().requires_grad_(True)
    l_u = obj.dlr_loss(lg, y)
    g_u, = torch.autograd.grad(l_u.sum(), lg)
    lg = logits.clone().requires_grad_(True)
    obj.y_target = yt
    l_t = obj.dlr_loss_targeted(lg, y)
    g_t, = torch.autograd.grad(l_t.sum(), lg)
    save("dlr_losses.npz", logits=logits.numpy(), y=y.numpy(), y_target=yt.numpy(), dlr=l_u.detach().numpy(),
         dlr_grad=g_u.numpy(), dlr_targeted=l_t.detach().numpy(), dlr_targeted_grad=g_t.numpy())

    # (b), (c) attacks on the tiny ViT + head.  Seed / eps chosen so that both stages matter: at eps = 5/255 apgd-ce
    # leaves one sample robust and apgd-t then fools it; at eps = 3/255 three samples survive every attack.
    cfg = vit_ref.VIT_TINY
    w = init_weights(cfg, seed=3)
    g = torch.Generator().manual_seed(80)
    Bx = 7
    x = torch.rand(Bx, 3, cfg.image_size, cfg.image_size, generator=g)
    T = torch.randn(cfg.out_dim, 10, generator=g)
    T = T / T.norm(dim=0, keepdim=True)
    clf = vit_ref.ClassificationModelRef(cfg, w, T, 100.0).eval()
    with torch.no_grad():
        yy = clf(x).max(1)[1]
    calls = []

    def predict(v):
        calls.append(tuple(v.shape))
        return clf(v)

    out = dict(x=x.numpy(), y=yy.numpy(), T=T.numpy(), n_iter=np.int64(6), n_target_classes=np.int64(3), bs=np.int64(4),
               weights_seed=np.int64(3), weights_sha256=np.array(weights_digest(w)))
    for tag, eps in (("e3", 3 / 255), ("e5", 5 / 255)):
        calls.clear()
        atk = RefTargeted(predict, n_iter=6, norm="Linf", n_restarts=1, eps=eps, seed=0, n_target_classes=3,
                          device="cpu", alpha=2.0, use_rs=True)
        adv_t = atk.perturb(x.clone(), yy.clone())
        out[f"{tag}_adv_targeted"] = adv_t.detach().numpy()
        out[f"{tag}_n_model_calls_targeted"] = np.int64(len(calls))
        calls.clear()
        aa = RefAutoAttack(predict, norm="Linf", eps=eps, seed=0, verbose=False, version="custom",
                           attacks_to_run=["apgd-ce", "apgd-t"], device="cpu", alpha=2.0, iterations_apgd=6, use_rs=True)
        aa.apgd.n_restarts = 1
        aa.apgd_targeted.n_target_classes = 3
        x_adv, y_adv = aa.run_standard_evaluation(x.clone(), yy.clone(), bs=4, return_labels=True)
        with torch.no_grad():
            final_pred = clf(x_adv).max(1)[1]
        out[f"{tag}_eps"] = np.float64(eps)
        out[f"{tag}_x_adv"] = x_adv.detach().numpy()
        out[f"{tag}_y_adv"] = y_adv.numpy()
        out[f"{tag}_robust"] = (final_pred == yy).numpy()
        out[f"{tag}_n_model_calls_aa"] = np.int64(len(calls))
        # apgd-ce alone (what the first stage leaves robust)
        aa1 = RefAutoAttack(clf, norm="Linf", eps=eps, seed=0, verbose=False, version="custom",
                            attacks_to_run=["apgd-ce"], device="cpu", alpha=2.0, iterations_apgd=6, use_rs=True)
        aa1.apgd.n_restarts = 1
        xa1 = aa1.run_standard_evaluation(x.clone(), yy.clone(), bs=4)
        with torch.no_grad():
            out[f"{tag}_robust_after_ce"] = (clf(xa1).max(1)[1] == yy).numpy()
    save("autoattack_tiny.npz", **out)

# ------------------------------------------------------------------ G10: the L2 norm of APGDAttack / AutoAttack (--norm l2)
def g10_autopgd_l2():
    """``CLIP_eval/clip_robustbench.py --norm l2``: APGDAttack(norm='L2').perturb (CE, two restarts: gaussian random start
    scaled to the L2 sphere, the L2 step of autopgd_base.py:343-351) and AutoAttack(norm='L2', version='custom',
    ['apgd-ce', 'apgd-t']) on the tiny ViT + 10-class head of g4 / g7.  eps chosen so that some points fall and some
    survive."""
    from autoattack import AutoAttack as RefAutoAttack
    cfg = vit_ref.VIT_TINY
    w = init_weights(cfg, seed=3)
    g = torch.Generator().manual_seed(44)
    B = 6
    pool = torch.rand(96, 3, cfg.image_size, cfg.image_size, generator=g)
    T = torch.randn(cfg.out_dim, 10, generator=g)
    T = T / T.norm(dim=0, keepdim=True)
    clf = vit_ref.ClassificationModelRef(cfg, w, T, 100.0).eval()
    # Points whose clean top-2 margin is moderate.  With logits = 100 cos(e, t) most random images are classified with a
    # margin beyond ~16.6, where the fp32 cross-entropy is exactly 0 and its gradient is rounding noise IN THE REFERENCE ITSELF
    # (z_y - logsumexp rounds to 0 or to one ulp of ~50, either of which swamps the 1e-8 softmax terms): two correct fp32
    # implementations then walk different L2 trajectories, and a golden vector there pins nothing.  (The L-inf goldens
    # compare sign patterns and live with it.)
    with torch.no_grad():
        top2 = clf(pool).topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    keep = ((margin > 0.5) & (margin < 6.0)).nonzero().squeeze(1)[:B]
    assert keep.numel() == B, keep
    x = pool[keep].clone()
    with torch.no_grad():
        y = clf(x).max(1)[1]
    y[0] = (y[0] + 1) % 10            # one sample starts misclassified (never attacked)
    seen = []

    def predict(v):
        seen.append(v.detach().clone())
        return clf(v)

    eps = float(os.environ.get("G10_EPS", "0.1"))
    out = dict(x=x.numpy(), y=y.numpy(), T=T.numpy(), eps=np.float64(eps), n_iter=np.int64(12), weights_seed=np.int64(3),
               weights_sha256=np.array(weights_digest(w)), clean_margin=margin[keep].numpy())
    print("clean margins", margin[keep].numpy())
    atk = RefAPGDAttack(predict, n_iter=12, norm="L2", n_restarts=2, eps=eps, seed=0, loss="ce", device="cpu", use_rs=True)
    adv = atk.perturb(x.clone(), y.clone())
    with torch.no_grad():
        rob = (clf(adv).max(1)[1] == y).numpy()
    out.update(adv=adv.detach().numpy(), first_start=seen[1].numpy(), n_model_calls=np.int64(len(seen)), robust=rob)
    print("apgd-ce L2: robust", rob, "|adv - x|_2", (adv.detach() - x).flatten(1).norm(dim=1).numpy())
    seen.clear()
    aa = RefAutoAttack(predict, norm="L2", eps=eps, seed=0, verbose=False, version="custom",
                       attacks_to_run=["apgd-ce", "apgd-t"], device="cpu", iterations_apgd=8, use_rs=True)
    aa.apgd.n_restarts = 1
    aa.apgd_targeted.n_target_classes = 3
    yy = y.clone()
    with torch.no_grad():
        yy = clf(x).max(1)[1]         # AutoAttack over the model's own predictions: every point starts robust
    x_adv, y_adv = aa.run_standard_evaluation(x.clone(), yy.clone(), bs=4, return_labels=True)
    with torch.no_grad():
        rob_aa = (clf(x_adv).max(1)[1] == yy).numpy()
    print("AutoAttack L2: robust", rob_aa)
    out.update(aa_y=yy.numpy(), aa_x_adv=x_adv.detach().numpy(), aa_y_adv=y_adv.numpy(), aa_robust=rob_aa,
               aa_n_model_calls=np.int64(len(seen)), aa_n_iter=np.int64(8), aa_n_target_classes=np.int64(3), aa_bs=np.int64(4))
    save("autopgd_tiny_l2.npz", **out)


# ------------------------------------------------------------------ G9: Square Attack (section 8(f) rank 3, black-box route)
def g9_square():
    """SquareAttack(L-inf).perturb and AutoAttack(version='custom', ['square']).run_standard_evaluation of the
    reference on the tiny ViT + 10-class head (CLIP_eval/clip_robustbench.py:150-151 is the caller): adversarial
    images, per-sample query counts, model-call count.  Two settings: margin loss with the rescaled p schedule, and
    the AutoAttack configuration (resc_schedule=False)."""
    from autoattack.square import SquareAttack as RefSquare
    from autoattack import AutoAttack as RefAutoAttack
    cfg = vit_ref.VIT_TINY
    w = init_weights(cfg, seed=3)
    g = torch.Generator().manual_seed(80)
    Bx = 7
    x = torch.rand(Bx, 3, cfg.image_size, cfg.image_size, generator=g)
    T = torch.randn(cfg.out_dim, 10, generator=g)
    T = T / T.norm(dim=0, keepdim=True)
    clf = vit_ref.ClassificationModelRef(cfg, w, T, 100.0).eval()
    with torch.no_grad():
        yy = clf(x).max(1)[1]
    calls = []

    def predict(v):
        calls.append(tuple(v.shape))
        with torch.no_grad():
            return clf(v)

    out = dict(x=x.numpy(), y=yy.numpy(), T=T.numpy(), weights_seed=np.int64(3),
               weights_sha256=np.array(weights_digest(w)))
    for tag, eps, nq, loss, resc in (("m", 12 / 255, 60, "margin", True), ("c", 20 / 255, 40, "ce", True),
                                      ("a", 8 / 255, 50, "margin", False)):
        calls.clear()
        atk = RefSquare(predict, norm="Linf", n_queries=nq, eps=eps, p_init=.8, n_restarts=2, seed=5, verbose=False,
                        loss=loss, resc_schedule=resc, device="cpu")
        adv = atk.perturb(x.clone(), yy.clone())
        with torch.no_grad():
            pred = clf(adv).max(1)[1]
        out[f"{tag}_eps"] = np.float64(eps)
        out[f"{tag}_n_queries"] = np.int64(nq)
        out[f"{tag}_adv"] = adv.numpy()
        out[f"{tag}_robust"] = (pred == yy).numpy()
        out[f"{tag}_n_model_calls"] = np.int64(len(calls))
        # one single run on every sample (the per-query trajectory: x_best and the query counters of all of them)
        torch.random.manual_seed(11)
        nq_used, xb = atk.attack_single_run(x.clone(), yy.clone())
        out[f"{tag}_run_queries"] = nq_used.numpy()
        out[f"{tag}_run_x_best"] = xb.numpy()
        print("square", tag, "robust", (pred == yy).numpy().astype(int), "calls", len(calls), "queries", nq_used.numpy())
    # through AutoAttack (its own SquareAttack instance: p_init .8, n_restarts 1, resc_schedule False)
    calls.clear()
    aa = RefAutoAttack(predict, norm="Linf", eps=8 / 255, seed=0, verbose=False, version="custom",
                       attacks_to_run=["square"], device="cpu")
    aa.square.n_queries = 50
    x_adv, y_adv = aa.run_standard_evaluation(x.clone(), yy.clone(), bs=4, return_labels=True)
    with torch.no_grad():
        pred = clf(x_adv).max(1)[1]
    out["aa_eps"] = np.float64(8 / 255)
    out["aa_n_queries"] = np.int64(50)
    out["aa_x_adv"] = x_adv.numpy()
    out["aa_y_adv"] = y_adv.numpy()
    out["aa_robust"] = (pred == yy).numpy()
    out["aa_n_model_calls"] = np.int64(len(calls))
    print("square aa robust", (pred == yy).numpy().astype(int), "calls", len(calls))
    save("square_tiny.npz", **out)


# ------------------------------------------------------------------ G8: input transform (section 8(f) rank 4)
def g8_preprocess():
    """Resize(size, bicubic) -> CenterCrop(size) -> ToTensor on synthetic 'decoded' images, computed by Pillow (the
    resampler torchvision 0.15.2 delegates to for PIL inputs) with torchvision's size / crop arithmetic.  Stored as the
    uint8 crop (ToTensor = /255 in float32 is exact to restate)."""
    from PIL import Image
    rng = np.random.default_rng(8)
    out = {}
    cases = [(61, 47, 32), (37, 90, 32), (40, 40, 32), (20, 27, 32), (300, 260, 224), (96, 400, 64)]
    for i, (h, w, size) in enumerate(cases):
        # low-frequency content + noise + saturated patches (clipping paths of the resampler)
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 120 * np.sin(xx / 7.0 + c) * np.cos(yy / 5.0 - c) for c in range(3)], -1)
        img = np.clip(base + rng.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)
        img[: h // 5, : w // 4] = 255
        img[-h // 6:, -w // 3:] = 0
        if w <= h:
            nw, nh = size, (size if w == h else int(size * h / w))
        else:
            nh, nw = size, int(size * w / h)
        r = Image.fromarray(img).resize((nw, nh), Image.BICUBIC)
        top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
        crop = np.asarray(r.crop((left, top, left + size, top + size)))
        out[f"img{i}"] = img
        out[f"crop{i}"] = crop
        out[f"size{i}"] = np.int64(size)
    out["n"] = np.int64(len(cases))
    save("preprocess_pil.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10"]
    fns = dict(g1=g1_pgd_elementwise, g2=g2_apgd_controller, g3=g3_tiny_vit_attacks,
               g4=g4_autopgd, g5=g5_vit_vs_hf, g6=g6_losses, g7=g7_autoattack, g8=g8_preprocess, g9=g9_square,
               g10=g10_autopgd_l2)
    for k in which:
        fns[k]()
