"""Host-side mirror of the reference's model / loss wrappers around the native engine.

Mirrors (same names, argument meaning and error behaviour):
  ClipVisionModel      train/adversarial_training_clip.py:246-257
  ComputeLossWrapper   train/adversarial_training_clip.py:260-274
  compute_loss, l2, ce train/adversarial_training_clip.py:495-528
  compute_acc          train/adversarial_training_clip.py:488-492
  ClassificationModel  CLIP_eval/clip_robustbench.py:50-69

The arithmetic runs in librvlm.so: ``ClipVisionModel.forward`` is a torch.autograd.Function over
rvlm_vit_forward / rvlm_vit_backward_input (so even the reference's unmodified ``pgd`` can
differentiate through it), the losses go through rvlm_loss_grad.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .engine import VitEngine, _require_cuda, _f32c


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vision, engine: VitEngine, output_normalize: bool):
        need = ctx.needs_input_grad[0]
        emb = engine.forward(vision, None, output_normalize, save=need)
        ctx.engine = engine
        ctx.generation = engine.generation
        return emb

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.engine
        if eng.generation != ctx.generation:
            raise RuntimeError("the engine's saved activations were overwritten by a later forward; "
                               "backward must follow its forward (one graph at a time)")
        return eng.backward_input(grad_out), None, None


class ClipVisionModel(torch.nn.Module):
    """forward(vision in [0,1], output_normalize) = visual(Normalize(vision)) [L2-normalised].

    ``model`` is a :class:`VitEngine` (the native replacement of open_clip's ``model.visual``);
    ``normalize`` (a torchvision-Normalize-like object with .mean/.std) is checked against the
    constants baked into the engine, because Normalize is fused into the patch-embed load."""

    def __init__(self, model: VitEngine, args=None, normalize=None):
        super().__init__()
        if not isinstance(model, VitEngine):
            raise TypeError("ClipVisionModel expects a robustvlm_amd.VitEngine as `model`")
        self.model = model
        self.args = args
        self.normalize = normalize
        if normalize is not None and hasattr(normalize, "mean"):
            m = tuple(float(v) for v in normalize.mean)
            s = tuple(float(v) for v in normalize.std)
            if max(abs(a - b) for a, b in zip(m + s, model.mean + model.std)) > 1e-7:
                raise ValueError("normalize constants differ from the ones fused into the engine")

    def forward(self, vision, output_normalize):
        _require_cuda(vision, "vision")
        return _EncodeFn.apply(vision, self.model, bool(output_normalize))


# --------------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------------
class _LossFn(torch.autograd.Function):
    """loss value + d loss/d embedding from one rvlm_loss_grad call."""

    @staticmethod
    def forward(ctx, emb, ref, targets, kind: int, reduction: int, logit_scale: float):
        lib = L.load()
        e = _f32c(emb)
        r = _f32c(ref)
        B, D = e.shape
        Ccls = int(r.shape[1]) if kind == L.LOSS_CE else 0
        per = torch.empty(B, dtype=torch.float32, device=e.device)
        scalar = torch.empty(1, dtype=torch.float32, device=e.device)
        d_emb = torch.empty_like(e)
        scratch = torch.empty(B * Ccls + D * Ccls, dtype=torch.float32, device=e.device) if Ccls else None
        tg = targets.detach().to(torch.int64).contiguous() if kind == L.LOSS_CE else None
        with torch.cuda.device(e.device):
            L.check(lib.rvlm_loss_grad(kind, reduction, e.data_ptr(), r.data_ptr(), L.ptr(tg), None, B, D, Ccls,
                                       float(logit_scale), per.data_ptr(), scalar.data_ptr(),
                                       d_emb.data_ptr(), None, L.ptr(scratch), L.stream_ptr()),
                    "rvlm_loss_grad")
        ctx.save_for_backward(d_emb)
        ctx.reduction = reduction
        return scalar.reshape(()) if reduction == L.RED_MEAN else per

    @staticmethod
    def backward(ctx, g):
        (d_emb,) = ctx.saved_tensors
        if ctx.reduction == L.RED_MEAN:
            return d_emb * g, None, None, None, None, None
        return d_emb * g.reshape(-1, 1), None, None, None, None, None


def l2(out, targets, reduction="none"):
    """squared l2, not divided by the latent dimension (…clip.py:509-521)."""
    assert out.shape == targets.shape, f"{out.shape} != {targets.shape}"
    assert out.shape[0] > 1
    _require_cuda(out, "out")
    red = L.RED_MEAN if reduction == "mean" else L.RED_NONE
    return _LossFn.apply(out, targets, None, L.LOSS_L2, red, 1.0)


class _CeLogitsFn(torch.autograd.Function):
    """F.cross_entropy(logits, targets, reduction) and d loss / d logits from one rvlm_ce_logits call."""

    @staticmethod
    def forward(ctx, logits, targets, reduction: int):
        lib = L.load()
        lg = _f32c(logits)
        B, Ccls = lg.shape
        tg = targets.detach().to(torch.int64).contiguous()
        if tg.numel() and (int(tg.min()) < 0 or int(tg.max()) >= Ccls):
            raise IndexError(f"Target out of bounds for {Ccls} classes")           # F.cross_entropy's check
        per = torch.empty(B, dtype=torch.float32, device=lg.device)
        scalar = torch.empty(1, dtype=torch.float32, device=lg.device)
        d_logits = torch.empty_like(lg)
        with torch.cuda.device(lg.device):
            L.check(lib.rvlm_ce_logits(lg.data_ptr(), tg.data_ptr(), B, Ccls, reduction, per.data_ptr(),
                                       scalar.data_ptr(), d_logits.data_ptr(), None, L.stream_ptr()), "rvlm_ce_logits")
        ctx.save_for_backward(d_logits)
        ctx.reduction = reduction
        return scalar.reshape(()) if reduction == L.RED_MEAN else per

    @staticmethod
    def backward(ctx, g):
        (d_logits,) = ctx.saved_tensors
        if ctx.reduction == L.RED_MEAN:
            return d_logits * g, None, None
        return d_logits * g.reshape(-1, 1), None, None


def ce(out, targets, reduction="mean"):
    """cross entropy on logits (…clip.py:523-528), on the device kernel of librvlm."""
    assert out.shape[0] == targets.shape[0], (out.shape, targets.shape)
    assert out.shape[0] > 1
    _require_cuda(out, "out")
    if reduction not in ("mean", "none", "sum"):
        raise ValueError(f"reduction {reduction} not supported")
    if reduction == "sum":
        return _CeLogitsFn.apply(out, targets, L.RED_NONE).sum()
    return _CeLogitsFn.apply(out, targets, L.RED_MEAN if reduction == "mean" else L.RED_NONE)


def compute_loss(loss_str, embedding, targets, embedding_orig, logit_scale,
                 embedding_text_labels_norm=None, reduction="mean"):
    """…clip.py:495-507."""
    _require_cuda(embedding, "embedding")
    if loss_str == "l2":
        return l2(out=embedding, targets=embedding_orig, reduction=reduction)
    if loss_str == "ce":
        assert embedding.shape[0] == targets.shape[0], (embedding.shape, targets.shape)
        assert embedding.shape[0] > 1
        if reduction not in ("mean", "none"):
            raise ValueError(f"reduction {reduction} not supported")
        red = L.RED_MEAN if reduction == "mean" else L.RED_NONE
        # logits = embedding @ (logit_scale * T), F.cross_entropy - fused in rvlm_loss_grad
        return _LossFn.apply(embedding, embedding_text_labels_norm, targets, L.LOSS_CE, red,
                             float(logit_scale))
    raise ValueError(f"loss {loss_str} not supported")


class ComputeLossWrapper:
    """…clip.py:260-274."""

    def __init__(self, embedding_orig, embedding_text_labels_norm, reduction="mean", loss=None,
                 logit_scale=100.):
        self.embedding_orig = embedding_orig
        self.embedding_text_labels_norm = embedding_text_labels_norm
        self.reduction = reduction
        self.loss_str = loss
        self.logit_scale = logit_scale

    def __call__(self, embedding, targets):
        return compute_loss(loss_str=self.loss_str, embedding=embedding, targets=targets,
                            embedding_orig=self.embedding_orig, logit_scale=self.logit_scale,
                            embedding_text_labels_norm=self.embedding_text_labels_norm,
                            reduction=self.reduction)

    # what the fused device loops need
    def fused_spec(self):
        if self.loss_str == "l2":
            return "l2", self.embedding_orig
        if self.loss_str == "ce":
            return "ce", self.embedding_text_labels_norm
        raise ValueError(f"loss {self.loss_str} not supported")


@torch.no_grad()
def compute_acc(logits, targets):
    """…clip.py:488-492."""
    preds_clean = logits.max(dim=1)[1].detach()
    return (preds_clean.eq(targets).sum() / targets.shape[0]).item() * 100


class _HeadLogitsFn(torch.autograd.Function):
    """logits = (emb @ T) * scale on rvlm_head_logits / rvlm_head_logits_bwd (clip_robustbench.py:66-68)."""

    @staticmethod
    def forward(ctx, emb, T, scale: float):
        lib = L.load()
        e, t = _f32c(emb), _f32c(T)
        B, D = e.shape
        assert t.dim() == 2 and t.shape[0] == D, f"text embedding must be [{D}, C], got {tuple(t.shape)}"
        logits = torch.empty(B, t.shape[1], dtype=torch.float32, device=e.device)
        with torch.cuda.device(e.device):
            L.check(lib.rvlm_head_logits(e.data_ptr(), t.data_ptr(), B, D, t.shape[1], float(scale), logits.data_ptr(),
                                         L.stream_ptr()), "rvlm_head_logits")
        ctx.save_for_backward(t)
        ctx.scale = float(scale)
        return logits

    @staticmethod
    def backward(ctx, g):
        (t,) = ctx.saved_tensors
        lib = L.load()
        g = _f32c(g)
        B, Ccls = g.shape
        d_emb = torch.empty(B, t.shape[0], dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            L.check(lib.rvlm_head_logits_bwd(g.data_ptr(), t.data_ptr(), B, t.shape[0], Ccls, ctx.scale,
                                             d_emb.data_ptr(), L.stream_ptr()), "rvlm_head_logits_bwd")
        return d_emb, None, None


class ClassificationModel(torch.nn.Module):
    """Zero-shot head on the native encoder (CLIP_eval/clip_robustbench.py:50-69):
    logits = normalize(encode_image(Normalize(resizer(x)))) @ T  [* exp(logit_scale) = 100]."""

    def __init__(self, model, text_embedding, args=None, input_normalize=None, resizer=None,
                 logit_scale=True, logit_scale_value: float = 100.0):
        super().__init__()
        self.vision = model if isinstance(model, ClipVisionModel) else ClipVisionModel(model, args,
                                                                                        input_normalize)
        self.model = self.vision.model
        self.args = args
        self._identity_resizer = resizer is None
        self.resizer = resizer if resizer is not None else (lambda x: x)
        self.text_embedding = text_embedding
        self.logit_scale = logit_scale
        self.logit_scale_value = float(logit_scale_value)

    def forward(self, vision, output_normalize=True):
        assert output_normalize
        mb = self.model.max_batch
        if vision.shape[0] > mb and not (torch.is_grad_enabled() and vision.requires_grad):
            # evaluation batches larger than the engine's workspace (AutoAttack's bs=250 on a max_batch=128 engine)
            # are encoded in chunks; a differentiable call must fit (one saved forward per engine).  Chunks are cut
            # BEFORE the resizer so that each image passes through it exactly once
            return torch.cat([self.forward(vision[i:i + mb]) for i in range(0, vision.shape[0], mb)], 0)
        vision = self.resizer(vision)
        embedding_norm_ = self.vision(vision, True)
        return _HeadLogitsFn.apply(embedding_norm_, self.text_embedding,
                                   self.logit_scale_value if self.logit_scale else 1.0)
