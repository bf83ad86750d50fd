"""Python handle of the native ViT forward + input-gradient engine (rvlm_vit in include/rvlm.h)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .config import VitConfig, CONFIGS, CLIP_MEAN, CLIP_STD, state_dict_shapes


def _require_cuda(t: torch.Tensor, name: str):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise L.RvlmError(f"{name} must be a CUDA (ROCm) tensor: the robustvlm_amd path runs on the "
                          f"MI355X HIP kernels only and has no CPU fallback")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


class VitEngine:
    """Owns a ``rvlm_vit`` handle: converted weights + a workspace sized for ``max_batch`` images.

    precision: 'bf16' (MFMA throughput path), 'fp32' (the reference's own precision - train/pgd_train.py:30-38 runs no
    autocast - on the fp32 matrix-pipe tiles: parity path, config C1), or 'bf16+fp32-first': a bf16 handle plus an fp32
    handle of the same weights; ``pgd_run`` evaluates its FIRST iteration on the fp32 handle (rvlm_pgd_run_mixed) and
    gradient-free forwards (``save=False``: the clean embedding FARE's loss is measured against) run there too, everything
    else in bf16.  Why: FARE's first cotangent 2 (phi(x + d0) - phi(x)) is a difference of nearly equal embeddings and
    one part bf16 rounding noise in three; with one fp32 iteration the attack takes the
    reference's first step (gradient-sign agreement 0.82 -> 0.9999) at the price of one fp32 iteration per call.
    'x3' (round 6): fp32 storage, LayerNorm, softmax and attention products like 'fp32', the encoder's LINEARS as split-bf16
    products (a_hi w_hi + a_hi w_lo + a_lo w_hi, fp32 accumulate: ~16 mantissa bits) on the bf16 matrix pipe -
    embeddings within ~1e-5 of the fp32 mode's at a third of its time; 'bf16+x3-first' = the mixed mode with that handle
    for the first iteration and the clean embedding (oracle/split_bf16_emulation.py: first-step sign agreement 0.9996).
    'bf16+x3fwd-first' (round 6): the same, but only the FORWARDS of the first iteration (and the clean embedding) run on the x3
    handle; the first input gradient is evaluated by the bf16 backward kernels from the x3 forward's saved tensors
    (rvlm_pgd_run_mixed_fwd / rvlm_vit_backward_input_from) - the noise of FARE's first step sits in the forward difference
    phi(x + d0) - phi(x), not in the cotangent's way back (emulation: 0.998 sign agreement), and the x3 backward was more than
    half of the mixed mode's extra time.  'bf16+x3fwd': that split for EVERY iteration of ``pgd_run`` (split-bf16 forwards, bf16
    backwards) - a rung between 'bf16+x3fwd-first' and 'x3'; ``faithful_iterations=k`` runs the first k iterations that way
    (identical pixels with the reference's pgd() on the seeded ViT-L/14: k = 1: 0.927, 10: 0.984; tests/test_gpu_fullsize.py)."""

    def __init__(self, cfg, state_dict: dict, precision: str = "bf16", max_batch: int = 128,
                 mean=CLIP_MEAN, std=CLIP_STD, device=None, trainable: bool = False,
                 inference_only: bool = False, faithful_iterations: int | None = None):
        if isinstance(cfg, str):
            cfg = CONFIGS[cfg]
        if not torch.cuda.is_available():
            raise L.RvlmError("no ROCm device visible: robustvlm_amd needs an MI355X (no CPU fallback)")
        self.lib = L.load()
        self.cfg = cfg
        self.precision = precision
        self.max_batch = int(max_batch)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.index is None:
            self.device = torch.device(f"cuda:{torch.cuda.current_device()}")
        self.mean, self.std = tuple(float(v) for v in mean), tuple(float(v) for v in std)
        self.generation = 0
        self._apgd_rho = 0.75
        self._h = C.c_void_p()
        c = L.VitConfigC()
        c.image_size, c.patch, c.width, c.layers = cfg.image_size, cfg.patch, cfg.width, cfg.layers
        c.heads, c.out_dim = cfg.heads, cfg.out_dim
        c.act = L.ACT_QUICK_GELU if cfg.act == "quick_gelu" else L.ACT_GELU
        if precision not in ("bf16", "fp32", "x3", "bf16+fp32-first", "bf16+x3-first", "bf16+x3fwd-first", "bf16+x3fwd"):
            raise ValueError(f"precision {precision!r} not supported")
        self.mixed = precision in ("bf16+fp32-first", "bf16+x3-first", "bf16+x3fwd-first", "bf16+x3fwd")
        self.handoff = precision in ("bf16+x3fwd-first", "bf16+x3fwd")
        self.n_first = 4096 if precision == "bf16+x3fwd" else 1      # 'bf16+x3fwd': EVERY iteration's forward on the x3 handle
        if faithful_iterations is not None:      # the continuum between the '-first' modes and 'bf16+x3fwd': the first k iterations
            if not self.mixed or int(faithful_iterations) < 0:
                raise ValueError("faithful_iterations needs a mixed precision and a count >= 0")
            self.n_first = int(faithful_iterations)
        if self.mixed and (trainable or inference_only):
            raise ValueError(f"precision {precision!r} is an attack-engine option (not trainable / inference_only)")
        if precision == "x3" and trainable:
            raise ValueError("precision 'x3' is an attack / inference precision (no weight gradients)")
        c.precision = L.PREC_F32 if precision == "fp32" else L.PREC_F32X3 if precision == "x3" else L.PREC_BF16
        self._h32 = C.c_void_p()
        c.max_batch = self.max_batch
        c.trainable = 1 if trainable else (-1 if inference_only else 0)
        self.trainable = bool(trainable)
        c.mean = (C.c_float * 3)(*self.mean)
        c.std = (C.c_float * 3)(*self.std)
        with torch.cuda.device(self.device):
            w, keep = self._weights_struct(state_dict)
            L.check(self.lib.rvlm_vit_create(C.byref(c), C.byref(w), L.stream_ptr(), C.byref(self._h)),
                    "rvlm_vit_create")
            if self.mixed:
                c.precision = L.PREC_F32 if precision == "bf16+fp32-first" else L.PREC_F32X3
                if self.handoff:
                    c.trainable = -2     # forward provider of the bf16 handle: no backward scratch, one slot for qkv / o / fc1 / P
                L.check(self.lib.rvlm_vit_create(C.byref(c), C.byref(w), L.stream_ptr(), C.byref(self._h32)),
                        "rvlm_vit_create (fp32 handle)")
                if self.handoff:     # its saving forwards are flash (run for the bf16 handle): the clean embedding too, bit-consistently
                    L.check(self.lib.rvlm_vit_set_flash_inference(self._h32, 1), "rvlm_vit_set_flash_inference")
        del keep

    # ---- weights ------------------------------------------------------------------------------
    def _weights_struct(self, sd: dict, inplace: bool = False):
        """ctypes view of a state-dict-shaped collection of fp32 device tensors.  inplace=True requires the
        tensors themselves (contiguous fp32 on this device) - used for gradient OUTPUT buffers."""
        cfg = self.cfg
        shapes = state_dict_shapes(cfg)
        keep = {}
        for k, shp in shapes.items():
            if k not in sd:
                raise KeyError(f"state_dict is missing {k!r}")
            if inplace:
                t = sd[k]
                if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                    raise ValueError(f"{k}: gradient buffers must be contiguous fp32 CUDA tensors")
            else:
                t = _f32c(sd[k]).to(self.device)
            if tuple(t.shape) != tuple(shp):
                raise ValueError(f"{k}: shape {tuple(t.shape)} != {tuple(shp)}")
            keep[k] = t
        w = L.VitWeightsC()
        for f in L.TOP_FIELDS:
            setattr(w, f, keep[L.TOP_KEYS[f]].data_ptr())
        blocks = (L.BlockWeightsC * cfg.layers)()
        for i in range(cfg.layers):
            for f in L.BLOCK_FIELDS:
                setattr(blocks[i], f, keep[f"transformer.resblocks.{i}.{L.BLOCK_KEYS[f]}"].data_ptr())
        w.blocks_host = C.cast(blocks, C.POINTER(L.BlockWeightsC))
        keep["__blocks__"] = blocks
        return w, keep

    def load_state_dict(self, state_dict: dict):
        """Re-import weights (e.g. after an optimizer step of the outer trainer)."""
        with torch.cuda.device(self.device):
            w, keep = self._weights_struct(state_dict)
            L.check(self.lib.rvlm_vit_load_weights(self._h, C.byref(w), L.stream_ptr()))
            if self.mixed:
                L.check(self.lib.rvlm_vit_load_weights(self._h32, C.byref(w), L.stream_ptr()))
            torch.cuda.current_stream().synchronize()
        del keep

    # ---- forward / backward --------------------------------------------------------------------
    def _check_images(self, x):
        _require_cuda(x, "vision")
        c = self.cfg
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, c.image_size, c.image_size):
            raise ValueError(f"expected [B,3,{c.image_size},{c.image_size}] images, got {tuple(x.shape)}")
        if x.shape[0] > self.max_batch:
            raise ValueError(f"batch {x.shape[0]} exceeds the engine's max_batch {self.max_batch}")
        if x.device != self.device:
            raise ValueError(f"images live on {x.device}, the engine on {self.device}")

    def forward(self, x, delta=None, output_normalize=False, save=False) -> torch.Tensor:
        self._check_images(x)
        x = _f32c(x)
        d = _f32c(delta) if delta is not None else None
        B = x.shape[0]
        out = torch.empty(B, self.cfg.out_dim, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            h = self._h32 if (self.mixed and not save) else self._h      # (mixed: gradient-free forwards in fp32)
            L.check(self.lib.rvlm_vit_forward(h, x.data_ptr(), L.ptr(d), B, int(bool(output_normalize)),
                                              int(save), out.data_ptr(), L.stream_ptr()),
                    "rvlm_vit_forward")
        # every forward (saving or not) overwrites state a pending backward would read: rvlm_vit_forward invalidates the
        # saved pass on the C side, the generation counter lets _EncodeFn.backward say so in Python terms
        self.generation += 1
        return out

    @property
    def n_stages(self) -> int:
        """Stages of the parameter backward: head, one per transformer block (last block first), embeddings."""
        return self.cfg.layers + 2

    def backward_params(self, d_emb, grads: dict, accumulate: bool = False, stages=None):
        """Weight gradients of the last ``forward(..., save=2)`` into the fp32 tensors of ``grads``
        (state_dict keys / shapes); rvlm_vit_backward_params.  ``stages=(begin, end)`` runs that slice of the pass
        only (rvlm_vit_backward_params_stages; slices of one backward must be run in order, begin 0 first)."""
        _require_cuda(d_emb, "d_emb")
        d = _f32c(d_emb)
        with torch.cuda.device(d.device):
            w, keep = self._weights_struct(grads, inplace=True)
            if stages is None:
                L.check(self.lib.rvlm_vit_backward_params(self._h, d.data_ptr(), d.shape[0], C.byref(w),
                                                          int(bool(accumulate)), L.stream_ptr()),
                        "rvlm_vit_backward_params")
            else:
                L.check(self.lib.rvlm_vit_backward_params_stages(self._h, d.data_ptr(), d.shape[0], C.byref(w),
                                                                 int(bool(accumulate)), int(stages[0]), int(stages[1]),
                                                                 L.stream_ptr()), "rvlm_vit_backward_params_stages")
        del keep

    def backward_input(self, d_emb) -> torch.Tensor:
        _require_cuda(d_emb, "d_emb")
        d = _f32c(d_emb)
        B = d.shape[0]
        c = self.cfg
        g = torch.empty(B, 3, c.image_size, c.image_size, device=d.device, dtype=torch.float32)
        with torch.cuda.device(d.device):
            L.check(self.lib.rvlm_vit_backward_input(self._h, d.data_ptr(), B, g.data_ptr(), L.stream_ptr()),
                    "rvlm_vit_backward_input")
        return g

    def handoff_inputgrad(self, x, delta, ref=None, output_normalize=False, cot=None, fused=True):
        """The first FARE iteration of the mixed modes' handoff, as two calls: forward(x + delta) on the fp32-storage handle
        (activations kept), then d mean_b |emb - ref|^2 / d(x + delta) - or d <cot, emb> / d(x + delta) for a given cotangent -
        on the bf16 handle's backward kernels (rvlm_vit_backward_input_from).  Returns (emb, grad_x)."""
        if not self.mixed:
            raise ValueError("handoff_inputgrad needs a mixed-precision engine (two handles)")
        self._check_images(x)
        x = _f32c(x)
        d = _f32c(delta) if delta is not None else None
        B = x.shape[0]
        emb = torch.empty(B, self.cfg.out_dim, device=x.device, dtype=torch.float32)
        g = torch.empty_like(x)
        with torch.cuda.device(x.device):
            if fused:    # the forward writes the bf16 handle's tensors itself (fp32 flash attention, no probabilities kept)
                L.check(self.lib.rvlm_vit_forward_for(self._h32, self._h, x.data_ptr(), L.ptr(d), B, int(bool(output_normalize)),
                                                      emb.data_ptr(), L.stream_ptr()), "rvlm_vit_forward_for")
            else:        # an ordinary saving forward; the bf16 tensors are exported at the handoff (refused by a provider handle:
                         # the mixed engines of this class create theirs as one - test hook for handles created otherwise)
                L.check(self.lib.rvlm_vit_forward(self._h32, x.data_ptr(), L.ptr(d), B, int(bool(output_normalize)), 1,
                                                  emb.data_ptr(), L.stream_ptr()), "rvlm_vit_forward")
            d_emb = _f32c(cot) if cot is not None else ((emb - _f32c(ref)) * (2.0 / B)).contiguous()
            L.check(self.lib.rvlm_vit_backward_input_from(self._h, self._h32, d_emb.data_ptr(), B, g.data_ptr(), L.stream_ptr()),
                    "rvlm_vit_backward_input_from")
        self.generation += 1
        return emb, g

    def fwd_inputgrad(self, x, delta, loss_kind, reduction, ref, targets, output_normalize, logit_scale=100.0):
        """One iteration's model work in one native call (rvlm_vit_fwd_inputgrad): returns
        (emb, loss_per_sample, loss_scalar, grad_x)."""
        self._check_images(x)
        x = _f32c(x)
        d = _f32c(delta) if delta is not None else None
        ref = _f32c(ref)
        tg = targets.detach().to(torch.int64).contiguous() if isinstance(targets, torch.Tensor) else None
        B = x.shape[0]
        ls = self._loss_spec(loss_kind, reduction, output_normalize, ref, tg, logit_scale, B=B)
        emb = torch.empty(B, self.cfg.out_dim, device=x.device, dtype=torch.float32)
        per = torch.empty(B, device=x.device, dtype=torch.float32)
        scalar = torch.empty(1, device=x.device, dtype=torch.float32)
        g = torch.empty_like(x)
        with torch.cuda.device(x.device):
            L.check(self.lib.rvlm_vit_fwd_inputgrad(self._h, x.data_ptr(), L.ptr(d), B, C.byref(ls), emb.data_ptr(),
                                                    per.data_ptr(), scalar.data_ptr(), g.data_ptr(), L.stream_ptr()),
                    "rvlm_vit_fwd_inputgrad")
        self.generation += 1
        return emb, per, scalar.reshape(()), g

    # ---- fused loops -----------------------------------------------------------------------------
    def _loss_spec(self, loss_kind, reduction, output_normalize, ref, targets, logit_scale, y_target=None, B=None):
        """The C loops take raw device pointers: everything the reference would catch with an assert / a shape error /
        F.cross_entropy's label check is caught HERE (l2: `out.shape == targets.shape`, …clip.py:512; head losses:
        T is [out_dim, C] and labels lie in [0, C))."""
        D = self.cfg.out_dim
        for name, t in (("loss reference", ref), ("targets", targets), ("y_target", y_target)):
            if t is None:
                continue
            _require_cuda(t, name)
            if t.device != self.device:
                raise ValueError(f"{name} lives on {t.device}, the engine on {self.device}")
        if loss_kind == "l2":
            assert B is None or tuple(ref.shape) == (B, D), f"{(B, D)} != {tuple(ref.shape)}"
        else:
            assert ref.dim() == 2 and ref.shape[0] == D, f"text embedding must be [{D}, C], got {tuple(ref.shape)}"
            if not 0 < ref.shape[1] <= 1024:
                raise ValueError(f"head losses support up to 1024 classes, got {ref.shape[1]}")
            if targets is None:
                raise ValueError(f"loss {loss_kind!r} needs integer targets")
        for name, t in (("targets", targets), ("y_target", y_target)):
            if t is None:
                continue
            assert B is None or t.shape[0] == B, f"{name}: {tuple(t.shape)} for a batch of {B}"
            if loss_kind != "l2" and t.numel():
                lo, hi = int(t.min()), int(t.max())
                if lo < 0 or hi >= ref.shape[1]:
                    raise IndexError(f"{name} out of range: [{lo}, {hi}] for {ref.shape[1]} classes")
        ls = L.LossSpecC()
        ls.loss_kind = {"l2": L.LOSS_L2, "ce": L.LOSS_CE, "dlr": L.LOSS_DLR, "dlr-targeted": L.LOSS_DLR_TARGETED}[loss_kind]
        ls.reduction = {"mean": L.RED_MEAN, "none": L.RED_NONE}[reduction]
        ls.output_normalize = int(bool(output_normalize))
        ls.logit_scale = float(logit_scale)
        ls.ref = ref.data_ptr()
        ls.n_classes = int(ref.shape[1]) if loss_kind != "l2" else 0
        ls.targets = targets.data_ptr() if targets is not None else None
        ls.y_target = y_target.data_ptr() if y_target is not None else None
        return ls

    def pgd_run(self, x, delta0, loss_kind, reduction, ref, targets, output_normalize, eps, iterations,
                stepsize, momentum, mode, logit_scale=100.0, want_trace=False, norm_kind=0):
        """Whole pgd() loop on the device (rvlm_pgd_run_norm; norm_kind 0 = L-inf, 2 = L2).
        Returns (x_adv, flags:int, loss_trace|None)."""
        self._check_images(x)
        x = _f32c(x)
        d0 = _f32c(delta0) if delta0 is not None else None
        ref = _f32c(ref)
        tg = targets.detach().to(torch.int64).contiguous() if isinstance(targets, torch.Tensor) else None
        ls = self._loss_spec(loss_kind, reduction, output_normalize, ref, tg, logit_scale, B=x.shape[0])
        out = torch.empty_like(x)
        flags = torch.zeros(1, dtype=torch.int32, device=x.device)
        trace = torch.zeros(max(iterations, 1), dtype=torch.float32, device=x.device) if want_trace else None
        with torch.cuda.device(x.device):
            if self.mixed:
                fn = self.lib.rvlm_pgd_run_mixed_fwd if self.handoff else self.lib.rvlm_pgd_run_mixed
                L.check(fn(self._h, self._h32, self.n_first, x.data_ptr(), L.ptr(d0), x.shape[0],
                           C.byref(ls), int(norm_kind), float(eps), int(iterations),
                           float(stepsize), float(momentum), 1 if mode == "max" else 0,
                           out.data_ptr(), L.ptr(trace), flags.data_ptr(), L.stream_ptr()),
                        "rvlm_pgd_run_mixed_fwd" if self.handoff else "rvlm_pgd_run_mixed")
            else:
                L.check(self.lib.rvlm_pgd_run_norm(self._h, x.data_ptr(), L.ptr(d0), x.shape[0], C.byref(ls),
                                                   int(norm_kind), float(eps), int(iterations), float(stepsize),
                                                   float(momentum), 1 if mode == "max" else 0, out.data_ptr(),
                                                   L.ptr(trace), flags.data_ptr(), L.stream_ptr()), "rvlm_pgd_run")
        self.generation += 1
        return out, flags, trace

    def apgd_run(self, x, x_init, loss_kind, ref, targets, output_normalize, eps, n_iter, step0,
                 train_variant, logits_from_head, logit_scale=100.0, want_extra=False, y_target=None, norm_kind=0,
                 rho=0.75):
        """Whole APGD Linf loop on the device (rvlm_apgd_run).  loss_kind: 'l2' | 'ce' | 'dlr' | 'dlr-targeted'
        (the DLR losses of AutoAttack need logits_from_head; 'dlr-targeted' needs y_target [B] int64).  ``rho``:
        APGDAttack's oscillation threshold (rvlm_vit_set_apgd_rho)."""
        self._check_images(x)
        x = _f32c(x)
        xi = _f32c(x_init) if x_init is not None else None
        ref = _f32c(ref)
        tg = targets.detach().to(torch.int64).contiguous()
        yt = y_target.detach().to(torch.int64).contiguous() if y_target is not None else None
        if (loss_kind == "dlr-targeted") != (yt is not None):
            raise ValueError("y_target goes with loss_kind='dlr-targeted'")
        B = x.shape[0]
        ls = self._loss_spec(loss_kind, "none", output_normalize, ref, tg, logit_scale, yt, B=B)
        x_best_adv = torch.empty_like(x)
        x_best = torch.empty_like(x) if want_extra else None
        loss_best = torch.empty(B, dtype=torch.float32, device=x.device) if want_extra else None
        acc = torch.empty(B, dtype=torch.uint8, device=x.device) if want_extra else None
        if float(rho) != self._apgd_rho:
            L.check(self.lib.rvlm_vit_set_apgd_rho(self._h, float(rho)), "rvlm_vit_set_apgd_rho")
            self._apgd_rho = float(rho)
        with torch.cuda.device(x.device):
            L.check(self.lib.rvlm_apgd_run_norm(self._h, x.data_ptr(), L.ptr(xi), B, C.byref(ls), int(norm_kind),
                                                float(eps), int(n_iter), float(step0), int(bool(train_variant)),
                                                int(bool(logits_from_head)), x_best_adv.data_ptr(), L.ptr(x_best),
                                                L.ptr(loss_best), L.ptr(acc), L.stream_ptr()), "rvlm_apgd_run")
        self.generation += 1
        return x_best_adv, x_best, loss_best, acc

    # ---- measurement -----------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        L.check(self.lib.rvlm_vit_set_profiling(self._h, int(bool(on))))

    def reset_profile(self):
        L.check(self.lib.rvlm_vit_reset_profile(self._h))

    def get_profile(self) -> dict:
        n = C.c_int(64)
        arr = (L.ProfileEntryC * 64)()
        L.check(self.lib.rvlm_vit_get_profile(self._h, arr, C.byref(n)))
        return {arr[i].name.decode(): dict(ms=arr[i].total_ms, flops=arr[i].flops, bytes=arr[i].bytes,
                                           launches=arr[i].launches) for i in range(n.value)}

    def workspace_bytes(self) -> int:
        return int(self.lib.rvlm_vit_workspace_bytes(self._h)) + (
            int(self.lib.rvlm_vit_workspace_bytes(self._h32)) if self._h32.value else 0)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.rvlm_vit_destroy(self._h)
            self._h = C.c_void_p()
        if getattr(self, "_h32", None) is not None and self._h32.value:
            self.lib.rvlm_vit_destroy(self._h32)
            self._h32 = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
