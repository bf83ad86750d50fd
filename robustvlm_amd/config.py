"""Model configurations of the CLIP vision towers the reference fine-tunes
(train/adversarial_training_clip.py:95-103; open_clip model configs ViT-L-14 / ViT-B-32)."""
from __future__ import annotations

from dataclasses import dataclass

# torchvision Normalize constants quoted at train/adversarial_training_clip.py:116
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass(frozen=True)
class VitConfig:
    image_size: int = 224
    patch: int = 14
    width: int = 1024
    layers: int = 24
    heads: int = 16
    out_dim: int = 768
    act: str = "quick_gelu"   # 'quick_gelu' (OpenAI weights) or 'gelu' (LAION weights)

    @property
    def grid(self):
        return self.image_size // self.patch

    @property
    def tokens(self):
        return self.grid * self.grid + 1

    @property
    def mlp(self):
        return 4 * self.width


CONFIGS = {
    "ViT-L-14": VitConfig(224, 14, 1024, 24, 16, 768),
    "ViT-B-32": VitConfig(224, 32, 768, 12, 12, 512),
    "ViT-L-14-336": VitConfig(336, 14, 1024, 24, 16, 768),
    "ViT-B-16": VitConfig(224, 16, 768, 12, 12, 512),
}


def state_dict_shapes(cfg: VitConfig) -> dict:
    """``visual.state_dict()`` key -> shape (checkpoint format, …clip.py:239,470)."""
    W, P = cfg.width, cfg.patch
    shapes = {"class_embedding": (W,), "positional_embedding": (cfg.tokens, W),
              "proj": (W, cfg.out_dim), "conv1.weight": (W, 3, P, P),
              "ln_pre.weight": (W,), "ln_pre.bias": (W,), "ln_post.weight": (W,), "ln_post.bias": (W,)}
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        shapes.update({p + "ln_1.weight": (W,), p + "ln_1.bias": (W,),
                       p + "attn.in_proj_weight": (3 * W, W), p + "attn.in_proj_bias": (3 * W,),
                       p + "attn.out_proj.weight": (W, W), p + "attn.out_proj.bias": (W,),
                       p + "ln_2.weight": (W,), p + "ln_2.bias": (W,),
                       p + "mlp.c_fc.weight": (cfg.mlp, W), p + "mlp.c_fc.bias": (cfg.mlp,),
                       p + "mlp.c_proj.weight": (W, cfg.mlp), p + "mlp.c_proj.bias": (W,)})
    return shapes


def parameter_order(cfg: VitConfig) -> list:
    """Keys in the order of open_clip's ``visual.parameters()`` - the positional index torch.optim.AdamW's
    ``state_dict()`` uses for its 'state' entries (…clip.py:196-197,239-240): a module's own Parameters first
    (class_embedding, positional_embedding, proj), then its sub-modules in registration order (conv1, ln_pre,
    transformer, ln_post)."""
    keys = ["class_embedding", "positional_embedding", "proj", "conv1.weight", "ln_pre.weight", "ln_pre.bias"]
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        keys += [p + s for s in ("ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias",
                                 "attn.out_proj.weight", "attn.out_proj.bias", "ln_2.weight", "ln_2.bias",
                                 "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")]
    return keys + ["ln_post.weight", "ln_post.bias"]


def backward_stage_keys(cfg: VitConfig) -> list:
    """Parameter keys grouped by the stage of the parameter backward that finishes their gradient
    (rvlm_vit_backward_params_stages): [head, block L-1, ..., block 0, embeddings]."""
    blocks = []
    for i in reversed(range(cfg.layers)):
        p = f"transformer.resblocks.{i}."
        blocks.append([k for k in state_dict_shapes(cfg) if k.startswith(p)])
    return [["proj", "ln_post.weight", "ln_post.bias"]] + blocks + \
           [["ln_pre.weight", "ln_pre.bias", "positional_embedding", "class_embedding", "conv1.weight"]]


def random_state_dict(cfg: VitConfig, seed: int = 0, device="cuda"):
    """Seeded random-init weights with open_clip-like scales, generated on ``device`` (used by
    bench.py: no checkpoints are reachable offline)."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    W, L = cfg.width, cfg.layers
    attn_std, proj_std, fc_std = W ** -0.5, (W ** -0.5) * ((2 * L) ** -0.5), (2 * W) ** -0.5
    out = {}
    for k, shp in state_dict_shapes(cfg).items():
        r = torch.randn(shp, generator=g, device=device, dtype=torch.float32)
        if "ln_" in k and k.endswith("weight"):
            t = 1.0 + 0.05 * r
        elif "ln_" in k:
            t = 0.05 * r
        elif k.endswith("bias"):
            t = 0.02 * r
        elif k == "conv1.weight":
            t = r * (3 * cfg.patch * cfg.patch) ** -0.5
        elif k.endswith("out_proj.weight") or k.endswith("c_proj.weight"):
            t = r * proj_std
        elif k.endswith("c_fc.weight"):
            t = r * fc_std
        else:
            t = r * attn_std
        out[k] = t.contiguous()
    return out
