"""On-disk checkpoint format of the fine-tuned vision tower.

Model files are exactly what the reference writes / reads, so FARE / TeCoA / OpenAI weights drop in and
the outputs feed ``llava/model/multimodal_encoder/clip_encoder.py:51-59`` unchanged:

* ``torch.save(visual.state_dict())`` with open_clip key names (train/adversarial_training_clip.py:239,470);
* TeCoA files wrap it as ``{'vision_encoder_state_dict': ...}`` (CLIP_eval/eval_utils.py:45-48);
* file naming ``step_{N}.pt`` (every steps//10), rolling ``fallback_{N}.pt`` (every 200 steps, older ones
  removed), ``final.pt``, optimizer state next to each as ``*_opt.pt`` (…clip.py:239-240,467-479), a
  ``*_temp`` output dir is renamed when training finishes (:242-244), and resuming asserts that the start step
  appears in the optimizer-state file name (:98-102).

The optimizer file is ``torch.optim.AdamW(visual.parameters()).state_dict()`` as the reference writes it ({'state':
{index: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]}, …clip.py:240,471): the positional index follows
open_clip's ``visual.parameters()`` order (robustvlm_amd.config.parameter_order), so a reference run resumes from a
file written here and ``AdversarialTrainer.load_optimizer_state_dict`` reads the reference's files (and the name-keyed
files round 1 of this package wrote).
"""
from __future__ import annotations

import os

import torch

from .config import VitConfig, state_dict_shapes


def load_visual_state_dict(checkpoint, cfg: VitConfig | None = None, map_location="cpu") -> dict:
    """Accepts a path or an already-loaded object; unwraps the TeCoA container; validates against cfg."""
    if isinstance(checkpoint, (str, os.PathLike)):
        checkpoint = torch.load(checkpoint, map_location=map_location)
    if "vision_encoder_state_dict" in checkpoint.keys():       # tecoa checkpoint (eval_utils.py:45-46)
        checkpoint = checkpoint["vision_encoder_state_dict"]
    if cfg is not None:
        shapes = state_dict_shapes(cfg)
        missing = [k for k in shapes if k not in checkpoint]
        if missing:
            raise KeyError(f"checkpoint is missing {len(missing)} keys, e.g. {missing[:3]}")
        for k, shp in shapes.items():
            if tuple(checkpoint[k].shape) != tuple(shp):
                raise ValueError(f"{k}: checkpoint shape {tuple(checkpoint[k].shape)} != {tuple(shp)}")
    return checkpoint


def resume_paths(optimizer_state: str, start_step: int):
    """…clip.py:98-102: the model file of an optimizer-state file, with the reference's consistency asserts."""
    assert start_step > 0
    assert str(start_step) in optimizer_state
    return optimizer_state.replace("_opt", ""), optimizer_state


class CheckpointWriter:
    """File layout / cadence of …clip.py:467-479 and :239-244."""

    def __init__(self, output_dir: str, steps: int, save_checkpoints: bool = True):
        self.output_dir, self.steps, self.save_checkpoints = output_dir, steps, save_checkpoints
        self.ckpt_dir = os.path.join(output_dir, "checkpoints")
        os.makedirs(self.ckpt_dir, exist_ok=True)

    def _save(self, stem, model_sd, opt_sd):
        torch.save(model_sd, os.path.join(self.ckpt_dir, f"{stem}.pt"))
        torch.save(opt_sd, os.path.join(self.ckpt_dir, f"{stem}_opt.pt"))

    def after_step(self, step_total: int, model_sd_fn, opt_sd_fn):
        """Call once per optimizer step with callables producing the state dicts (evaluated only on save)."""
        tenth = max(self.steps // 10, 1)
        if self.save_checkpoints and step_total % tenth == 0:
            self._save(f"step_{step_total}", model_sd_fn(), opt_sd_fn())
        if step_total % 200 == 0:
            self._save(f"fallback_{step_total}", model_sd_fn(), opt_sd_fn())
            for f in os.listdir(self.ckpt_dir):
                if f.startswith("fallback") and str(step_total) not in f:
                    os.remove(os.path.join(self.ckpt_dir, f))

    def final(self, model_sd, opt_sd) -> str:
        self._save("final", model_sd, opt_sd)
        if self.output_dir.endswith("_temp"):
            os.rename(self.output_dir, self.output_dir[:-5])
            self.output_dir = self.output_dir[:-5]
            self.ckpt_dir = os.path.join(self.output_dir, "checkpoints")
        return self.output_dir
