"""robustvlm_amd - MI355X-native adversarial inner loop for RobustVLM-style CLIP fine-tuning.

Host-side mirror of the reference's interface for the hot path; every
arithmetic op runs in the in-tree HIP library ``librvlm.so`` (include/rvlm.h):

    reference                                           here
    train/pgd_train.py::pgd                             robustvlm_amd.pgd
    train/apgd_train.py::apgd_train                     robustvlm_amd.apgd_train
    autoattack/autopgd_base.py::APGDAttack              robustvlm_amd.APGDAttack
    vlm_eval/attacks/utils.py                           robustvlm_amd.attack_utils
    train/adversarial_training_clip.py::ClipVisionModel robustvlm_amd.ClipVisionModel
      ::ComputeLossWrapper/compute_loss/l2/ce           robustvlm_amd.{ComputeLossWrapper,...}
    CLIP_eval/clip_robustbench.py::ClassificationModel  robustvlm_amd.ClassificationModel
    open_clip VisionTransformer (third party)           robustvlm_amd.VitEngine
"""
from .config import VitConfig, CONFIGS, CLIP_MEAN, CLIP_STD, random_state_dict, state_dict_shapes
from .engine import VitEngine
from .clip_model import (ClipVisionModel, ComputeLossWrapper, ClassificationModel, compute_loss, l2, ce,
                         compute_acc)
from .attack_utils import project_perturbation, normalize_grad
from .pgd_train import pgd
from .apgd_train import apgd_train
from .autopgd import APGDAttack, APGDAttack_targeted
from .autoattack import AutoAttack, EvaluationState
from .square import SquareAttack
from .eval_utils import compute_accuracy_no_dataloader, zeroshot_head
from .preprocess import ResizeCenterCropToTensor
from .checkpoint import load_visual_state_dict, CheckpointWriter, resume_paths

__all__ = ["VitConfig", "CONFIGS", "CLIP_MEAN", "CLIP_STD", "random_state_dict", "state_dict_shapes",
           "VitEngine", "ClipVisionModel", "ComputeLossWrapper", "ClassificationModel", "compute_loss",
           "l2", "ce", "compute_acc", "project_perturbation", "normalize_grad", "pgd", "apgd_train",
           "APGDAttack", "APGDAttack_targeted", "AutoAttack", "EvaluationState", "SquareAttack", "compute_accuracy_no_dataloader", "zeroshot_head", "ResizeCenterCropToTensor", "load_visual_state_dict", "CheckpointWriter", "resume_paths"]
