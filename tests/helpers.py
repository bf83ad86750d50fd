"""Shared test helpers (CPU side)."""
import hashlib
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def weights_digest(w: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(w[k].numpy().tobytes())
    return h.hexdigest()


def cfg_from_array(arr, act="quick_gelu"):
    from oracle.vit_ref import VitConfig
    a = [int(v) for v in arr]
    return VitConfig(a[0], a[1], a[2], a[3], a[4], a[5], str(act))


def weights_from_golden(z):
    return {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


class SmallNet(torch.nn.Module):
    """tanh-MLP used by tests/golden/apgd_train_smallnet_*.npz (weights stored in the fixture)."""

    def __init__(self, w1, w2):
        super().__init__()
        self.w1, self.w2 = w1, w2
        self.seen = []

    def forward(self, x, output_normalize=True):
        self.seen.append(x.detach().clone())
        return torch.tanh(x.flatten(1) @ self.w1) @ self.w2


class InjectGrad(torch.autograd.Function):
    """Scalar whose gradient w.r.t. its input is exactly a prescribed tensor."""

    @staticmethod
    def forward(ctx, v, G):
        ctx.save_for_backward(G)
        return v.sum() * 0.0 + 1.0

    @staticmethod
    def backward(ctx, go):
        (G,) = ctx.saved_tensors
        return G.clone(), None
