"""oracle/linf_ref.c (the bit-exact checker used by the GPU tests) against the reference's outputs."""
import numpy as np
import pytest
import torch

from oracle import linf_c
from oracle.attacks_ref import _fwd_bwd
from tests.helpers import load_golden, SmallNet

F32 = np.float32


@pytest.mark.parametrize("mode", ["max", "min"])
def test_c_pgd_update_bit_exact(mode):
    z = load_golden(f"pgd_linf_elementwise_{mode}.npz")
    x = np.ascontiguousarray(z["x"])
    delta, vel = z["delta0"].copy(), np.zeros_like(z["delta0"])
    for i in range(4):
        linf_c.pgd_linf_update(x, np.ascontiguousarray(z["grads"][i]), delta, vel,
                               float(z["eps"]), float(z["stepsize"]), 0.9, mode)
        assert np.array_equal(x + delta, z["xadv"][i]), f"step {i}"


@pytest.mark.parametrize("n_iter", [10, 50, 100])
def test_c_apgd_bit_exact(n_iter):
    z = load_golden(f"apgd_train_smallnet_{n_iter}.npz")
    net = SmallNet(torch.from_numpy(z["w1"]), torch.from_numpy(z["w2"])).eval()
    ce = lambda lg, yy: torch.nn.functional.cross_entropy(lg, yy, reduction="none")  # noqa: E731
    call = lambda t: net(t, output_normalize=True)  # noqa: E731
    x = np.ascontiguousarray(z["x"])
    y = torch.from_numpy(z["y"])
    B = x.shape[0]
    eps = float(z["eps"])
    x_adv = np.clip(x, F32(0), F32(1))
    x_best, x_best_adv, x_adv_old = x_adv.copy(), x_adv.copy(), x_adv.copy()
    logits, loss, grad = _fwd_bwd(call, ce, x_adv, y)
    grad_best = grad.copy()
    st = linf_c.ApgdStateC(n_iter, loss, np.full(B, F32(2.0 * eps), F32),
                           (logits.max(1)[1] == y).numpy())
    iterates = [x_adv.copy()]
    for i in range(n_iter):
        linf_c.apgd_linf_step(x, x_adv, x_adv_old, grad, st.step, 0.75 if i > 0 else 1.0, eps)
        iterates.append(x_adv.copy())
        last = i == n_iter - 1
        logits, loss, g = _fwd_bwd(call, ce, x_adv, y, need_grad=not last)
        if not last:
            grad = g
        st.update(i, loss, (logits.max(1)[1] == y).numpy(), x_adv, grad, x_best, grad_best,
                  x_best_adv)
    for i, it in enumerate(iterates):
        assert np.array_equal(it, z["iterates"][i]), f"iterate {i}"
    assert np.array_equal(x_best_adv, z["x_best_adv"])
