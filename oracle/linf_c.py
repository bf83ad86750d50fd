"""ctypes binding of oracle/linf_ref.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblinf_ref.so")
_lib = None

F32P = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
U8P = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.ref_pgd_linf_update.argtypes = [F32P, F32P, F32P, F32P, C.c_size_t, C.c_float, C.c_float,
                                          C.c_float, C.c_int]
        L.ref_apgd_linf_step.argtypes = [F32P, F32P, F32P, F32P, F32P, C.c_float, C.c_float,
                                         C.c_size_t, C.c_int]
        L.ref_apgd_controller.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, F32P, U8P,
                                          F32P, F32P, F32P, F32P, F32P, U8P, U8P, U8P, U8P]
        L.ref_apgd_select.argtypes = [F32P, F32P, F32P, F32P, F32P, U8P, U8P, U8P, C.c_size_t,
                                      C.c_int]
        for f in (L.ref_pgd_linf_update, L.ref_apgd_linf_step, L.ref_apgd_controller,
                  L.ref_apgd_select):
            f.restype = None
        _lib = L
    return _lib


def pgd_linf_update(x, g, delta, vel, eps, step, mom=0.9, mode="max"):
    """In place on delta/vel (contiguous fp32 numpy arrays)."""
    lib().ref_pgd_linf_update(x, g, delta, vel, x.size, eps, step, mom, 1 if mode == "max" else 0)


def apgd_linf_step(x, x_adv, x_adv_old, grad, step, a, eps):
    B = x.shape[0]
    lib().ref_apgd_linf_step(x, x_adv, x_adv_old, grad, step, a, eps, x.size // B, B)


class ApgdStateC:
    """Drives ref_apgd_controller / ref_apgd_select with the data-independent (k, counter3)
    schedule of train/apgd_train.py:153-156,329-355 kept on the host."""

    def __init__(self, n_iter, loss0, step0, acc0):
        from oracle.attacks_ref import apgd_schedule
        self.n_iter = n_iter
        self.k, self.n_iter_min, self.size_decr = apgd_schedule(n_iter)
        self.counter3 = 0
        B = loss0.shape[0]
        self.B = B
        self.loss_steps = np.zeros((n_iter, B), np.float32)
        self.loss_best = loss0.astype(np.float32).copy()
        self.loss_best_last_check = self.loss_best.copy()
        self.reduced_last_check = np.ones(B, np.float32)
        self.step = step0.astype(np.float32).copy()
        self.acc = acc0.astype(np.uint8).copy()
        self.f_notpred = np.zeros(B, np.uint8)
        self.f_improved = np.zeros(B, np.uint8)
        self.f_reduced = np.zeros(B, np.uint8)

    def update(self, i, loss_i, pred, x_adv, grad, x_best, grad_best, x_best_adv):
        self.counter3 += 1
        do_check = int(self.counter3 == self.k)
        lib().ref_apgd_controller(i, self.B, self.n_iter, self.k, do_check,
                                  np.ascontiguousarray(loss_i, np.float32),
                                  np.ascontiguousarray(pred, np.uint8), self.loss_steps,
                                  self.loss_best, self.loss_best_last_check,
                                  self.reduced_last_check, self.step, self.acc, self.f_notpred,
                                  self.f_improved, self.f_reduced)
        lib().ref_apgd_select(x_adv, grad, x_best, grad_best, x_best_adv, self.f_notpred,
                              self.f_improved, self.f_reduced, x_adv.size // self.B, self.B)
        if do_check:
            self.counter3 = 0
            self.k = max(self.k - self.size_decr, self.n_iter_min)
