"""Weight-gradient GEMM of the training step at ViT-L/14 B=128 shapes: the copy-free contraction-major form against the
token-chunk transposes + NT form it replaced (rvlm_k_wgrad_set_transposed), same process, alternating."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
M = 128 * 257
shapes = [("fc2", 1024, 4096), ("fc1", 4096, 1024), ("out", 1024, 1024), ("qkv", 3072, 1024)]
st = torch.cuda.current_stream().cuda_stream
res = {}
for name, N, K in shapes:
    dY = torch.randn(M, N, device=dev).bfloat16()
    X = torch.randn(M, K, device=dev).bfloat16()
    dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    nb = lib.rvlm_k_wgrad_work_bytes(M, N, K)
    work = torch.empty(nb, dtype=torch.uint8, device=dev)
    out = {}
    for rep in range(3):
        for form in (1, 0):
            lib.rvlm_k_wgrad_set_transposed(form)
            for _ in range(3):
                L.check(lib.rvlm_k_wgrad_bf16(dY.data_ptr(), N, X.data_ptr(), K, M, N, K, dW.data_ptr(), K, 0, db.data_ptr(),
                                              work.data_ptr(), nb, st))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                L.check(lib.rvlm_k_wgrad_bf16(dY.data_ptr(), N, X.data_ptr(), K, M, N, K, dW.data_ptr(), K, 0, db.data_ptr(),
                                              work.data_ptr(), nb, st))
            b.record()
            torch.cuda.synchronize()
            out.setdefault("transposed" if form else "copy_free", []).append(round(a.elapsed_time(b) / 10 * 1e3, 1))
    lib.rvlm_k_wgrad_set_transposed(0)
    ref = dY.float().t() @ X.float()
    out["rel_err"] = float((dW - ref).abs().max() / ref.abs().max())
    out["tflops_copy_free"] = round(2.0 * M * N * K / (min(out["copy_free"]) * 1e-6) / 1e12, 1)
    res[name] = out
    print(name, json.dumps(out), flush=True)
