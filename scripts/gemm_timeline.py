#!/usr/bin/env python3
"""Issue timeline of two consecutive K-steps (steps 8 and 9 of the first tile) of every wave of workgroup 0 of the
persistent GEMM (rvlm_k_gemm_set_ablate(2048)): cycles relative to the earliest barrier release of the step."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
arm = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
g = torch.Generator(device=dev).manual_seed(0)
_wa = torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(300):
    torch.matmul(_wa, _wa)
torch.cuda.synchronize()
lib.rvlm_k_gemm_set_variant(2)
names = ["barrier", "B req", "slice3", "slice0", "slice1", "slice2", "A req", "landed"]
for name, m, n, k in [("cube4k", 4096, 4096, 4096), ("fc1_dgrad", 128 * 257, 1024, 4096)]:
    A = torch.randn((m + 255) // 256 * 256, k, generator=g, device=dev).bfloat16()
    Bw = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, generator=g, device=dev)
    out = torch.zeros(m, n, dtype=torch.bfloat16, device=dev)
    trace = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dev)
    lib.rvlm_k_gemm_set_ablate(arm)

    def run():
        L.check(lib.rvlm_k_gemm_bf16_nt(A.data_ptr(), k, Bw.data_ptr(), k, m, n, k, A.shape[0], 0, bias.data_ptr(), out.data_ptr(), n,
                                        None, None, None, 0, L.stream_ptr()))
    for _ in range(10):
        run()
    lib.rvlm_k_gemm_set_trace(trace.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.rvlm_k_gemm_set_trace(None)
    lib.rvlm_k_gemm_set_ablate(0)
    t = trace[:128].view(8, 2, 8).cpu()
    print(f"{name} (arm {arm}): stamps in cycles after the first wave passed the barrier of step 8; order within a step: "
          "barrier, B req (+ first fragments), slice 3 of the previous stage issued, slices 0 / 1 / 2 issued, A req, operands landed")
    base = int(t[:, 0, 0].min())
    for w in range(8):
        row = []
        for st in range(2):
            v = t[w, st]
            # stamp 2 (slice 3) belongs to the step that STARTED at this step's barrier but is written a step late: shift
            order = [0, 1, 3, 4, 5, 6, 7]
            row.append(" ".join(f"{names[i]}={int(v[i]) - base:5d}" for i in order))
        s3 = [int(t[w, st, 2]) - base for st in range(2)]
        print(f"  wave {w}: step8 [{row[0]}]  step9 [{row[1]}]  slice3 stamps {s3}")
