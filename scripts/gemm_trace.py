#!/usr/bin/env python3
"""Per-tile timeline of the persistent 256x256 GEMM (s_memtime stamps of wave 0 of every workgroup).
Prints, per shape, the mean over workgroups of: first K-step (incl. the seam wait), whole mainloop, epilogue issue."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
M = 128 * 257
shapes = [("qkv", M, 3072, 1024, 0), ("out", M, 1024, 1024, 1), ("fc1", M, 4096, 1024, 2), ("fc2", M, 1024, 4096, 1),
          ("fc2_dgrad", M, 4096, 1024, 3), ("fc1_dgrad", M, 1024, 4096, 0), ("plain_f32", M, 1024, 1024, 4),
          ("cube8k", 8192, 8192, 8192, 0)]
lib.rvlm_k_gemm_set_variant(int(os.environ.get("GEMM_VARIANT", "2")))
if os.environ.get("GEMM_ABLATE"):
    lib.rvlm_k_gemm_set_ablate(int(os.environ["GEMM_ABLATE"]))
only = os.environ.get("TRACE_ONLY")
if only:
    shapes = [s for s in shapes if s[0] in only.split(",")]
g = torch.Generator(device=dev).manual_seed(0)
_wa = torch.randn(8192, 8192, device=dev).bfloat16()     # leave the idle clocks before the first stamp is taken
for _ in range(300):
    torch.matmul(_wa, _wa)
torch.cuda.synchronize()
trace = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dev)
for name, m, n, k, epi in shapes:
    mp = (m + 255) // 256 * 256
    A = torch.randn(mp, k, generator=g, device=dev).bfloat16()
    Bw = (torch.randn(n, k, generator=g, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, generator=g, device=dev)
    res = torch.randn(m, n, generator=g, device=dev) if epi == 1 else None
    hp = torch.randn(m, n, generator=g, device=dev).bfloat16() if epi == 3 else None
    out = torch.empty(m, n, dtype=torch.float32 if epi in (1, 4) else torch.bfloat16, device=dev)
    pre = torch.empty(m, n, dtype=torch.bfloat16, device=dev) if epi == 2 else None

    def run():
        L.check(lib.rvlm_k_gemm_bf16_nt(A.data_ptr(), k, Bw.data_ptr(), k, m, n, k, mp, epi, bias.data_ptr(),
                                        out.data_ptr(), n, L.ptr(pre), L.ptr(hp), L.ptr(res), 0, L.stream_ptr()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    trace.zero_()
    lib.rvlm_k_gemm_set_trace(trace.data_ptr())
    for _ in range(int(os.environ.get("TRACE_REPS", "20"))):    # the stamps of the LAST launch survive: sustained clocks
        run()
    torch.cuda.synchronize()
    lib.rvlm_k_gemm_set_trace(None)
    t = trace.view(256, 8, 4).cpu().double()
    ntile = (m // 256) * (n // 256) // 256
    cal = t[:, 7, :]
    ticks = (cal[:, 2] - cal[:, 0]).mean(); real = (cal[:, 3] - cal[:, 1]).mean()
    ntile_dummy = 0
    ghz = ticks / (real / 100) / 1000
    print(f"   kernel {real / 100:.1f} us, shader clock {ghz:.3f} GHz")
    print(f"{name}: {ntile} tiles/WG, K-steps/tile {k // 64}; columns in shader cycles")
    last = min(ntile, 7) - 1
    tail = (cal[:, 2] - t[:, last, 3])
    head = (t[:, 0, 0] - cal[:, 0])
    print(f"   before the first tile (prologue) {head.mean():7.0f} cycles | after the last tile's epilogue (remainder-row phase + drain) "
          f"{tail.mean():7.0f} cycles (min {tail.min():.0f}, max {tail.max():.0f})")
    for i in range(min(ntile, 7)):
        s = t[:, i, :]
        first = (s[:, 1] - s[:, 0]).mean()
        main = (s[:, 2] - s[:, 0]).mean()
        epi_t = (s[:, 3] - s[:, 2]).mean()
        print(f"   tile {i}: first K-step {first:7.0f} | mainloop {main:8.0f} = {main / (k // 64):6.0f}/K-step (MFMA floor 2048) | "
              f"epilogue {epi_t:7.0f} | at {ghz:.2f} GHz: tile {(main + epi_t) / ghz / 1000:6.1f} us")
lib.rvlm_k_gemm_set_variant(-1)
