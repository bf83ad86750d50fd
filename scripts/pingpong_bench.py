#!/usr/bin/env python3
"""Same-process A/B of the lockstep persistent GEMM (gemm_bf16_256p.hip) and its phase-shifted form (gemm_bf16_256x.hip)
on the encoder's eight launch types (ViT-L/14, B = 128, M = 32 896), alternating, + the 256x timeline of workgroup 0
(per wave: start / end of the MFMAs and end of the epilogue of its first tiles, shader cycles).
usage: python scripts/pingpong_bench.py [reps]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device("cuda:0")
M = 128 * 257
shapes = [("qkv_fwd", M, 3072, 1024, 0), ("out_fwd", M, 1024, 1024, 1), ("fc1_fwd", M, 4096, 1024, 2), ("fc2_fwd", M, 1024, 4096, 1),
          ("fc2_bwd", M, 4096, 1024, 3), ("fc1_bwd", M, 1024, 4096, 0), ("qkv_bwd", M, 1024, 3072, 0), ("out_bwd", M, 1024, 1024, 0)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator(device=dev).manual_seed(0)
_wa = torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(400):            # the GPU leaves idle at a low clock: burn ~0.5 s first
    torch.matmul(_wa, _wa)
torch.cuda.synchronize()
lib.rvlm_k_gemm_set_variant(3)
tot = {0: 0.0, 1: 0.0}
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
    res_ms = {0: [], 1: []}
    for rnd in range(3):
        for pp in (0, 1):
            lib.rvlm_k_gemm_set_pingpong(31 if pp else 0, 1 << 30)
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            res_ms[pp].append(e0.elapsed_time(e1) / reps)
            fam = lib.rvlm_k_gemm_last_kernels()
            assert bool(fam & 64) == bool(pp), (name, pp, fam)
    a, b = min(res_ms[0]), min(res_ms[1])
    tot[0] += a
    tot[1] += b
    fl = 2.0 * m * n * k
    print(f"{name:8s} N={n:5d} K={k:5d} epi={epi}: lockstep {a*1e3:7.1f} us {fl/a/1e9:7.1f} TFLOP/s | phase-shifted {b*1e3:7.1f} us "
          f"{fl/b/1e9:7.1f} TFLOP/s | {100*(a/b-1):+5.1f} %   (runs {[round(x*1e3,1) for x in res_ms[0]]} / {[round(x*1e3,1) for x in res_ms[1]]})", flush=True)
    if os.environ.get("PP_TRACE", "1") == "1":
        tr = torch.zeros(256 * 8 * 26, dtype=torch.int64, device=dev)
        lib.rvlm_k_gemm_x_set_trace(tr.data_ptr())
        lib.rvlm_k_gemm_set_pingpong(31, 1 << 30)
        run()
        torch.cuda.synchronize()
        lib.rvlm_k_gemm_x_set_trace(None)
        t = tr.cpu().view(256, 8, 26)
        for wgi in (0, 100):
            t0 = int(t[wgi, :, 0].min())
            for w in (0, 4):
                row = t[wgi, w]
                tiles = [(int(row[2 + 3 * i] - t0), int(row[3 + 3 * i] - t0), int(row[4 + 3 * i] - t0)) for i in range(8) if int(row[2 + 3 * i])]
                print(f"    wg {wgi} wave {w}: end {int(row[1] - t0)} cycles; tiles (mma start, mma end, epilogue end): {tiles}")
        ends = (t[:, :, 1] - t[:, :, 0].min(dim=1, keepdim=True).values).max(dim=1).values.float()
        print(f"    workgroup durations (cycles): min {int(ends.min())} median {int(ends.median())} max {int(ends.max())}")
print(f"sum over the eight launch types: lockstep {tot[0]*1e3:.1f} us, phase-shifted {tot[1]*1e3:.1f} us ({100*(tot[0]/tot[1]-1):+.1f} %)")
lib.rvlm_k_gemm_set_pingpong(-1, -1)
lib.rvlm_k_gemm_set_variant(-1)
