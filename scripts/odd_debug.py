import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robustvlm_amd import _lib as L
lib = L.load(); dev = torch.device("cuda:0")
lib.rvlm_k_gemm_set_variant(3)
for (M, N, K) in [(257, 256, 128), (1028, 3072, 1024)]:
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(M, K, generator=g, device=dev).bfloat16()
    Bw = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).bfloat16()
    mp = (M + 255) // 256 * 256
    Ap = torch.zeros(mp, K, dtype=torch.bfloat16, device=dev); Ap[:M] = A
    out = torch.zeros(M, N, dtype=torch.float32, device=dev)
    L.check(lib.rvlm_k_gemm_bf16_nt(Ap.data_ptr(), K, Bw.data_ptr(), K, M, N, K, mp, 4, None, out.data_ptr(), N, None, None, None, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    ref = A.float() @ Bw.float().t()
    err = (out - ref).abs().amax(dim=1)
    odd = torch.zeros(M, dtype=torch.bool, device=dev); odd[256::257] = True
    print(M, N, K, "families", hex(lib.rvlm_k_gemm_last_kernels()), "max err mfma rows", err[~odd].max().item(), "odd rows", err[odd].max().item(), "ref scale", ref.abs().max().item())
    r = 256
    print("  odd row out[:8]", out[r, :8].tolist()); print("  ref        [:8]", ref[r, :8].tolist())
    print("  out[r, 32:36]", out[r, 32:36].tolist(), "ref", ref[r, 32:36].tolist())
    bad = (err > 1e-2).nonzero().flatten()[:10].tolist(); print("  bad rows", bad)
