"""Does an HBM-bound LayerNorm run under a persistent MFMA GEMM launched on another stream?
Serial (one stream) vs concurrent (two streams) time of NG GEMMs [32896,1024]x[4096,1024]^T and NL LayerNorms."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from robustvlm_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")
M, K, N = 32896, 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = torch.randn(M + 256, K, device=dev).bfloat16()
Bw = torch.randn(N, K, device=dev).bfloat16()
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
x = torch.randn(M, 1024, device=dev)
gam = torch.ones(1024, device=dev); bet = torch.zeros(1024, device=dev)
y = torch.empty_like(x); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
NG, NL = 20, 100


def gemm(s):
    L.check(lib.rvlm_k_gemm_bf16_nt(A.data_ptr(), K, Bw.data_ptr(), K, M, N, K, M + 256, 0, None, out.data_ptr(), N,
                                    None, None, None, 0, s), "gemm")


def ln(s):
    L.check(lib.rvlm_k_layernorm_fwd_f32(x.data_ptr(), gam.data_ptr(), bet.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), M, 1024, s), "ln")


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t0 = time.time()
while time.time() - t0 < 0.5:
    gemm(s1.cuda_stream)
torch.cuda.synchronize()


def timed(fn):
    torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize(); return (time.time() - t) * 1e3


only_g = timed(lambda: [gemm(s1.cuda_stream) for _ in range(NG)])
only_l = timed(lambda: [ln(s1.cuda_stream) for _ in range(NL)])
serial = timed(lambda: [(gemm(s1.cuda_stream), [ln(s1.cuda_stream) for _ in range(NL // NG)]) for _ in range(NG)])
conc = timed(lambda: [(gemm(s1.cuda_stream), [ln(s2.cuda_stream) for _ in range(NL // NG)]) for _ in range(NG)])
two_g = timed(lambda: [(gemm(s1.cuda_stream), gemm(s2.cuda_stream)) for _ in range(NG // 2)])
print(f"N={N}: gemm x{NG} {only_g:.2f} ms | ln x{NL} {only_l:.2f} ms | serial {serial:.2f} ms | two streams {conc:.2f} ms | "
      f"gemms on two streams {two_g:.2f} ms")
