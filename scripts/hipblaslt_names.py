import torch
dev = torch.device("cuda:0")
M = 128 * 257
for (m, n, k) in [(8192, 8192, 8192), (M, 3072, 1024), (M, 1024, 4096), (M, 4096, 1024), (M, 1024, 1024)]:
    a = torch.randn(m, k, device=dev).bfloat16()
    w = torch.randn(n, k, device=dev).bfloat16()
    for _ in range(3):
        torch.matmul(a, w.t())
torch.cuda.synchronize()
