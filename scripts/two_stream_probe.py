"""Two half-batch attack pipelines side by side (two engines of B/2 on two streams) against the one-batch pipeline.
Needs the EXPERIMENTAL library with RVLM_GEMM_MAX_WG=128 for the two-stream arm (each GEMM takes half the CUs)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import robustvlm_amd as R  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "one"
B = 128
dev = torch.device("cuda:0")
cfg = R.CONFIGS["ViT-L-14"]
sd = R.random_state_dict(cfg, seed=0, device=dev)
eps, stepsize = 4 / 255, 1 / 255
g = torch.Generator(device=dev).manual_seed(1000)
x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g, device=dev)
d0 = (torch.rand(x.shape, generator=g, device=dev) * 2 - 1) * eps
y = torch.zeros(B, dtype=torch.long, device=dev)


def make(nb):
    eng = R.VitEngine(cfg, sd, precision="bf16", max_batch=nb, device=dev)
    return R.ClipVisionModel(eng).eval()


if mode == "one":
    m = make(B)
    e0 = m(x, False)
    wrap = R.ComputeLossWrapper(e0, None, "mean", "l2", 100.)

    def step():
        return R.pgd(m, wrap, x, y, "linf", eps, 10, stepsize, False, perturbation=d0, mode="max")
else:
    h = B // 2
    ms = [make(h), make(h)]
    xs, ds, ys = [x[:h].contiguous(), x[h:].contiguous()], [d0[:h].contiguous(), d0[h:].contiguous()], [y[:h], y[h:]]
    refs = [ms[i](xs[i], False) for i in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def step():
        cur = torch.cuda.current_stream()
        outs = []
        for i in range(2):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                # the engine-level loop: R.pgd() reads its flags word back (a host sync) before the second half could be enqueued
                outs.append(ms[i].model.pgd_run(xs[i], ds[i], "l2", "mean", refs[i], ys[i], False, eps, 10, stepsize, 0.9, "max"))
        for i in range(2):
            cur.wait_stream(streams[i])
        return outs

for _ in range(2):
    step()
torch.cuda.synchronize()
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(json.dumps({"mode": mode, "max_wg": os.environ.get("RVLM_GEMM_MAX_WG"), "ms": [round(t, 1) for t in ts],
                  "img_s": round(B / (min(ts) * 1e-3), 1)}))
