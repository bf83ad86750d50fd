"""Worker of tests/test_gpu_trainer.py::test_bucketed_allreduce_on_rccl_single_rank: a ONE-rank process group on the real
RCCL backend (backend "nccl" on ROCm) - the data-parallel trainer's bucketed, asynchronous gradient all-reduce runs through
RCCL's own streams and work handles on this GPU; with one rank the reduction is the identity, so the result must equal
the trainer that does not reduce at all, bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import robustvlm_amd as R                                   # noqa: E402
from robustvlm_amd.trainer import AdversarialTrainer         # noqa: E402
from oracle import vit_ref as V                              # noqa: E402  (weights / config only: test infrastructure)


def main(out):
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    c = V.VIT_TINY2
    cfg = R.VitConfig(c.image_size, c.patch, c.width, c.layers, c.heads, c.out_dim, c.act)
    w = {k: v.to(dev) for k, v in V.init_weights(c, seed=21).items()}
    g = torch.Generator().manual_seed(7)
    x = torch.rand(4, 3, c.image_size, c.image_size, generator=g).to(dev)
    xa = (x + 0.03 * (torch.rand(x.shape, generator=g).to(dev) * 2 - 1)).clamp(0, 1)
    kw = dict(batch_size=4, precision="bf16", lr=1e-3, wd=1e-2, warmup=2, steps=10, attack="none")
    plain = AdversarialTrainer(cfg, w, **kw)
    for _ in range(3):
        plain.train_step(x, None, data_adv=xa)
    ref = plain.params.flat.clone()
    plain.close()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    tr = AdversarialTrainer(cfg, w, n_buckets=3, always_reduce=True, **kw)
    assert tr._device_collectives and len(tr.buckets) == 3
    for _ in range(3):
        tr.train_step(x, None, data_adv=xa)
    torch.cuda.synchronize()
    same = bool(torch.equal(tr.params.flat, ref))
    tr.close()
    dist.destroy_process_group()
    torch.save(dict(same=same), out)


if __name__ == "__main__":
    main(sys.argv[1])
