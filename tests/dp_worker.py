"""Worker of tests/test_gpu_trainer.py::test_data_parallel_step_two_ranks_one_gpu (launched by torch.distributed.run,
2 ranks, gloo, both on cuda:0).  Writes rank{r}.pt (and rank 0 the single-process reference single.pt)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import robustvlm_amd as R                                   # noqa: E402
from robustvlm_amd.dist import shard_range                   # noqa: E402
from robustvlm_amd.trainer import AdversarialTrainer         # noqa: E402
from oracle import vit_ref as V                              # noqa: E402  (weights / config only: test infrastructure)


def main(out):
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    c = V.VIT_TINY2
    cfg = R.VitConfig(c.image_size, c.patch, c.width, c.layers, c.heads, c.out_dim, c.act)
    w = {k: v.to(dev) for k, v in V.init_weights(c, seed=21).items()}
    g = torch.Generator().manual_seed(7)
    B = 5                                                    # uneven shards: 3 + 2
    x = torch.rand(B, 3, c.image_size, c.image_size, generator=g).to(dev)
    d0 = ((torch.rand(x.shape, generator=g) * 2 - 1) * (4 / 255)).to(dev)
    xa_fix = (x + 0.03 * (torch.rand(x.shape, generator=g).to(dev) * 2 - 1)).clamp(0, 1)
    lo, hi = shard_range(B, rank, world)
    kw = dict(batch_size=B, precision="fp32", lr=1e-3, wd=1e-2, warmup=2, steps=10, attack="none", n_buckets=3)

    def attack(xs, ds):
        eng = R.VitEngine(cfg, w, precision="fp32", max_batch=B)
        model = R.ClipVisionModel(eng).eval()
        e0 = model(xs, False)
        xa = R.pgd(model, R.ComputeLossWrapper(e0, None, "mean", "l2", 100.), xs, None, "linf", 4 / 255, 5, 1 / 255,
                   False, perturbation=ds.clone(), mode="max")
        eng.close()
        return xa.cpu()

    tr = AdversarialTrainer(cfg, w, **kw)
    assert tr.world == 2 and len(tr.buckets) == 3
    losses = []
    for _ in range(2):
        o = tr.train_step(x[lo:hi], None, data_adv=xa_fix[lo:hi], global_metrics=True)
        losses.append(float(o["loss"]))
    # global_batch given: every step's value is checked against the all-reduced shard sizes - the first step at once,
    # later ones without a host sync, reported by the following call
    tr2 = AdversarialTrainer(cfg, w, **kw)
    for _ in range(2):
        tr2.train_step(x[lo:hi], None, data_adv=xa_fix[lo:hi], global_batch=B)
    tr2.train_step(x[lo:hi], None, data_adv=xa_fix[lo:hi], global_batch=B + 1)          # wrong on every rank
    try:
        tr2.train_step(x[lo:hi], None, data_adv=xa_fix[lo:hi], global_batch=B)
        gb_caught = False
    except ValueError as e:
        gb_caught = f"global_batch={B + 1}" in str(e)
    tr2.close()
    torch.save(dict(x_adv=attack(x[lo:hi], d0[lo:hi]), params={k: v.cpu() for k, v in tr.state_dict().items()}, loss_global=losses[-1],
                    cos_global=float(o["cos_sim"]), cos_clean_global=float(o["cos_sim_clean"]), gb_caught=gb_caught),
               os.path.join(out, f"rank{rank}.pt"))
    tr.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:                                            # the single-process reference on the whole batch
        tr1 = AdversarialTrainer(cfg, w, **kw)
        assert tr1.world == 1
        p0 = {k: v.cpu() for k, v in tr1.state_dict().items()}
        for _ in range(2):
            o = tr1.train_step(x, None, data_adv=xa_fix)
        torch.save(dict(x_adv=attack(x, d0), params={k: v.cpu() for k, v in tr1.state_dict().items()}, params0=p0, loss=float(o["loss"]),
                        cos=float(o["cos_sim"]), cos_clean=float(o["cos_sim_clean"])),
                   os.path.join(out, "single.pt"))
        tr1.close()


if __name__ == "__main__":
    main(sys.argv[1])
