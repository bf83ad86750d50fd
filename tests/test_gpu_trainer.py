"""Outer training step: weight gradients, AdamW and the step semantics against
torch autograd + torch.optim.AdamW on the oracle ViT."""
import numpy as np
import pytest
import torch

import robustvlm_amd as R
from robustvlm_amd import _lib as L
from robustvlm_amd.trainer import AdversarialTrainer, FlatParams, cosine_lr_value
from oracle import vit_ref as V
from oracle.train_ref import TrainStepRef, cosine_lr_ref
from tests.gpu_helpers import dev, cos_sim, rel_max

pytestmark = pytest.mark.gpu
torch.set_num_threads(8)


def to_cfg(c):
    return R.VitConfig(c.image_size, c.patch, c.width, c.layers, c.heads, c.out_dim, c.act)


# width 1024 / 65 tokens / B=8: the encoder-shaped case - weight gradients take the split-K persistent-GEMM path
VIT_WIDE = V.VitConfig(64, 8, 1024, 2, 16, 64)
VIT_B16_SHALLOW = V.VitConfig(64, 16, 768, 2, 12, 512)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
# (VIT_WIDE, 3): 195 tokens - fewer than the copy-free weight gradient takes: every linear goes through the transposing fallback
# with the short leading dimension (the scratch of a W % 256 == 0 handle is sized for conv1 and this case only since round 5)
# (VIT_B16_SHALLOW, 16 / 3): the widths of ViT-B/16 (W = 768, P = 16), 272 / 51 tokens - ADVICE r5: at W = 768 the QKV plan wants 9
# slabs of [3W, W], more than the 16 W^2 floats a trainable handle used to own, so every B/16 training step failed in the fallback
@pytest.mark.parametrize("cfg,B,norm", [(V.VIT_TINY, 4, False), (V.VIT_TINY2, 3, True), (VIT_WIDE, 8, True), (VIT_WIDE, 3, True),
                                        (VIT_B16_SHALLOW, 16, True), (VIT_B16_SHALLOW, 3, False)])
def test_weight_gradients_vs_autograd(cfg, B, norm, precision):
    w = V.init_weights(cfg, seed=11)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g)
    cot = torch.randn(B, cfg.out_dim, generator=g)
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    e = V.vit_forward(cfg, wr, V.normalize_pixels(x))
    if norm:
        e = torch.nn.functional.normalize(e, dim=-1)
    (e * cot).sum().backward()
    eng = R.VitEngine(to_cfg(cfg), {k: v.to(dev()) for k, v in w.items()}, precision=precision, max_batch=B,
                      trainable=True)
    grads = FlatParams(to_cfg(cfg), None, dev())
    emb = eng.forward(x.to(dev()), None, norm, save=2)
    eng.backward_params(cot.to(dev()), grads.views, accumulate=False)
    torch.cuda.synchronize()
    worst = None
    for k, v in wr.items():
        got, ref = grads.views[k].cpu(), v.grad
        if precision == "fp32":
            r = rel_max(got, ref)
            assert r < 2e-3, f"{k}: rel {r}"
        else:
            c = cos_sim(got, ref)
            assert c > 0.97, f"{k}: cos {c}"
    # accumulate=True adds on top
    eng.forward(x.to(dev()), None, norm, save=2)
    eng.backward_params(cot.to(dev()), grads.views, accumulate=True)
    k = "transformer.resblocks.0.mlp.c_fc.weight"
    ref2 = 2 * wr[k].grad
    if precision == "fp32":
        assert rel_max(grads.views[k].cpu(), ref2) < 2e-3
    else:
        assert cos_sim(grads.views[k].cpu(), ref2) > 0.97
    eng.close()


def test_backward_stages_must_run_in_order():
    """rvlm_vit_backward_params_stages hands the residual gradient from stage to stage through the handle -
    a skipped or repeated stage is refused (RVLM_ERR_STATE) instead of writing gradients from stale buffers; stage 0
    restarts the backward; the in-order slices equal the one-call backward bit for bit."""
    cfg = V.VIT_TINY2
    w = V.init_weights(cfg, seed=11)
    g = torch.Generator().manual_seed(2)
    B = 3
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g).to(dev())
    cot = torch.randn(B, cfg.out_dim, generator=g).to(dev())
    eng = R.VitEngine(to_cfg(cfg), {k: v.to(dev()) for k, v in w.items()}, precision="fp32", max_batch=B, trainable=True)
    one, sliced = FlatParams(to_cfg(cfg), None, dev()), FlatParams(to_cfg(cfg), None, dev())
    n_stages = cfg.layers + 2
    eng.forward(x, None, False, save=2)
    eng.backward_params(cot, one.views)
    eng.forward(x, None, False, save=2)
    with pytest.raises(L.RvlmError, match="in order"):
        eng.backward_params(cot, sliced.views, stages=(1, 2))            # stage 0 never ran for this forward
    eng.backward_params(cot, sliced.views, stages=(0, 2))
    with pytest.raises(L.RvlmError, match="in order"):
        eng.backward_params(cot, sliced.views, stages=(1, 3))            # repeats stage 1
    with pytest.raises(L.RvlmError, match="in order"):
        eng.backward_params(cot, sliced.views, stages=(3, n_stages))     # skips stage 2
    eng.backward_params(cot, sliced.views, stages=(0, 1))                # stage 0 restarts the pass
    eng.backward_params(cot, sliced.views, stages=(1, n_stages))
    torch.cuda.synchronize()
    assert torch.equal(one.flat, sliced.flat)
    eng.close()


def test_adamw_kernel_vs_torch():
    l = L.load()
    g = torch.Generator().manual_seed(3)
    n = 10007
    p = torch.randn(n, generator=g); grad = torch.randn(n, generator=g) * 0.1
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-3, weight_decay=1e-2)
    dp, dm, dv = p.to(dev()), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    for step in range(1, 4):
        pt.grad = grad.clone() * step
        opt.step()
        dg = (grad * step * 2).to(dev())          # grad_scale 0.5 undoes the x2
        L.check(l.rvlm_adamw_step(dp.data_ptr(), dg.data_ptr(), dm.data_ptr(), dv.data_ptr(), n, 1e-3, 0.9, 0.999,
                                  1e-8, 1e-2, step, 0.5, L.stream_ptr()))
        torch.cuda.synchronize()
        np.testing.assert_allclose(dp.cpu().numpy(), pt.detach().numpy(), rtol=2e-5, atol=2e-7)


def test_cosine_lr():
    for s in (0, 1, 1399, 1400, 5000, 19999):
        assert cosine_lr_value(s, 1e-5, 1400, 20000) == cosine_lr_ref(s, 1e-5, 1400, 20000)


def test_train_step_fp32_matches_oracle():
    cfg = V.VIT_TINY2
    w = V.init_weights(cfg, seed=21)
    g = torch.Generator().manual_seed(4)
    B = 4
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g)
    tr = AdversarialTrainer(to_cfg(cfg), {k: v.to(dev()) for k, v in w.items()}, batch_size=B, precision="fp32",
                            lr=1e-3, wd=1e-2, warmup=2, steps=10, loss="l2", inner_loss="l2", attack="none",
                            output_normalize=False)
    ref = TrainStepRef(cfg, w, lr=1e-3, wd=1e-2, warmup=2, steps=10, loss="l2")
    with torch.no_grad():
        e0 = V.vit_forward(cfg, w, V.normalize_pixels(x))
    # attack='none' -> data_adv = data (…clip.py:334-335); perturb the inputs a little so the loss is not 0
    xa = (x + 0.02 * torch.rand(x.shape, generator=g)).clamp(0, 1)
    for it in range(3):
        out = tr.train_step(x.to(dev()), None, data_adv=xa.to(dev()))
        loss_ref, _ = ref.step(x, xa, None, e0)
        # the trainer recomputes e0 from its frozen copy (same weights as the oracle's e0)
        assert abs(float(out["loss"]) - loss_ref) <= 2e-3 * abs(loss_ref) + 1e-7, (it, float(out["loss"]), loss_ref)
    sd = tr.state_dict()
    W = cfg.width
    for k, v in ref.w.items():
        got, want = sd[k].cpu(), v.detach()
        if k.endswith("attn.in_proj_bias"):
            # softmax is invariant to the key bias: its true gradient is 0, so both sides feed pure rounding
            # noise to Adam (which normalises it to +-lr steps) - compare the q and v thirds only
            got = torch.cat([got[:W], got[2 * W:]]); want = torch.cat([want[:W], want[2 * W:]])
        r = rel_max(got, want)
        # Adam normalises the step: elements whose gradient is rounding noise move by +-lr either way (three steps at
        # the base LR since round 4: the reference's first step is not a warm-up step)
        assert r < 2e-2, f"{k}: {r}"
    tr.engine.close(); tr.engine_orig.close()


def test_train_step_bf16_runs_and_learns():
    cfg = R.CONFIGS["ViT-B-32"]
    sd = R.random_state_dict(cfg, seed=0, device=dev())
    B = 8
    tr = AdversarialTrainer(cfg, sd, batch_size=B, precision="bf16", lr=2e-6, wd=1e-4, warmup=1, steps=100,
                            attack="pgd", iterations_adv=2)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(B, 3, 224, 224, generator=g, device=dev())
    first = tr.train_step(x, None)                        # full path once: e0, fused PGD, fwd, bwd, AdamW
    assert np.isfinite(float(first["loss"])) and float(first["loss"]) > 0
    xa = (x + (torch.rand(x.shape, generator=g, device=dev()) * 2 - 1) * (4 / 255)).clamp(0, 1)
    losses = [float(tr.train_step(x, None, data_adv=xa)["loss"]) for _ in range(5)]   # fixed adversarial batch
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    tr.engine.close(); tr.engine_orig.close()


# --------------------------------------------------------------------------------------------------------------------
# round 2: loss_clean / trades / logging metrics / periodic eval / checkpoints on the device / data parallel
# --------------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def _tiny_problem(B=4, C=7, seed=4):
    cfg = V.VIT_TINY2
    w = V.init_weights(cfg, seed=21)
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=g)
    xa = (x + 0.03 * (torch.rand(x.shape, generator=g) * 2 - 1)).clamp(0, 1)
    T = torch.nn.functional.normalize(torch.randn(cfg.out_dim, C, generator=g), dim=0)
    y = torch.randint(0, C, (B,), generator=g)
    return cfg, w, x, xa, T, y


@pytest.mark.parametrize("trades", [False, True])
def test_train_step_clean_weight_metrics_vs_oracle(trades):
    """TeCoA with a clean term: loss = ce(adv), loss_clean = --loss_clean (l2 to e0, T=None, …clip.py:341-347),
    loss_total, and the logging metrics cos-sim-clean / cos-sim / acc / racc (…clip.py:368-387) against the oracle's
    restatement; parameters after two steps."""
    cfg, w, x, xa, T, y = _tiny_problem()
    kw = dict(lr=1e-3, wd=1e-2, warmup=2, steps=10, loss="ce", clean_weight=0.3, output_normalize=True)
    tr = AdversarialTrainer(to_cfg(cfg), {k: v.to(dev()) for k, v in w.items()}, batch_size=4, precision="fp32",
                            inner_loss="ce", attack="none", embedding_text_labels_norm=T.to(dev()), loss_clean="l2",
                            trades=trades, **kw)
    ref = TrainStepRef(cfg, w, T=T, loss_clean="l2", trades=trades, **kw)
    with torch.no_grad():
        e0 = torch.nn.functional.normalize(V.vit_forward(cfg, w, V.normalize_pixels(x)), dim=-1)
    for it in range(2):
        out = tr.train_step(x.to(dev()), y.to(dev()), data_adv=xa.to(dev()))
        loss_ref, _ = ref.step(x, xa, y, e0)
        m = ref.last_metrics
        for key, want in (("loss", loss_ref), ("loss_clean", m["loss_clean"]), ("loss_total", m["loss_total"]),
                          ("cos_sim_clean", m["cos_sim_clean"]), ("cos_sim", m["cos_sim"])):
            got = float(out[key])
            assert abs(got - want) <= 2e-3 * abs(want) + 2e-6, (it, key, got, want)
        assert out["acc"] == m["acc"] and out["racc"] == m["racc"], (out["acc"], m["acc"], out["racc"], m["racc"])
    sd = tr.state_dict()
    W = cfg.width
    for k, v in ref.w.items():
        got, want = sd[k].cpu(), v.detach()
        if k.endswith("attn.in_proj_bias"):
            got = torch.cat([got[:W], got[2 * W:]]); want = torch.cat([want[:W], want[2 * W:]])
        assert rel_max(got, want) < 1e-2, k
    tr.close()


def test_eval_step_vs_oracle():
    """The periodic validation of …clip.py:389-424: 50-step supervised APGD with initial_stepsize 0.05 * eps when
    clean_weight > 0, acc / racc / clean-vs-adversarial cosine similarity."""
    from oracle import attacks_ref as A
    from oracle import losses_ref as Lr
    cfg, w, x, _, T, y = _tiny_problem(B=6)
    eps = 4 / 255
    tr = AdversarialTrainer(to_cfg(cfg), {k: v.to(dev()) for k, v in w.items()}, batch_size=6, precision="fp32",
                            loss="ce", inner_loss="ce", attack="none", embedding_text_labels_norm=T.to(dev()),
                            clean_weight=0.5, eps=eps)
    logs = tr.eval_step(x.to(dev()), y.to(dev()))
    assert set(logs) == {"eval/racc", "eval/acc", "eval/cos-sim"}
    model = V.ClipVisionModelRef(cfg, w).eval()
    adv = A.apgd_train_ref(model, x, y, "linf", eps, n_iter=50, initial_stepsize=0.05 * eps,
                           loss_fn=Lr.ComputeLossWrapperRef(None, T, "none", "ce", 100.))
    with torch.no_grad():
        ea, ec = model(adv, True), model(x, True)
        racc, acc = Lr.compute_acc_ref(ea @ T, y), Lr.compute_acc_ref(ec @ T, y)
        cs = float(torch.nn.functional.cosine_similarity(ea, ec, dim=1).mean())
    assert logs["eval/acc"] == acc
    assert abs(logs["eval/racc"] - racc) <= 100 / 6 + 1e-6            # at most one borderline sample differs
    assert abs(logs["eval/cos-sim"] - cs) < 0.02
    assert tr.model.training                                          # back in train mode (…clip.py:424)
    tr.close()


def test_checkpoint_round_trip_on_the_device(tmp_path):
    """SURVEY 8(f) rank 2 on the device: trainer -> CheckpointWriter files -> load_visual_state_dict -> a fresh engine
    gives bit-identical embeddings (plain file and TeCoA container); the optimizer file is torch.optim.AdamW's own
    layout (loads into a real AdamW over visual.parameters()-ordered tensors); resuming from the files reproduces
    the uninterrupted run bit for bit."""
    from robustvlm_amd.config import parameter_order
    cfg, w, x, xa, T, y = _tiny_problem()
    wd = {k: v.to(dev()) for k, v in w.items()}
    kw = dict(batch_size=4, precision="bf16", lr=1e-3, wd=1e-2, warmup=2, steps=10, attack="none")
    xd, xad = x.to(dev()), xa.to(dev())
    tr = AdversarialTrainer(to_cfg(cfg), wd, **kw)
    for _ in range(2):
        tr.train_step(xd, None, data_adv=xad)
    out_dir = str(tmp_path / "run_temp")
    cw = R.CheckpointWriter(out_dir, steps=10)
    final_dir = cw.final(tr.state_dict(), tr.optimizer_state_dict())
    assert final_dir.endswith("run") and not final_dir.endswith("_temp")
    model_file = f"{final_dir}/checkpoints/final.pt"
    sd = R.load_visual_state_dict(model_file, to_cfg(cfg))
    assert list(sd.keys()) == list(R.state_dict_shapes(to_cfg(cfg)).keys())     # visual.state_dict() key order
    eng = R.VitEngine(to_cfg(cfg), sd, precision="bf16", max_batch=4)
    e_file = eng.forward(xd, None, True)
    e_live = tr.engine.forward(xd, None, True)
    assert torch.equal(e_file, e_live)
    torch.save({"vision_encoder_state_dict": sd}, str(tmp_path / "tecoa.pt"))    # CLIP_eval/eval_utils.py:45-48
    sd2 = R.load_visual_state_dict(str(tmp_path / "tecoa.pt"), to_cfg(cfg))
    eng.load_state_dict(sd2)
    assert torch.equal(eng.forward(xd, None, True), e_live)
    eng.close()
    # the optimizer file is what torch.optim.AdamW would have written
    opt_sd = torch.load(f"{final_dir}/checkpoints/final_opt.pt")
    params = [torch.nn.Parameter(sd[k].clone()) for k in parameter_order(to_cfg(cfg))]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-2)
    opt.load_state_dict(opt_sd)
    k5 = parameter_order(to_cfg(cfg))[5]
    o, c = tr.params.offsets[k5]
    assert torch.equal(opt.state[params[5]]["exp_avg"].reshape(-1), tr.exp_avg[o:o + c].cpu())
    assert float(opt.state[params[5]]["step"]) == 2.0
    # resume (…clip.py:98-102,207-208): model + optimizer files, start_step from the file name convention
    torch.save(tr.state_dict(), str(tmp_path / "step_2.pt"))
    torch.save(tr.optimizer_state_dict(), str(tmp_path / "step_2_opt.pt"))
    mpath, opath = R.resume_paths(str(tmp_path / "step_2_opt.pt"), 2)
    tr2 = AdversarialTrainer(to_cfg(cfg), wd, **kw)
    tr2.load_state_dict(R.load_visual_state_dict(mpath, to_cfg(cfg)))
    tr2.load_optimizer_state_dict(torch.load(opath), start_step=2)
    a = tr.train_step(xd, None, data_adv=xad)
    b = tr2.train_step(xd, None, data_adv=xad)
    assert float(a["loss"]) == float(b["loss"]) and a["lr"] == b["lr"]
    # model_orig of the resumed trainer was built from the ORIGINAL weights, like the reference's (:95-97)
    assert torch.equal(tr.params.flat, tr2.params.flat)
    tr.close(); tr2.close()


def test_data_parallel_step_two_ranks_one_gpu(tmp_path):
    """Two ranks (one process each, gloo rendezvous on 127.0.0.1) share the one GPU of the test box: the sharded FARE
    PGD equals the unsharded one, and AdversarialTrainer's data-parallel step (bucketed gradient all-reduce between
    the backward stages, uneven shards 3 + 2) equals the single-process step on the concatenated batch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "dp")
    os.makedirs(out)
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(root, "tests", "dp_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = [torch.load(os.path.join(out, f"rank{i}.pt")) for i in range(2)]
    single = torch.load(os.path.join(out, "single.pt"))
    # attack: per-sample, no collective - the ranks' shards put together are the unsharded result
    assert torch.equal(torch.cat([res[0]["x_adv"], res[1]["x_adv"]]), single["x_adv"])
    # training step: same parameters on both ranks, equal to the single-process step (fp32: summation order only)
    W = V.VIT_TINY2.width
    for k, want in single["params"].items():
        assert torch.equal(res[0]["params"][k], res[1]["params"][k]), k
        got, p0 = res[0]["params"][k], single["params0"][k]
        if k.endswith("attn.in_proj_bias"):     # key bias: true gradient 0, Adam turns rounding noise into +-lr steps
            got, want, p0 = (torch.cat([t[:W], t[2 * W:]]) for t in (got, want, p0))
        moved = float((want - p0).abs().max())
        assert moved > 1e-4, k                  # two Adam steps at lr 1e-3 did move the parameter
        assert float((got - want).abs().max()) < 5e-2 * moved, (k, float((got - want).abs().max()), moved)
    # logging values with global_metrics=True: the global-batch means the reference's DataParallel would log, on every rank
    for r_ in res:
        assert r_["gb_caught"] is True                    # a wrong global_batch in a LATER step is reported too
        assert abs(r_["loss_global"] - single["loss"]) <= 1e-5 * abs(single["loss"])
        assert abs(r_["cos_global"] - single["cos"]) <= 1e-5 and abs(r_["cos_clean_global"] - single["cos_clean"]) <= 1e-5


def test_bucketed_allreduce_on_rccl_single_rank(tmp_path):
    """The trainer's bucketed asynchronous all-reduce on the REAL collective backend (RCCL, one-rank group on the one GPU
    of the test box): async work handles, RCCL's internal stream and the wait before AdamW - identity reduction, so the
    parameters after three steps equal the non-reducing trainer's bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "rccl.pt")
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_worker.py"), out], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert torch.load(out)["same"]


def test_cabi_allreduce_grads_on_rccl(tmp_path):
    """SURVEY.md 8(b) `rvlm_allreduce_grads`: the C ABI's own RCCL collective (csrc/comm.hip), driven through ctypes in a
    process that never imports torch.distributed - rendezvous token, communicator, in-place fp32 / bf16 sums ordered on a side
    stream, argument errors as status codes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "comm.pt")
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "comm_worker.py"), out], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = torch.load(out)
    assert got["ok"] and got["bad_dtype_rc"] == L.RVLM_ERR_ARG, got


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_cabi_allreduce_grads_two_ranks_sum(tmp_path):
    """a REAL multi-rank sum through rvlm_allreduce_grads (two processes, one GPU each, no torch.distributed), and
    the current-device check of the call."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    idfile = str(tmp_path / "rccl_id.bin")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "comm_worker.py"), str(tmp_path / f"c{r}.pt"), str(r), "2",
                               idfile], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    for r in range(2):
        got = torch.load(str(tmp_path / f"c{r}.pt"))
        assert got["ok"] and got["wrong_dev_rc"] == L.RVLM_ERR_STATE, got


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_train_step_vit_l14_vs_oracle(precision):
    """VERDICT r2 missing 4 / next 4(d): the optimizer step the train bench times (train/adversarial_training_clip.py:
    338-366: adversarial forward, FARE loss, loss.backward(), AdamW) on THE HEADLINE MODEL - ViT-L/14, 24 layers,
    W = 1024, S = 257: weight gradients through the batched split-K persistent GEMM, the class-token tail's backward,
    bucket-ordered flat buffer - against oracle/train_ref.py (torch autograd + torch.optim.AdamW on the CPU), B = 4.
    Per-tensor weight-gradient agreement, the loss, and the parameters after the step."""
    from tests.gpu_helpers import record
    torch.set_num_threads(32)
    cfg = V.VIT_L_14
    w = V.init_weights(cfg, seed=3)
    g = torch.Generator().manual_seed(6)
    B = 4
    x = torch.rand(B, 3, 224, 224, generator=g)
    # "adversarial" images far from the clean ones (independent draws): FARE's cotangent 2 (phi(xa) - phi(x)) / B is then
    # O(|phi|).  With xa = x + U(-eps, eps) it is ~1e-2 |phi|, three times the rounding noise of a bf16 forward, and the
    # test would measure that noise instead of the weight-gradient kernels (first version of this test: cos 0.84)
    xa = torch.rand(x.shape, generator=g)
    lr = 1e-5
    tr = AdversarialTrainer(to_cfg(cfg), {k: v.to(dev()) for k, v in w.items()}, batch_size=B, precision=precision,
                            lr=lr, wd=1e-4, warmup=1, steps=10, loss="l2", inner_loss="l2", attack="none",
                            output_normalize=False, metrics=False)
    ref = TrainStepRef(cfg, w, lr=lr, wd=1e-4, warmup=1, steps=10, loss="l2")
    with torch.no_grad():
        e0 = V.vit_forward(cfg, w, V.normalize_pixels(x))
    out = tr.train_step(x.to(dev()), None, data_adv=xa.to(dev()))
    loss_ref, grads_ref = ref.step(x, xa, None, e0)
    torch.cuda.synchronize()
    W = cfg.width
    worst_cos, worst_key, worst_rel = 1.0, None, 0.0
    for k, gr in grads_ref.items():
        got = tr.grads.views[k].cpu()
        if k.endswith("attn.in_proj_bias"):       # the key bias has a true gradient of 0 (softmax invariance): noise on both sides
            got = torch.cat([got[:W], got[2 * W:]]); gr = torch.cat([gr[:W], gr[2 * W:]])
        c, r = cos_sim(got, gr), rel_max(got, gr)
        if c < worst_cos:
            worst_cos, worst_key = c, k
        worst_rel = max(worst_rel, r)
        if precision == "fp32":
            assert r < 2e-3, f"{k}: rel {r}"
        else:
            assert c > (0.999 if gr.dim() == 2 and gr.numel() >= W * W else 0.99), f"{k}: cos {c}"
    loss_rel = abs(float(out["loss"]) - loss_ref) / abs(loss_ref)
    # parameters after AdamW: Adam's first step moves every element by ~lr * sign(g); compare the UPDATE directions
    sd = tr.state_dict()
    agree, total = 0, 0
    for k in ("transformer.resblocks.0.mlp.c_fc.weight", "transformer.resblocks.23.attn.out_proj.weight",
              "transformer.resblocks.11.attn.in_proj_weight", "conv1.weight", "proj"):
        du_got = (sd[k].cpu() - w[k]).flatten()
        du_ref = (ref.w[k].detach() - w[k]).flatten()
        agree += int((torch.sign(du_got) == torch.sign(du_ref)).sum()); total += du_ref.numel()
        assert float(du_got.abs().max()) <= 1.01 * lr + 1e-4 * lr * float(w[k].abs().max()) + 1e-12
    record(f"train_step_vit_l14_vs_oracle[{precision}]", loss_rel=loss_rel, worst_wgrad_cos=worst_cos,
           worst_wgrad_rel=worst_rel, update_sign_agree=agree / total)
    # measured (profiles/r03_parity_metrics.jsonl): fp32 mode loss_rel 0.0, worst wgrad rel 2.4e-5, update signs 0.999996;
    # bf16 mode loss_rel 2.3e-3 (the bf16 embeddings themselves are 2e-3 relative from the fp32 ones), worst per-tensor
    # wgrad cos 0.99967, update signs 0.9931
    assert loss_rel < (1e-3 if precision == "fp32" else 5e-3), (float(out["loss"]), loss_ref)
    assert agree / total > (0.999 if precision == "fp32" else 0.98), agree / total
    tr.engine.close(); tr.engine_orig.close()
    torch.set_num_threads(8)


# --------------------------------------------------------------------------------------------------------------------
# round 4: the product's train_step / eval_step against the reference's OWN train_one_epoch
# (tests/golden/train_step_tiny.npz, written by tests/golden/make_golden_train.py in the build container)
# --------------------------------------------------------------------------------------------------------------------
TRAIN_CASES = {
    "fare_pgd": dict(attack="pgd", loss="l2", inner_loss="l2", clean_weight=0.0, trades=False, output_normalize=False),
    "tecoa_pgd": dict(attack="pgd", loss="ce", inner_loss="ce", clean_weight=0.0, trades=False, output_normalize=True),
    "none_cw": dict(attack="none", loss="ce", inner_loss="ce", clean_weight=0.5, trades=False, output_normalize=True),
    "apgd_trades_cw": dict(attack="apgd", loss="l2", inner_loss="l2", clean_weight=0.3, trades=True,
                           output_normalize=True),
}


@pytest.mark.parametrize("name", sorted(TRAIN_CASES))
def test_train_step_vs_reference_train_one_epoch(name):
    """AdversarialTrainer.train_step (fp32 mode) on the batches the reference's train_one_epoch (…clip.py:276-486) was
    run on, with the adversarial batches that run produced: the learning rate of every step EXACTLY (step 1 at the
    base LR - the reference calls its scheduler only after a step), losses / cos-sims to fp32 rounding, acc / racc
    equal, parameters after the steps within what Adam's normalised step allows for rounding-noise gradients."""
    from tests.helpers import load_golden, cfg_from_array, weights_from_golden
    z = load_golden("train_step_tiny.npz")
    cfg, w, c = cfg_from_array(z["cfg"]), weights_from_golden(z), TRAIN_CASES[name]
    B = z["x"].shape[1]
    tr = AdversarialTrainer(to_cfg(cfg), {k: v.to(dev()) for k, v in w.items()}, batch_size=6, precision="fp32",
                            lr=float(z["lr"]), wd=float(z["wd"]), warmup=int(z["warmup"]), steps=int(z["steps"]),
                            embedding_text_labels_norm=torch.from_numpy(z["T"]).to(dev()), loss_clean="l2",
                            eps=float(z["eps"]), **c)
    keys, W = list(w), cfg.width
    for s in range(z["x"].shape[0]):
        x, y = torch.from_numpy(z["x"][s]).to(dev()), torch.from_numpy(z["y"][s]).to(dev())
        xa = torch.from_numpy(z[f"{name}::x_adv"][s]).to(dev()) if f"{name}::x_adv" in z.files else x
        out = tr.train_step(x, y, data_adv=xa)
        assert out["lr"] == float(z[f"{name}::lr_used"][s]), (s, out["lr"])
        assert tr.cur_lr == float(z[f"{name}::lr_after"][s])
        for key in ("loss", "loss_total", "cos_sim_clean", "cos_sim"):
            got, want = float(out[key]), float(z[f"{name}::{key}"][s])
            assert abs(got - want) <= 2e-3 * abs(want) + 2e-5, (s, key, got, want)
        assert out["acc"] == float(z[f"{name}::acc"][s]) and out["racc"] == float(z[f"{name}::racc"][s]), s
        if s == 0:
            logs = tr.eval_step(torch.from_numpy(z["x_eval"]).to(dev()), torch.from_numpy(z["y_eval"]).to(dev()))
            want = z[f"{name}::eval"]
            assert logs["eval/acc"] == want[0], (logs, want)
            assert abs(logs["eval/racc"] - want[1]) <= 100 / 6 + 1e-6         # at most one borderline sample differs
            assert abs(logs["eval/cos-sim"] - want[2]) < 0.02
        if f"{name}::w{s + 1}::{keys[0]}" in z.files:
            sd = tr.state_dict()
            for k in keys:
                got, want = sd[k].cpu(), torch.from_numpy(z[f"{name}::w{s + 1}::{k}"])
                if k.endswith("attn.in_proj_bias"):        # key bias: true gradient 0, Adam steps on rounding noise
                    got = torch.cat([got[:W], got[2 * W:]]); want = torch.cat([want[:W], want[2 * W:]])
                assert rel_max(got, want) < 2e-2, (s, k, rel_max(got, want))
    tr.close()
