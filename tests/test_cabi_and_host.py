"""CPU-side checks: the C-ABI library loads and exports every symbol include/*.h declares (no compute
calls without a GPU), the host logic (schedules, config, error mapping), and the N>1 helpers over gloo."""
import os
import re
import subprocess
import sys

import pytest
import torch

import robustvlm_amd as R
from robustvlm_amd import _lib as L
from robustvlm_amd.apgd_train import apgd_schedule
from oracle.attacks_ref import apgd_schedule as ref_schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(rvlm_[a-z0-9_]+)\s*\(", txt))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = L.load()
    declared = _declared("rvlm.h") | _declared("rvlm_kernels.h")
    assert len(declared) >= 25
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/ but not exported by librvlm.so"
    assert declared == set(L.EXPORTED_SYMBOLS), declared ^ set(L.EXPORTED_SYMBOLS)
    assert lib.rvlm_version() == 109


def test_ab_gemm_file_is_generated_from_the_production_file(tmp_path):
    """The A/B library's persistent GEMM is gemm_bf16_256p.hip + a patch (one source): the patch must apply cleanly."""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "robustvlm_amd", "csrc")
    src, pat = os.path.join(csrc, "gemm_bf16_256p.hip"), os.path.join(csrc, "experimental", "gemm_bf16_256p_abl.patch")
    # the Makefile's fallback where patch(1) is missing: a strict applier (no fuzz) - always checked
    out = tmp_path / "abl_py.hip"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "apply_unified_patch.py"), src, pat, str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    if shutil.which("patch"):
        out2 = tmp_path / "abl.hip"
        r = subprocess.run(["patch", "-s", "-o", str(out2), src, pat], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert out2.read_text() == out.read_text()
    text = out.read_text()
    assert "launch_256p_abl" in text and "P_KNOBS" in text
    assert not os.path.exists(os.path.join(csrc, "experimental", "gemm_bf16_256p_abl.hip")), "a second copy of the kernel source"


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    x = torch.rand(2, 3, 8, 8)
    with pytest.raises(L.RvlmError):
        R.pgd(lambda v, output_normalize=False: v, lambda o, t: o.sum(), x, None, "linf", 4 / 255, 1, 1 / 255,
              False, mode="max")
    with pytest.raises(L.RvlmError):
        R.VitEngine("ViT-B-32", {}, precision="bf16", max_batch=2)


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "robustvlm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_schedule_and_configs():
    for n in (1, 2, 5, 10, 50, 100, 1000):
        assert apgd_schedule(n) == ref_schedule(n)
    c = R.CONFIGS["ViT-L-14"]
    assert (c.tokens, c.mlp, c.grid) == (257, 4096, 16)
    n_params = sum(int(torch.tensor(s).prod()) for s in R.state_dict_shapes(c).values())
    assert n_params == 303_966_208
    n_b32 = sum(int(torch.tensor(s).prod()) for s in R.state_dict_shapes(R.CONFIGS["ViT-B-32"]).values())
    assert n_b32 == 87_849_216


def test_error_code_mapping():
    with pytest.raises(ValueError):
        L.check(L.RVLM_ERR_ARG)
    with pytest.raises(NotImplementedError):
        L.check(L.RVLM_ERR_UNSUPPORTED)
    with pytest.raises(L.RvlmError):
        L.check(L.RVLM_ERR_HIP)
    L.check(L.RVLM_OK)


def test_shard_range_partitions():
    from robustvlm_amd.dist import shard_range
    for n, w in ((1024, 8), (130, 4), (7, 3)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["RVLM_ROOT"])
from robustvlm_amd.dist import shard_batch, max_over_ranks, sum_over_ranks, allreduce_mean_
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
x = torch.arange(10.).view(10, 1)
mine = shard_batch(x, rank, world)
tot = sum_over_ranks(float(mine.sum()))
assert tot == 45.0, tot
assert max_over_ranks(1.0 + rank) == float(world)
g = [torch.full((3,), float(rank)), torch.full((2, 2), 2.0 * rank)]
allreduce_mean_(g)
assert torch.allclose(g[0], torch.full((3,), (world - 1) / 2)) and torch.allclose(g[1], torch.full((2, 2), float(world - 1)))
# the trainer's gradient path on CPU tensors: flat buffer in backward-stage order, bucket plan, per-bucket reduce
from robustvlm_amd import VitConfig
from robustvlm_amd.dist import allreduce_sum_span
from robustvlm_amd.trainer import FlatParams, bucket_plan
cfg = VitConfig(32, 8, 64, 3, 1, 16)
gr = FlatParams(cfg, None, "cpu")
for i, (k, v) in enumerate(sorted(gr.views.items())):
    v.fill_(float(i + 1) * (rank + 1))
plan = bucket_plan(cfg.layers + 2, 3)
assert plan[0][0] == 0 and plan[-1][1] == cfg.layers + 2 and all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
for a, b in plan:
    lo, hi = gr.stage_span(a, b)
    assert allreduce_sum_span(gr.flat, lo, hi, None, False) is None
for i, (k, v) in enumerate(sorted(gr.views.items())):
    assert torch.all(v == float(i + 1) * sum(r + 1 for r in range(world))), k
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    procs = []
    port = 29500 + (os.getpid() % 2000)
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RVLM_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


def test_flat_params_layout_and_parameter_order():
    """The flat gradient buffer is laid out in backward-stage order (a bucket = a contiguous slice), every key has a
    view, and parameter_order() (the index space of torch.optim.AdamW's state) covers the same keys with the
    module's own Parameters first (class_embedding, positional_embedding, proj), ln_post last."""
    import torch
    from robustvlm_amd import VitConfig, state_dict_shapes
    from robustvlm_amd.config import parameter_order, backward_stage_keys
    from robustvlm_amd.trainer import FlatParams, bucket_plan
    cfg = VitConfig(32, 8, 64, 4, 1, 16)
    shapes = state_dict_shapes(cfg)
    order = parameter_order(cfg)
    assert sorted(order) == sorted(shapes) and len(set(order)) == len(order)
    assert order[:4] == ["class_embedding", "positional_embedding", "proj", "conv1.weight"]
    assert order[-2:] == ["ln_post.weight", "ln_post.bias"]
    assert order[6].startswith("transformer.resblocks.0.") and order[-3].startswith("transformer.resblocks.3.")
    stages = backward_stage_keys(cfg)
    assert len(stages) == cfg.layers + 2 and stages[0][0] == "proj" and "conv1.weight" in stages[-1]
    assert all(k.startswith("transformer.resblocks.3.") for k in stages[1])
    fp = FlatParams(cfg, {k: torch.full(s, 2.0) for k, s in shapes.items()}, "cpu")
    assert fp.stage_offsets[0] == 0 and fp.stage_offsets[-1] == fp.numel
    for i, keys in enumerate(stages):
        lo, hi = fp.stage_span(i, i + 1)
        for k in keys:
            o, c = fp.offsets[k]
            assert lo <= o and o + c <= hi and o % 4 == 0
    assert list(fp.state_dict().keys()) == list(shapes.keys())        # visual.state_dict() key order on the way out
    assert float(fp.flat.sum()) == 2.0 * sum(torch.Size(s).numel() for s in shapes.values())
    for n in (1, 3, 4, 6, 9):
        plan = bucket_plan(6, n)
        assert plan[0][0] == 0 and plan[-1][1] == 6 and len(plan) == min(n, 6)
        assert all(a[1] == b[0] and a[0] < a[1] for a, b in zip(plan, plan[1:] + [(6, 7)]))


def test_bench_host_helpers():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    model, physical, threads = b.host_cpu_info()
    assert isinstance(model, str) and 1 <= physical <= threads
    assert b.bind_rank_to_numa(0, 1) is None


def test_checkpoint_layout_and_formats(tmp_path):
    """File names / cadence of …clip.py:239-244,467-479 and the two on-disk model formats."""
    import torch
    from robustvlm_amd import CheckpointWriter, load_visual_state_dict, resume_paths, VitConfig, state_dict_shapes
    cfg = VitConfig(32, 8, 64, 1, 1, 16)
    sd = {k: torch.zeros(s) for k, s in state_dict_shapes(cfg).items()}
    out = str(tmp_path / "run_temp")
    w = CheckpointWriter(out, steps=1000)
    for step in (100, 200, 300, 400):
        w.after_step(step, lambda: sd, lambda: {"step": step})
    files = sorted(os.listdir(os.path.join(out, "checkpoints")))
    assert files == ["fallback_400.pt", "fallback_400_opt.pt", "step_100.pt", "step_100_opt.pt", "step_200.pt",
                     "step_200_opt.pt", "step_300.pt", "step_300_opt.pt", "step_400.pt", "step_400_opt.pt"]
    final_dir = w.final(sd, {"step": 1000})
    assert final_dir.endswith("run") and os.path.exists(os.path.join(final_dir, "checkpoints", "final_opt.pt"))
    model_file, opt_file = resume_paths(os.path.join(final_dir, "checkpoints", "step_300_opt.pt"), 300)
    assert model_file.endswith("step_300.pt")
    got = load_visual_state_dict(model_file, cfg)
    assert set(got) == set(sd)
    tecoa = tmp_path / "tecoa.pt"
    torch.save({"vision_encoder_state_dict": sd}, tecoa)                 # CLIP_eval/eval_utils.py:45-46
    assert set(load_visual_state_dict(str(tecoa), cfg)) == set(sd)
    bad = dict(sd); bad.pop("proj")
    with pytest.raises(KeyError):
        load_visual_state_dict(bad, cfg)


# ------------------------------------------------------------------ section 8(f) rank 3: AutoAttack orchestration (host logic)
class _FakeAttack:
    """Stands in for the device attacks on the CPU: 'fools' the samples whose first pixel exceeds a threshold by
    overwriting the image with a constant the toy classifier maps to class 1."""

    def __init__(self, thr):
        self.thr, self.loss, self.seed, self.calls = thr, None, None, []
        self.n_target_classes, self.n_restarts = 3, 1

    def perturb(self, x, y):
        self.calls.append((self.loss, int(x.shape[0])))
        adv = x.clone()
        hit = x[:, 0, 0, 0] > self.thr
        adv[hit] = 0.9
        return adv


def _toy_classifier(x):
    m = x.reshape(x.shape[0], -1).mean(1)
    return torch.stack([1.0 - m, m, torch.zeros_like(m), torch.zeros_like(m) - 1], dim=1) * 10   # class 1 iff mean > 0.5


def test_autoattack_bookkeeping_and_state_resume(tmp_path):
    import torch
    from robustvlm_amd import AutoAttack, EvaluationState
    g = torch.Generator().manual_seed(0)
    x = torch.rand(10, 3, 4, 4, generator=g) * 0.4           # all clean-classified as class 0
    x[:, 0, 0, 0] = torch.linspace(0.05, 0.95, 10)
    y = torch.zeros(10, dtype=torch.long)
    y[9] = 1                                                   # one clean error: never attacked
    aa = AutoAttack(_toy_classifier, norm='Linf', eps=8 / 255, seed=0, verbose=False, version='custom',
                    attacks_to_run=['apgd-ce', 'apgd-t'], device='cpu')
    aa.apgd, aa.apgd_targeted = _FakeAttack(0.7), _FakeAttack(0.4)
    state_file = tmp_path / "state.json"
    x_adv, y_adv = aa.run_standard_evaluation(x, y, bs=4, return_labels=True, state_path=state_file)
    # apgd-ce sees the 9 clean-correct points in chunks of 4, apgd-t only the survivors
    assert aa.apgd.calls == [('ce', 4), ('ce', 4), ('ce', 1)]
    fooled_ce = (x[:9, 0, 0, 0] > 0.7)
    assert sum(n for _, n in aa.apgd_targeted.calls) == int((~fooled_ce).sum())
    fooled = x[:, 0, 0, 0] > 0.4
    fooled[9] = False
    assert torch.equal((x_adv != x).reshape(10, -1).any(1), fooled)
    assert torch.equal(y_adv[fooled], torch.ones(int(fooled.sum()), dtype=torch.long))
    st = EvaluationState.from_disk(state_file)
    assert st.run_attacks == {'apgd-ce', 'apgd-t'} and abs(st.clean_accuracy - 0.9) < 1e-9
    assert torch.equal(st.robust_flags, ~fooled & (y == 0))
    assert abs(st.robust_accuracy - float((~fooled & (y == 0)).float().mean())) < 1e-6
    # resuming with the finished state runs nothing
    aa.apgd.calls.clear(); aa.apgd_targeted.calls.clear()
    aa.run_standard_evaluation(x, y, bs=4, state_path=state_file)
    assert aa.apgd.calls == [] and aa.apgd_targeted.calls == []
    # a different attack list must refuse the state file; unbuilt attacks fail loudly when reached
    bb = AutoAttack(_toy_classifier, eps=8 / 255, seed=0, verbose=False, version='custom', attacks_to_run=['apgd-ce'],
                    device='cpu')
    with pytest.raises(ValueError):
        bb.run_standard_evaluation(x, y, bs=4, state_path=state_file)
    cc = AutoAttack(_toy_classifier, eps=8 / 255, seed=0, verbose=False, version='standard', device='cpu')
    assert cc.attacks_to_run == ['apgd-ce', 'apgd-t', 'fab-t', 'square'] and cc.apgd.n_restarts == 1
    cc.apgd, cc.apgd_targeted = _FakeAttack(2.0), _FakeAttack(2.0)     # fool nothing -> fab-t is reached
    with pytest.raises(NotImplementedError):
        cc.run_standard_evaluation(x, y, bs=4)
    with pytest.raises(ValueError):
        AutoAttack(_toy_classifier, eps=0.1, version='standard', attacks_to_run=['apgd-ce'], verbose=False)


def test_zeroshot_head_and_accuracy_helper():
    import torch
    from robustvlm_amd import zeroshot_head, compute_accuracy_no_dataloader
    g = torch.Generator().manual_seed(1)
    emb = torch.randn(5, 7, 16, generator=g)                  # 5 classes, 7 templates, D = 16
    T = zeroshot_head(emb)
    assert T.shape == (16, 5)
    for c in range(5):
        e = torch.nn.functional.normalize(emb[c], dim=-1).mean(0)
        assert torch.allclose(T[:, c], e / e.norm(), atol=1e-7)
    assert torch.allclose(zeroshot_head(emb[:, 0]), torch.nn.functional.normalize(emb[:, 0], dim=-1).t(), atol=1e-7)
    x = torch.rand(10, 3, 4, 4, generator=g) * 0.4
    y = torch.zeros(10, dtype=torch.long)
    y[:3] = 1
    assert abs(compute_accuracy_no_dataloader(_toy_classifier, x, y, "cpu", batch_size=4) - 0.7) < 1e-9


def test_square_schedule_matches_oracle():
    """The product's closed form of the Square Attack p schedule against the oracle's threshold table (square.py:192-219),
    with and without the rescaling to n_queries."""
    from robustvlm_amd.square import p_selection
    from oracle.square_ref import p_schedule
    for nq in (50, 123, 5000, 10000):
        for resc in (True, False):
            for it in list(range(0, 600)) + [999, 1000, 1001, 2000, 2001, 4000, 4001, 6000, 6001, 8000, 8001, 9999]:
                assert p_selection(it, nq, .8, resc) == p_schedule(it, nq, .8, resc), (it, nq, resc)


def test_comm_abi_argument_checks():
    """rvlm_comm_* / rvlm_allreduce_grads (include/rvlm.h): argument errors are status codes, never a
    crash; no RCCL call is made on this path (no GPU here)."""
    lib = L.load()
    import ctypes as C
    assert lib.rvlm_comm_unique_id(None) == L.RVLM_ERR_ARG
    h = C.c_void_p()
    ident = (C.c_uint8 * L.COMM_ID_BYTES)()
    assert lib.rvlm_comm_create(ident, 2, 2, C.byref(h)) == L.RVLM_ERR_ARG          # rank out of range
    assert lib.rvlm_comm_create(ident, 0, 0, C.byref(h)) == L.RVLM_ERR_ARG          # empty world
    assert lib.rvlm_comm_create(None, 0, 1, C.byref(h)) == L.RVLM_ERR_ARG
    assert lib.rvlm_allreduce_grads(None, None, 0, L.DTYPE_F32, None) == L.RVLM_ERR_ARG
    assert lib.rvlm_comm_info(None, None, None) == L.RVLM_ERR_ARG
    assert lib.rvlm_comm_destroy(None) == L.RVLM_OK                                  # like free(NULL)
