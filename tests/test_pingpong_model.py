"""Host-side model of gemm_bf16_256x.hip's step schedule (scripts/pingpong_model.py): barrier counts of the two wave
groups agree, every stage is requested once into a free ring slot two steps ahead and covered by a wait before it is
read, every half tile sees K slices 0 .. nk-1 in order.  (A mismatch in the kernel would be a hang or a silent wrong
result on the GPU; the model executes the same programs / cursor rules on the CPU.)"""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pingpong_model", os.path.join(ROOT, "scripts", "pingpong_model.py"))
pm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pm)


@pytest.mark.parametrize("M,N,K,epi", [(32768, 3072, 1024, 0), (32768, 1024, 1024, 1), (32768, 4096, 1024, 2),
                                       (32768, 1024, 4096, 1), (2560, 7680, 1024, 0), (512, 256, 512, 0), (256, 256, 512, 4)])
def test_schedule(M, N, K, epi):
    tm, tn = M // 256, N // 256
    grid = min(tm * tn, 256)
    cover = []
    for wg in range(grid):
        tl = pm.tile_list(wg, grid, tm, tn)
        st = pm.simulate(tl, K, pm.epilogue_steps(epi))
        assert st["none"] == pm.epilogue_steps(epi)          # MFMAs in every step but the last half tile's epilogue
        cover += tl
    assert sorted(cover) == sorted((m * 256, n * 256) for m in range(tm) for n in range(tn))


def test_model_catches_a_broken_schedule():
    tl = pm.tile_list(0, 256, 128, 4)
    with pytest.raises(AssertionError):
        pm.simulate(tl, 1024, 20)        # more epilogue steps than K-steps
