"""Worker of tests/test_gpu_trainer.py::test_cabi_allreduce_grads_on_rccl: the C ABI's own collective (csrc/comm.hip ->
RCCL) in a fresh process WITHOUT torch.distributed: rendezvous token, communicator on this GPU, in-place sums of an fp32
and a bf16 buffer on a side stream, ordering against a producer kernel on that stream.  One rank (the box has one GPU; RCCL
refuses two ranks on one device): the sum over the ranks is the identity, which pins the call sequence, the stream
semantics and the dtype mapping."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from robustvlm_amd import _lib as L     # noqa: E402


def main(out):
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    lib = L.load()
    ident = (C.c_uint8 * L.COMM_ID_BYTES)()
    L.check(lib.rvlm_comm_unique_id(ident), "rvlm_comm_unique_id")
    assert any(ident), "empty rendezvous token"
    comm = C.c_void_p()
    L.check(lib.rvlm_comm_create(ident, 0, 1, C.byref(comm)), "rvlm_comm_create")
    rank, world = C.c_int(-1), C.c_int(-1)
    L.check(lib.rvlm_comm_info(comm, C.byref(rank), C.byref(world)))
    assert (rank.value, world.value) == (0, 1)
    side = torch.cuda.Stream(device=dev)
    g = torch.Generator(device=dev).manual_seed(3)
    res = {}
    with torch.cuda.stream(side):
        for name, dt, code, n in (("f32", torch.float32, L.DTYPE_F32, 3_000_001), ("bf16", torch.bfloat16, L.DTYPE_BF16, 1_000_003)):
            src = torch.randn(n, generator=g, device=dev)
            buf = (src * 2.0).to(dt)                    # produced by a kernel on the SAME stream right before the reduction
            want = buf.clone()
            L.check(lib.rvlm_allreduce_grads(comm, buf.data_ptr(), n, code, side.cuda_stream), "rvlm_allreduce_grads")
            after = buf * 1.0                           # consumer on the same stream: ordered behind the reduction
            res[name] = (want, after)
        L.check(lib.rvlm_allreduce_grads(comm, None, 0, L.DTYPE_F32, side.cuda_stream))     # empty bucket: no-op
    side.synchronize()
    ok = all(bool(torch.equal(a, b)) for a, b in res.values())
    bad_dtype = lib.rvlm_allreduce_grads(comm, res["f32"][0].data_ptr(), 4, 7, side.cuda_stream)
    L.check(lib.rvlm_comm_destroy(comm), "rvlm_comm_destroy")
    torch.save(dict(ok=ok, bad_dtype_rc=int(bad_dtype)), out)


def main_two_ranks(out, rank, world, idfile):
    """One of `world` processes, one GPU each (needs >= world GPUs): rank 0 creates the rendezvous token and publishes it in
    `idfile`; every rank reduces rank-dependent fp32 / bf16 buffers in place and checks the SUM (the one-rank run
    above can only ever see the identity).  Also: a call from a thread whose current device is not the communicator's is
    refused with RVLM_ERR_STATE."""
    import time
    dev = torch.device(f"cuda:{rank}")
    torch.cuda.set_device(dev)
    lib = L.load()
    ident = (C.c_uint8 * L.COMM_ID_BYTES)()
    if rank == 0:
        L.check(lib.rvlm_comm_unique_id(ident), "rvlm_comm_unique_id")
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(ident))
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 120, "rank 0 never published the rendezvous token"
            time.sleep(0.05)
        ident = (C.c_uint8 * L.COMM_ID_BYTES).from_buffer_copy(open(idfile, "rb").read())
    comm = C.c_void_p()
    L.check(lib.rvlm_comm_create(ident, rank, world, C.byref(comm)), "rvlm_comm_create")
    ok = True
    s = torch.cuda.current_stream(dev)
    for dt, code, n in ((torch.float32, L.DTYPE_F32, 3_000_001), (torch.bfloat16, L.DTYPE_BF16, 1_000_003)):
        parts = [(torch.randn(n, generator=torch.Generator(device=dev).manual_seed(10 + r), device=dev) * (r + 1)).to(dt)
                 for r in range(world)]          # every rank can form every rank's buffer: the expected sum is local
        buf = parts[rank].clone()
        L.check(lib.rvlm_allreduce_grads(comm, buf.data_ptr(), n, code, s.cuda_stream), "rvlm_allreduce_grads")
        s.synchronize()
        want = parts[0].float()
        for r in range(1, world):
            want = want + parts[r].float()
        ok = ok and bool(torch.allclose(buf.float(), want.to(dt).float(), rtol=0, atol=0 if dt == torch.float32 and world == 2 else 1e-2))
    wrong_dev_rc = None
    if torch.cuda.device_count() > 1:
        torch.cuda.set_device((rank + 1) % torch.cuda.device_count())
        wrong_dev_rc = int(lib.rvlm_allreduce_grads(comm, buf.data_ptr(), 4, L.DTYPE_BF16, s.cuda_stream))
        torch.cuda.set_device(dev)
    L.check(lib.rvlm_comm_destroy(comm), "rvlm_comm_destroy")
    torch.save(dict(ok=ok, wrong_dev_rc=wrong_dev_rc), out)


if __name__ == "__main__":
    if len(sys.argv) > 2:
        main_two_ranks(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    else:
        main(sys.argv[1])
