"""Stream semantics of the asynchronous SyncBatchNorm exchange (models/fused_bn._Exchange; reference managers/BaseManager.py:447-455:
nn.SyncBatchNorm over the process group) on the REAL backend.

``_Exchange`` issues ``dist.all_reduce(t, async_op=True)`` right behind the kernel that produced the partial sums and calls
``work.wait()`` right before the kernel that consumes them.  On RCCL the collective runs on the backend's own stream: at issue time
that stream is ordered behind the CURRENT stream (the producer's), and ``wait()`` orders the stream that is current AT THE TIME OF
THE WAIT behind the collective -- without blocking the host.  With one HIP stream per HRNet branch the current stream is a side
stream, which the two-rank tests (gloo on this one-GPU box: every wait completes on the host) cannot exercise.  A process group
of ONE rank over "nccl" (= RCCL) can, on one GPU: the all-reduce moves no data, but it goes through the same stream bookkeeping.

The producer is made slow (a long spin kernel in front of the write), so a consumer whose stream was NOT ordered behind the collective
reads the old value."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, backend, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import mscs_amd  # noqa: F401
    from mscs_amd.models import fused_bn
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=0, world_size=1)
    res = {"backend": dist.get_backend()}
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    spin = int(2e8)                                   # ~0.1 s of s_sleep at 2 GHz: far longer than any launch latency
    for case in ("same_stream", "consumer_on_other_stream"):
        t = torch.zeros(1024, device=dev)
        out = torch.full((1024,), -1.0, device=dev)
        torch.cuda.synchronize()
        before = fused_bn.COLLECTIVES["host_waits"]
        with torch.cuda.stream(s1):
            torch.cuda._sleep(spin)                   # the producer of the sums is still running ...
            t.fill_(7.0)                              # ... when the exchange is issued behind it
            ex = fused_bn._all_reduce_async(t)
            issued = torch.cuda.Event()
            issued.record()
            host_ran_ahead = not issued.query()       # the host got here while the producer still runs (always, on RCCL)
            if case == "same_stream":
                ex.wait()
                out.copy_(t)                          # the consumer on the producer's stream
        if case == "consumer_on_other_stream":
            with torch.cuda.stream(s2):
                ex.wait()                             # the stream that is current HERE is the one that must wait
                out.copy_(t)
        torch.cuda.synchronize()
        res[case] = {"out_ok": bool((out == 7.0).all().item()), "host_ran_ahead": bool(host_ran_ahead),
                     "host_waits": fused_bn.COLLECTIVES["host_waits"] - before}
    torch.save(res, out_path)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_exchange_wait_orders_the_consumers_stream_behind_the_collective(tmp_path, backend):
    if backend == "nccl" and not torch.distributed.is_nccl_available():
        pytest.skip("no RCCL in this build")
    out = os.path.join(str(tmp_path), "res.pt")
    mp.spawn(_worker, args=(_free_port(), backend, out), nprocs=1, join=True)
    res = torch.load(out)
    assert res["backend"] == backend
    for case in ("same_stream", "consumer_on_other_stream"):
        r = res[case]
        assert r["out_ok"], f"{backend} / {case}: the consumer read the sums before the collective behind their producer had finished"
        if backend == "nccl":
            # RCCL: issue and wait return at once -- the host was ahead of the GPU and no wait blocked it
            assert r["host_ran_ahead"] and r["host_waits"] == 0, (case, r)
        else:
            assert r["host_waits"] == 1, (case, r)      # gloo completes on the host (what the two-rank tests count)
