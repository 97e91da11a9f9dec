"""Process-group helpers with the reference's names (utils/distributed.py:5-60).
backend 'nccl' on PyTorch-ROCm is RCCL (xGMI inside a node)."""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_distributed() else 1


def get_rank():
    return dist.get_rank() if is_distributed() else 0


def barrier():
    if is_distributed():
        dist.barrier()


def reduce_tensor(inp):
    """Average of ``inp`` over ranks, valid on rank 0 (reference: distributed.py:26-37)."""
    world = get_world_size()
    if world < 2:
        return inp
    with torch.no_grad():
        out = inp.clone()
        dist.reduce(out, dst=0)
    return out / world


def all_reduce_numpy(array):
    t = torch.from_numpy(array).cuda()
    dist.all_reduce(t)
    return t.cpu().numpy()


@torch.no_grad()
def concat_all_gather(tensor, concat_dim=0):
    """all_gather + cat; no gradient flows (reference: distributed.py:50-60)."""
    if not is_distributed():
        return tensor
    out = [torch.empty_like(tensor) for _ in range(dist.get_world_size())]
    dist.all_gather(out, tensor.contiguous())
    return torch.cat(out, dim=concat_dim)
