"""Sharding helpers for one-process-per-GPU runs.

The hot path has NO data-path collective: every BFP block, every element (with precomputed scales), every
M-group is independent, so tensors shard along dim 0 and each rank quantises its own rows.  The only constraint
is alignment of the shard boundary when dim 0 itself carries the block / group / M structure.  The one real
exchange is calibration under row sharding: SmoothQuant's per-input-channel max over rows needs an
`all_reduce(MAX)` of a `[C_in]` fp32 vector (RCCL on GPU, gloo in the CPU tests).
"""
from typing import List, Tuple

import torch


def row_shards(n_rows: int, world: int, multiple: int = 1) -> List[Tuple[int, int]]:
    """[start, end) row range of every rank: contiguous, covering, boundaries on multiples of `multiple`
    (block size for block_dim = 0, group_size for group-quant slabs along ch_axis = 0, M for N:M along dim 0).
    Earlier ranks get the extra units; a rank may be empty when there are fewer units than ranks."""
    if world < 1 or multiple < 1 or n_rows < 0:
        raise ValueError("row_shards: world, multiple >= 1 and n_rows >= 0 required")
    units = -(-n_rows // multiple)
    base, extra = divmod(units, world)
    out, start = [], 0
    for r in range(world):
        u = base + (1 if r < extra else 0)
        end = min(n_rows, start + u * multiple)
        out.append((start, end))
        start = end
    return out


def my_rows(x: torch.Tensor, rank: int, world: int, multiple: int = 1) -> torch.Tensor:
    s, e = row_shards(x.shape[0], world, multiple)[rank]
    return x[s:e]


def _host_staged(t: torch.Tensor, group=None) -> bool:
    """gloo carries host memory: device tensors go through a host copy (harness-only transport, bench.py --dist-backend gloo)"""
    import torch.distributed as dist

    return t.is_cuda and dist.get_backend(group) == "gloo"


def allreduce_max_(v: torch.Tensor, group=None) -> torch.Tensor:
    """in-place MAX all-reduce (per-channel maxabs under row sharding; timing max-over-ranks in the bench)"""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if _host_staged(v, group):
            h = v.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
            v.copy_(h)
        else:
            dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
    return v


def allreduce_min_(v: torch.Tensor, group=None) -> torch.Tensor:
    """in-place MIN all-reduce (the running minimum of a per-tensor MinMaxObserver whose tensor is sharded over the ranks)"""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if _host_staged(v, group):
            h = v.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MIN, group=group)
            v.copy_(h)
        else:
            dist.all_reduce(v, op=dist.ReduceOp.MIN, group=group)
    return v


WORLD = "world"   # `process_group=parallel.WORLD`: the default process group (torch.distributed spells that `None`, which here means "no exchange")


def resolve_group(process_group):
    """the `group=` argument torch.distributed expects for a module's `process_group` attribute"""
    return None if isinstance(process_group, str) and process_group == WORLD else process_group


def gather_rows(shard: torch.Tensor, n_rows: int, world: int, multiple: int = 1, group=None) -> torch.Tensor:
    """all-gather of row shards back into the full tensor (harness-only: result checks)"""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return shard
    sizes = [e - s for s, e in row_shards(n_rows, world, multiple)]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
    pad[: shard.shape[0]] = shard
    if _host_staged(pad, group):
        pad_h = pad.cpu()
        bufs = [torch.empty_like(pad_h) for _ in range(world)]
        dist.all_gather(bufs, pad_h, group=group)
        return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0).to(shard.device)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)
