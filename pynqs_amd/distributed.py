"""torch.distributed wrappers of the local-energy path (mirror of PyNQS' utils/distributed/comm.py:9-73).

One process per GPU; the process group is created by the launcher (`torchrun`) with backend "nccl" (= RCCL
over xGMI on ROCm) or "gloo" (CPU tests).  The payloads on this path are a handful of scalars, so they are
latency bound: `all_reduce_packed` sends them as ONE buffer with no barrier, where the reference issues one
all-reduce plus one barrier per scalar (comm.py:62-67, dist_stats.py:37,54).
"""
from __future__ import annotations

from typing import List, Sequence, Union

import torch
import torch.distributed as dist
from torch import Tensor


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def synchronize() -> None:
    if get_world_size() > 1:
        dist.barrier()


def all_reduce_tensor(tensors: Union[Tensor, List[Tensor]], op=dist.ReduceOp.SUM, world_size: int = 1,
                      in_place: bool = True):
    """comm.py:40-73: all-reduce then divide by `world_size` (callers pre-scale probabilities by world_size,
    vmc/sample.py:772).  Returns [tensors] when not distributed / the clones when in_place=False."""
    if isinstance(tensors, Tensor):
        tensor_list = [tensors]
    elif isinstance(tensors, list):
        tensor_list = tensors
    else:
        raise TypeError("tensors must be Tensor or List[Tensor]")
    if get_world_size() == 1:
        return [tensors]
    out = []
    for t in tensor_list:
        if not in_place:
            t = t.clone()
        dist.all_reduce(t, op)  # synchronous on the collective's stream; no extra barrier needed
        t.div_(world_size)
        out.append(t)
    if not in_place:
        return out


def all_reduce_packed(values: Sequence[Tensor], world_size: int = 1) -> List[Tensor]:
    """SUM-all-reduce several scalars / small tensors of one dtype as a single message, then divide by
    `world_size` (same convention as all_reduce_tensor).  Complex values travel as (re, im) pairs."""
    flat = [v.reshape(-1) for v in values]
    is_c = [torch.is_complex(v) for v in flat]
    parts = [torch.view_as_real(v).reshape(-1) if c else v for v, c in zip(flat, is_c)]
    dt = parts[0].dtype
    for q in parts[1:]:
        dt = torch.promote_types(dt, q.dtype)
    buf = torch.cat([p.to(dt) for p in parts])
    if get_world_size() > 1:
        dist.all_reduce(buf, dist.ReduceOp.SUM)
        buf.div_(world_size)
    out, o = [], 0
    for v, p, c in zip(values, parts, is_c):
        seg = buf[o:o + p.numel()]
        o += p.numel()
        seg = torch.complex(seg[0::2], seg[1::2]) if c else seg
        out.append(seg.reshape(v.shape).to(v.dtype))
    return out


def shard_bounds(n: int, world_size: int, rank: int):
    """Contiguous walker shard of `rank`: the first n % world_size ranks get one extra walker
    (utils/public_function.py:720-746 / comm.py:108-111)."""
    k, res = divmod(n, world_size)
    begin = rank * k + min(rank, res)
    return begin, begin + k + (1 if rank < res else 0)


def all_gather_varlen(t: Tensor) -> Tensor:
    """Concatenation over the ranks (rank order) of tensors whose first dimension may differ from rank to rank
    (walker shards differ by at most one, shard_bounds): sizes first, then one padded all-gather.  The reference
    gathers to rank 0 and scatters back (comm.py:76-131); every rank keeping the whole array removes that round trip."""
    ws = get_world_size()
    if ws == 1:
        return t
    n = torch.tensor([t.size(0)], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(v.item()) for v in sizes]
    m = max(sizes)
    pad = t if t.size(0) == m else torch.cat([t, t.new_zeros((m - t.size(0),) + tuple(t.shape[1:]))])
    parts = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad.contiguous())
    return torch.cat([q[:k] for q, k in zip(parts, sizes)])
