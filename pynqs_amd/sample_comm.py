"""Cross-rank merge of the sampler's unique determinants (mirror of PyNQS' Sampler.gather_scatter_sample,
vmc/sample.py:627-772, and of utils/public_function.py:343-360).

The reference gathers (unique, counts, psi) on rank 0, merges there (torch.unique + merge_rank_sample), scatters the
merged shards back and broadcasts the table for the look-up: two collectives through one rank plus a broadcast.
Here every rank all-gathers the three arrays (RCCL all-gather over xGMI, a few MB) and performs the identical,
deterministic merge locally, then keeps its own shard: one exchange step, no rank-0 serialisation.
The shards are the reference's (first n % world_size ranks get one extra, comm.py:108-111) and the merged order is
the reference's (byte-lexicographic torch.unique), so the same walkers land on the same ranks.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from .C_extension import merge_rank_sample, tensor_to_onv
from .distributed import all_gather_varlen, get_rank, get_world_size, shard_bounds
from .public_function import WavefunctionLUT


def torch_unique_index(x: Tensor, dim: int = 0) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """utils/public_function.py:343-360: (unique, inverse, index of the first occurrence, counts)."""
    unique, inverse, counts = torch.unique(x, dim=dim, sorted=True, return_inverse=True, return_counts=True)
    inv_sorted = inverse.argsort(stable=True)
    tot_counts = torch.cat((counts.new_zeros(1), counts.cumsum(dim=0)))[:-1]
    return unique, inverse, inv_sorted[tot_counts], counts


def gather_scatter_sample(unique: Tensor, counts: Tensor, wf_value: Optional[Tensor], sorb: int, use_LUT: bool = True,
                          use_same_tree: bool = True, is_onv: bool = False):
    """vmc/sample.py:627-772.  unique: this rank's distinct samples (0/1 occupations [n, sorb], or packed ONVs with
    is_onv=True), counts int64[n], wf_value psi of them (needed when use_LUT).  Returns
    (unique_rank uint8[n_r, 8*len], placeholder, prob_rank * world_size, WF_LUT or None, merged_counts)."""
    ws, rank = get_world_size(), get_rank()
    onv = unique if is_onv else tensor_to_onv(unique.byte(), sorb)
    dev = onv.device
    sizes = all_gather_varlen(torch.tensor([onv.size(0)], dtype=torch.int64, device=dev))
    onv_all = all_gather_varlen(onv.contiguous())
    count_all = all_gather_varlen(counts.contiguous())
    wf_all = all_gather_varlen(wf_value.contiguous()) if use_LUT else None
    if not use_same_tree:
        # the ranks may have drawn the same determinant: merge duplicates, add their counts
        merge_unique, merge_inv, merge_idx = torch_unique_index(onv_all)[:3]
        wf_unique = wf_all[merge_idx] if use_LUT else None
        split_idx = torch.cat([sizes.new_zeros(1), sizes]).cumsum(0)
        merge_counts = merge_rank_sample(merge_inv.contiguous(), count_all, split_idx, merge_unique.size(0))
    else:
        # every rank sampled a different part of the tree: the concatenation is already duplicate-free
        merge_unique, merge_counts, wf_unique = onv_all, count_all, wf_all
    merge_prob = merge_counts / merge_counts.sum()
    b, e = shard_bounds(merge_unique.size(0), ws, rank)
    lut = WavefunctionLUT(merge_unique, wf_unique, sorb, dev) if use_LUT else None
    placeholder = torch.ones([], device=dev, dtype=torch.int64)
    return merge_unique[b:e].contiguous(), placeholder, merge_prob[b:e] * ws, lut, merge_counts
