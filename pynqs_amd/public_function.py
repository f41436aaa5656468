"""Host-side helpers of the local-energy path (mirror of the parts of PyNQS' utils/public_function.py that
sit on the path: SURVEY.md 8a rows a3, a16).  Same names and argument meaning as the reference so that
`vmc/energy`-style code reads the same; all determinant arithmetic goes through pynqs_amd.C_extension.
"""
from __future__ import annotations

from functools import partial
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import C_extension as CX
from .C_extension import get_Num_SinglesDoubles, onv_to_tensor, wavefunction_lut  # noqa: F401 (re-exported)
from .distributed import get_rank, get_world_size

USE_HASH = True  # WavefunctionLUT keeps a GPU hash table next to the sorted keys (reference: USE_HASH = False, :23)


def check_para(bra: Tensor) -> None:
    """utils/public_function.py: onv tensors must be uint8."""
    if bra.dtype != torch.uint8:
        raise Exception(f"The type of bra {bra.dtype} must be torch.uint8")


def split_batch_idx(dim: int, min_batch: int) -> List[int]:
    """utils/public_function.py:695-717: cumulative ends of chunks of `min_batch` (last one shorter)."""
    length = int(np.ceil(dim / min_batch))
    ends = [min(dim, (i + 1) * min_batch) for i in range(length)]
    return ends


def split_length_idx(dim: int, length: int) -> List[int]:
    """utils/public_function.py:720-746: `length` nearly equal parts, the first dim % length one longer."""
    k, res = divmod(dim, length)
    out, acc = [], 0
    for i in range(length):
        acc += k + (1 if i < res else 0)
        out.append(acc)
    return out


def torch_lexsort(keys: List[Tensor], dim: int = -1) -> Tensor:
    """utils/public_function.py:615-648 (np.lexsort semantics: last key is the primary one)."""
    if len(keys) < 2:
        raise ValueError(f"keys must be at least 2 sequences, but {len(keys)=}.")
    idx = keys[0].argsort(dim=dim, stable=True)
    for k in keys[1:]:
        idx = idx.gather(dim, k.gather(dim, idx).argsort(dim=dim, stable=True))
    return idx


def torch_sort_onv(bra: Tensor, little_endian: bool = True) -> Tensor:
    """utils/public_function.py:651-692: argsort of onv rows as little-endian big integers.
    The reference lexsorts byte columns; sorting the 64-bit words (most significant last) is the same order
    and 8x fewer passes."""
    assert bra.dim() == 2
    if not little_endian:
        raise NotImplementedError("Little_endian has not been implemented")
    words = bra.contiguous().view(torch.int64)  # [n, len]; compare as unsigned
    n, L = words.shape
    idx = torch.arange(n, device=bra.device)
    for w in range(L):  # least significant word first, stable sorts
        col = words[idx, w]
        # unsigned order of int64: flip the sign bit
        key = col ^ torch.iinfo(torch.int64).min
        idx = idx[key.argsort(stable=True)]
    return idx


def unique_onv(x: Tensor) -> Tuple[Tensor, Tensor]:
    """(unique rows, inverse) of a uint8 onv batch, like torch.unique(x, dim=0, return_inverse=True) up to the
    ORDER of the unique rows (callers only use rows[inverse]; vmc/energy/flip.py:44-50).  torch's row-wise unique
    sorts with a byte-by-byte comparator; here rows are compared as 64-bit words: one radix sort for one-word
    determinants (3.3x faster on 7e5 Fe2S2 rows), len stable sorts otherwise."""
    assert x.dim() == 2 and x.dtype == torch.uint8 and x.size(1) % 8 == 0
    n, L = x.size(0), x.size(1) // 8
    words = x.contiguous().view(torch.int64)  # [n, L]
    if L == 1 or n == 0:
        u, inv = torch.unique(words.view(-1) if L == 1 else words, return_inverse=True, dim=0 if L > 1 else None)
        return u.view(-1, L).view(torch.uint8).view(-1, 8 * L), inv
    order = torch_sort_onv(x)
    sw = words[order]
    new = torch.ones(n, dtype=torch.bool, device=x.device)
    new[1:] = (sw[1:] != sw[:-1]).any(dim=1)
    group = torch.cumsum(new, 0) - 1
    inv = torch.empty(n, dtype=torch.int64, device=x.device)
    inv[order] = group
    return sw[new].view(torch.uint8).view(-1, 8 * L), inv


class WavefunctionLUT:
    """utils/public_function.py:749-868: sorted (onv -> psi) table with binary-search lookup.
    Lookup runs on the GPU through pynqs_amd.C_extension.wavefunction_lut."""

    def __init__(self, bra_key: Tensor, wf_value: Tensor, sorb: int, device=None, sort: bool = True) -> None:
        check_para(bra_key)
        assert bra_key.size(0) == wf_value.size(0)
        self.sort = sort
        if sort:
            idx = torch_sort_onv(bra_key)
            self._bra_key = bra_key[idx].to(device).contiguous()
            self._wf_value = wf_value[idx].to(device)
            self.idx_sorted = torch.argsort(idx, stable=True)
        else:
            self._bra_key = bra_key.to(device).contiguous()
            self._wf_value = wf_value.to(device)
        self.sorb = sorb
        self.hashtable = None
        if USE_HASH and self._bra_key.is_cuda and self._bra_key.size(0) > 0:
            self.hashtable = CX.hash_build(self._bra_key, sorb)  # values = positions in the sorted key array
        self.rank = get_rank()
        self.world_size = get_world_size()
        rank_idx = [0] + split_length_idx(bra_key.size(0), self.world_size)
        self.rank_idx = rank_idx
        self.rank_begin = rank_idx[self.rank]
        self.rank_end = rank_idx[self.rank + 1]

    @property
    def bra_key(self) -> Tensor:
        return self._bra_key

    @property
    def wf_value(self) -> Tensor:
        return self._wf_value

    @property
    def dtype(self):
        return self._wf_value.dtype

    def to(self, device) -> None:
        self._bra_key = self._bra_key.to(device=device)
        self._wf_value = self._wf_value.to(device=device)
        self.hashtable = CX.hash_build(self._bra_key, self.sorb) if (USE_HASH and self._bra_key.is_cuda) else None

    @property
    def memory(self) -> float:
        return self.bra_key.numel() / 2**20

    def lookup(self, onv: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """(indices of onv found, indices not found, psi of the found ones) -- public_function.py:817-838."""
        nbatch = onv.size(0)
        baseline = torch.arange(nbatch, device=onv.device, dtype=torch.int64)
        if self.hashtable is not None and onv.is_cuda:
            idx_array, mask = CX.hash_lookup(self.hashtable, onv)
        else:
            idx_array, mask = wavefunction_lut(self._bra_key, onv, self.sorb)
        idx_array, mask = idx_array.to(onv.device), mask.to(onv.device)
        onv_idx = baseline[mask]
        onv_not_idx = baseline[torch.logical_not(mask)]
        value = self._wf_value[idx_array.masked_select(mask)]
        return onv_idx, onv_not_idx, value

    def index_value(self, begin: int, end: int) -> Tensor:
        assert self.sort, "not-sorted does not support index-value"
        begin = self.rank_begin + begin
        end = self.rank_begin + end
        assert self.rank_end >= end, "Index date must be in the same rank"
        return self.wf_value[self.idx_sorted[begin:end]]

    def clean_memory(self) -> None:
        del self._bra_key, self._wf_value

    def __repr__(self) -> str:
        return (f"{type(self).__name__}(\n    bra-key shape: {tuple(self.bra_key.size())}\n"
                f"    wf-value shape: {self.wf_value.size(0)}\n    sorb: {self.sorb}\n    Memory: {self.memory:.3f} MiB\n)")


def ansatz_batch(func: Callable[[Tensor], Tensor], x: Tensor, batch: int, sorb: int, device, dtype) -> Tensor:
    """utils/public_function.py:934-960: uint8 onv -> +-1 -> func, in chunks of `batch` rows."""
    if x.dtype == torch.uint8:
        convert = partial(onv_to_tensor, sorb=sorb)
    else:
        assert x.size(1) == sorb
        convert = lambda t: t  # noqa: E731
    if batch == -1 or x.size(0) == 0 or batch >= x.size(0):
        return func(convert(x)).to(dtype)
    ends = [0] + split_batch_idx(x.size(0), batch)
    result = torch.empty(x.size(0), device=device, dtype=dtype)
    for a, b in zip(ends[:-1], ends[1:]):
        result[a:b] = func(convert(x[a:b])).to(dtype).view(-1)
    return result


# ---- spin-flip symmetry helpers (utils/public_function.py:966-1018) ----------------------------------
def swap_odd_even_bits_8bit(n: Tensor) -> Tensor:
    return ((n & 0xAA) >> 1) | ((n & 0x55) << 1)


def popcount_8bit(x: Tensor) -> Tensor:
    t = x - ((x >> 1) & 0x55)
    t = (t & 0x33) + ((t >> 2) & 0x33)
    return (t + (t >> 4)) & 0x0F


def spin_flip_sign(x: Tensor, sorb: int) -> Tensor:
    """+1 / -1 for an even / odd number of doubly occupied spatial orbitals."""
    if x.dtype == torch.uint8:
        both = x & swap_odd_even_bits_8bit(x)  # two bits per doubly occupied orbital
        return 2 * ((popcount_8bit(both).sum(dim=-1) & 0b11) == 0).to(torch.int64) - 1
    assert x.size(1) == sorb
    idxs = x[:, ::2] + x[:, 1::2] * 2
    counts = (idxs == 3).sum(dim=1)
    return 1 - counts % 2 * 2


def spin_flip_onv(x: Tensor, sorb: int) -> Tensor:
    """swap alpha <-> beta occupations."""
    if x.dtype == torch.uint8:
        return swap_odd_even_bits_8bit(x)
    assert x.size(1) == sorb
    x1 = torch.empty_like(x)
    x1[:, ::2], x1[:, 1::2] = x[:, 1::2], x[:, ::2]
    return x1


class _SpinProjection:
    """eta = (-1)^(N//2 - S)  (utils/public_function.py:1020-1040)."""

    _eta: Optional[int] = None

    def init(self, N: int, S: int) -> None:
        assert isinstance(N, int) and isinstance(S, int)
        self._eta = (-1) ** (N // 2 - S)

    @property
    def eta(self) -> int:
        if self._eta is None:
            raise NotImplementedError
        return self._eta


SpinProjection = _SpinProjection()
