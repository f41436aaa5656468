"""Weighted Monte-Carlo statistics with all-reduce (mirror of PyNQS' utils/stats/{dist_stats,mc_stats}.py).

<O> = sum_rank sum_i O_i p_i / world_size with p pre-scaled by world_size (vmc/sample.py:772);
var = sum |<O> - O_i|^2 p_i, sd = sqrt(var), se = sd / sqrt(counts)   (dist_stats.py:18-79).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from .distributed import all_reduce_packed, get_world_size


def _wdot(x: Tensor, p: Tensor) -> Tensor:
    if torch.is_complex(x):
        return torch.complex(torch.dot(x.real, p), torch.dot(x.imag, p))
    return torch.dot(x, p)


def dist_mean(x: Tensor, prob: Tensor, world_size: int = 1) -> Tensor:
    assert x.dim() == 1 and prob.dim() == 1
    return all_reduce_packed([_wdot(x, prob)], world_size)[0]


def dist_var(x: Tensor, prob: Tensor, world_size: int = 1) -> Tuple[Tensor, Tensor]:
    mean = dist_mean(x, prob, world_size)
    corr = mean - x
    var = all_reduce_packed([(corr * corr.conj() * prob).sum()], world_size)[0]
    return mean, var


def dist_stats(x: Tensor, prob: Tensor, counts: Optional[int] = None, world_size: int = 1):
    """Two-pass form, identical in arithmetic to the reference (dist_stats.py:59-79)."""
    mean, var = dist_var(x, prob, world_size)
    sd = torch.sqrt(var)
    if counts is None:
        n = torch.tensor([float(x.size(0))], dtype=torch.float64, device=x.device)
        counts = all_reduce_packed([n], 1)[0].item()  # total number of samples over ranks
    se = sd / counts**0.5
    return mean, var, sd, se


_WS: dict = {}  # device -> zero-initialised workspace of pynqs_weighted_moments


def _moments(x: Tensor, prob: Tensor) -> Tensor:
    """[sum p Re x, sum p Im x, sum p |x|^2, sum p] as one float64 tensor: one HIP kernel for float64 / complex128 on
    the GPU (pynqs_weighted_moments), torch ops otherwise."""
    if x.is_cuda and prob.is_cuda and prob.dtype == torch.float64 and x.dtype in (torch.float64, torch.complex128) and x.dim() == 1:
        from . import _native as N

        dev = x.device
        ws = _WS.get(dev)
        if ws is None:
            ws = _WS[dev] = torch.zeros(N.lib().pynqs_moments_workspace() // 8, dtype=torch.float64, device=dev)
        xc, pc = x.contiguous(), prob.contiguous()
        N.check(N.lib().pynqs_weighted_moments(xc.data_ptr(), int(x.is_complex()), pc.data_ptr(), x.numel(), ws.data_ptr(),
                                               torch.cuda.current_stream(dev).cuda_stream), "pynqs_weighted_moments")
        # (a view, no copy kernel: the kernel overwrites these four words on its next call, by which time dist_stats_moments -- the one caller --
        # has reduced and consumed them in stream order)
        return ws[:4]
    w = _wdot(x, prob)
    z = torch.zeros((), dtype=prob.dtype, device=x.device)
    return torch.stack([w.real if torch.is_complex(w) else w, w.imag if torch.is_complex(w) else z,
                        torch.dot((x * x.conj()).real, prob), prob.sum()]).to(torch.float64)


def dist_stats_moments(x: Tensor, prob: Tensor, counts: Optional[int] = None, world_size: int = 1):
    """dist_stats_onepass on the fused moments kernel: 1 kernel + 1 all-reduce of 4 doubles + the closing arithmetic.
    counts defaults to len(x) * world_size (equal shards), so no device-to-host copy is needed."""
    m = _moments(x, prob)
    if counts is None:
        counts = x.size(0) * get_world_size()
    if m.is_cuda:
        # sum over the ranks (no division here: the finishing kernel applies 1 / world_size), then ONE launch
        import torch.distributed as dist

        from . import _native as N

        if get_world_size() > 1:
            dist.all_reduce(m, dist.ReduceOp.SUM)
        out = torch.empty(6, dtype=torch.float64, device=m.device)
        N.check(N.lib().pynqs_stats_finish(m.data_ptr(), 1.0 / world_size, float(counts), out.data_ptr(),
                                           torch.cuda.current_stream(m.device).cuda_stream), "pynqs_stats_finish")
        mean = torch.view_as_complex(out[0:2]).reshape(()) if torch.is_complex(x) else out[0]
        return mean, out[2], out[3], out[4]
    m = all_reduce_packed([m], world_size)[0]
    mean = torch.complex(m[0], m[1]) if torch.is_complex(x) else m[0]
    var = (m[2] - (m[0] * m[0] + m[1] * m[1]) * (2.0 - m[3])).clamp_min(0)
    if counts is None:
        counts = x.size(0) * get_world_size()
    sd = torch.sqrt(var)
    return mean, var, sd, sd / counts**0.5


def dist_stats_onepass(x: Tensor, prob: Tensor, counts: Optional[int] = None, world_size: int = 1):
    """Same quantities from ONE packed all-reduce of (sum p O, sum p |O|^2, sum p[, n]):
    var = E|O|^2 - |E O|^2 (valid when the global weights sum to world_size, i.e. normalised p).
    One message instead of the reference's two all-reduce + barrier pairs; differs from the two-pass
    value only by rounding."""
    one = torch.ones((), dtype=prob.dtype, device=x.device)
    vals = [_wdot(x, prob), torch.dot((x * x.conj()).real, prob), prob.sum() * one]
    if counts is None:
        vals.append(torch.tensor(float(x.size(0)), dtype=prob.dtype, device=x.device) * get_world_size())
    red = all_reduce_packed([v.to(torch.complex128) if torch.is_complex(vals[0]) else v for v in vals], world_size)
    mean, m2, psum = red[0], red[1].real if torch.is_complex(red[1]) else red[1], red[2]
    var = (m2 - (mean * mean.conj()).real * (2.0 - (psum.real if torch.is_complex(psum) else psum))).to(vals[1].dtype)
    if counts is None:
        counts = float((red[3].real if torch.is_complex(red[3]) else red[3]).item())
    sd = torch.sqrt(var.clamp_min(0))
    return mean, var, sd, sd / counts**0.5


class operator_statistics:
    """utils/stats/mc_stats.py:19-54."""

    def __init__(self, x: Tensor, prob: Tensor, counts: Optional[int] = None, operator: Optional[str] = None) -> None:
        self.world_size = get_world_size()
        self.operator = operator if operator is not None else "O"
        mean, var, sd, se = dist_stats(x, prob, counts, self.world_size)
        self.stats_dict = {"mean": mean, "var": var, "sd": sd, "se": se}

    def __getitem__(self, key: str) -> Tensor:
        return self.stats_dict[key]

    def to_dict(self):
        return self.stats_dict

    def __repr__(self) -> str:
        m, se, var = self["mean"], self["se"], self["var"]
        return f"<{self.operator}> = {m.real:.9E} ± {se.real:.3E} [σ² = {var.real:.3E}]"
