"""Energy-gradient estimator (mirror of PyNQS' vmc/grad/energy_grad.py:118-184, the "AD" method).

loss = 2 Re sum_n p(n) conj(ln psi(n)) (E_loc(n) - <E> c(n)); backward() in micro-batches of AD_MAX_DIM
walkers.  Under DistributedDataParallel the micro-batches run inside no_sync() except the last one, whose
backward triggers DDP's bucketed all-reduce (RCCL over xGMI with backend "nccl") -- one gradient exchange
per VMC step, exactly like the reference.
"""
from __future__ import annotations

import contextlib
from typing import Union

import torch
from torch import Tensor, nn

from .distributed import all_reduce_packed, get_world_size
from .public_function import split_batch_idx


def grad(nqs: nn.Module, states: Tensor, state_prob: Tensor, eloc: Tensor, e_total: Union[complex, float, Tensor],
         extra_psi_pow: Union[Tensor, float] = 1.0, dtype=torch.double, AD_MAX_DIM: int = -1, empty_shard_state: "Tensor | None" = None) -> Tensor:
    """Accumulates d<E>/dtheta into the parameters' .grad; returns the all-reduced loss (logging value).
    empty_shard_state: one VALID configuration ([1, sorb], e.g. the first state of the unsharded batch) for the dummy forward a rank
    with an empty shard needs to join DDP's reduction; without it an all-zero row is used, which is not a configuration."""
    device = states.device
    dim = states.size(0)
    loss_sum = torch.zeros(1, device=device, dtype=torch.double)
    batch = dim if (AD_MAX_DIM == -1 or AD_MAX_DIM > dim) else AD_MAX_DIM
    ends = split_batch_idx(dim, batch) if dim > 0 else []

    def batch_loss_backward(begin: int, end: int) -> None:
        nonlocal loss_sum
        state = states[begin:end]
        if state.is_floating_point():
            state = state.requires_grad_()
        log_psi = nqs(state).to(dtype).log()
        if torch.any(torch.isnan(log_psi)):
            raise ValueError("There are negative numbers in the log-psi, please use complex128")
        prob_b = state_prob[begin:end].real.to(dtype)
        eloc_b = eloc[begin:end].to(dtype)
        c = 1.0 if isinstance(extra_psi_pow, float) else extra_psi_pow[begin:end].to(dtype)
        loss = 2 * (log_psi.conj() * (eloc_b - e_total * c) * prob_b).sum().real
        loss.backward()
        loss_sum += loss.detach()

    no_sync = nqs.no_sync if hasattr(nqs, "no_sync") else contextlib.nullcontext
    begin = 0
    with no_sync():
        for end in ends[:-1]:
            batch_loss_backward(begin, end)
            begin = end
    if ends:
        batch_loss_backward(begin, ends[-1])  # gradient synchronisation happens in this backward
    elif get_world_size() > 1 and hasattr(nqs, "no_sync"):
        # an empty shard (fewer unique samples than ranks): the other ranks' last backward waits in DDP's bucketed
        # all-reduce, so this rank must take part with a zero gradient (the reference indexes idx_lst[-1] of an empty list
        # and dies, leaving the others hanging).  A forward through the DDP wrapper arms its reducer.
        dummy = empty_shard_state if empty_shard_state is not None else states.new_zeros((1,) + tuple(states.shape[1:]))
        out = nqs(dummy.reshape((1,) + tuple(states.shape[1:])).to(states.dtype)).to(dtype).sum()
        if not bool(torch.isfinite(torch.view_as_real(out) if out.is_complex() else out).all()):
            # 0 * inf = nan would reach every rank's gradient through the all-reduce
            raise ValueError("empty shard: the ansatz is not finite on the dummy configuration; pass empty_shard_state=<a valid state>")
        (out.real * 0.0).backward()
    return all_reduce_packed([loss_sum], get_world_size())[0]


class GraphedGrad:
    """The same estimator as grad() with the forward + backward of ONE fixed-shape batch captured in a HIP graph.

    For small amplitude modules the autograd step is launch bound (a complex128 RBM on 8192 x 40 inputs: ~150 kernels of a
    few microseconds each, 2.2 ms eager against 0.2 ms for the whole local-energy kernel), so the graph replays it as one
    submission.  The cross-rank reduction DistributedDataParallel would do in its last backward is ONE all-reduce of the
    flat gradient buffer after the replay (RCCL over xGMI; mean over the ranks, DDP's convention).  Parameters keep their
    identity: after the call every p.grad is a view into that buffer, so any torch optimizer works unchanged.
    Restrictions (otherwise use grad()): fixed number of walkers per call, float +-1 states, parameters not re-allocated.
    With a REAL dtype the logarithm of a negative amplitude is nan (grad() raises "negative numbers in the log-psi" there): the replay's
    loss is checked after every call (one scalar read-back; check_nan=False leaves it to the caller) and the gradients are not
    installed when it is nan.  Complex dtypes need no check (ln of a non-zero complex number is finite).
    """

    def __init__(self, nqs: nn.Module, n: int, sorb: int, dtype=torch.double, device=None, use_pow: bool = False, warmup: int = 3,
                 check_nan: "bool | None" = None) -> None:
        m = getattr(nqs, "module", nqs)
        self.module, self.dtype = m, dtype
        self.check_nan = (not dtype.is_complex) if check_nan is None else check_nan
        self.params = [p for p in m.parameters() if p.requires_grad]
        dev = device if device is not None else self.params[0].device
        rdt = dtype.to_real() if dtype.is_complex else dtype
        self.states = torch.zeros((n, sorb), dtype=rdt, device=dev)
        self.prob = torch.zeros(n, dtype=rdt, device=dev)
        self.eloc = torch.zeros(n, dtype=dtype, device=dev)
        self.e_total = torch.zeros((), dtype=dtype, device=dev)
        self.pow = torch.ones(n, dtype=dtype, device=dev) if use_pow else None
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=self.params[0].dtype, device=dev)
        assert all(p.dtype == self.flat.dtype for p in self.params), "one parameter dtype per module"
        views, o = [], 0
        for p in self.params:
            views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.views = views
        self.loss = torch.zeros(1, dtype=torch.double, device=dev)
        self.events = None  # set to a list to collect (before, after) events of the gradient all-reduce of every call
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        # thread_local: other threads (RCCL's watchdog) may touch the HIP runtime while this one captures
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self._body()

    def _body(self) -> None:
        log_psi = self.module(self.states).to(self.dtype).log()
        c = 1.0 if self.pow is None else self.pow
        loss = 2 * (log_psi.conj() * (self.eloc - self.e_total * c) * self.prob).sum().real
        grads = torch.autograd.grad(loss, self.params)
        for v, g in zip(self.views, grads):
            v.copy_(g)
        self.loss.copy_(loss.detach().reshape(1))

    def __call__(self, states: Tensor, state_prob: Tensor, eloc: Tensor, e_total, extra_psi_pow=1.0) -> Tensor:
        self.states.copy_(states)
        self.prob.copy_(state_prob.real if state_prob.is_complex() else state_prob)
        self.eloc.copy_(eloc)
        self.e_total.copy_((e_total if isinstance(e_total, Tensor) else torch.as_tensor(e_total)).reshape(()))
        if self.pow is not None:
            self.pow.copy_(extra_psi_pow)
        self.graph.replay()
        if self.check_nan and bool(torch.isnan(self.loss).any()):
            for p in self.params:
                p.grad = None
            raise ValueError("There are negative numbers in the log-psi, please use complex128")  # (grad()'s message, energy_grad.py:150-151)
        ws = get_world_size()
        ev = None
        if self.events is not None:  # caller wants the GPU-timeline split replay | all-reduce
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
        if ws > 1:
            import torch.distributed as dist

            dist.all_reduce(self.flat, dist.ReduceOp.SUM)
            self.flat.div_(ws)
        if ev is not None:
            ev[1].record()
            self.events.append(ev)
        for p, v in zip(self.params, self.views):
            p.grad = v
        return all_reduce_packed([self.loss], ws)[0]
