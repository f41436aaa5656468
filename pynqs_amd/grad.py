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


class FusedRbmGrad:
    """The estimator of grad() for the reference's RBM amplitudes (vmc/ansatz/rbm/rbm.py:186-211), analytically from the packed
    determinants: ONE kernel pair (pynqs_rbm_grad) instead of the forward + backward of the module -- d loss / d theta_k = 2 Re G_k
    (complex parameters stored as (re, im): (2 Re G_k, -2 Im G_k)), G_k = sum_n conj(p_n (E_loc,n - <E> c_n)) O_k(x_n), O = d ln psi / d theta
    = (x_o, tanh theta_h, tanh theta_h x_o).  Same calling convention and the same cross-rank reduction as GraphedGrad (one all-reduce of the
    flat gradient buffer, mean over the ranks; every p.grad is a view into it), but the walkers come as packed onv (uint8 [n, 8 len]) and
    nothing has a fixed shape.  Modules: pynqs_amd.rbm.RealRBM with rbm_type "real", pynqs_amd.rbm.ComplexRBM (anything else: ValueError --
    use grad() / GraphedGrad).  The sums run in a fixed order: the gradient is bit-reproducible."""

    def __init__(self, nqs: nn.Module, sorb: int) -> None:
        from . import _native as N
        from .rbm import ComplexRBM, RealRBM

        m = getattr(nqs, "module", nqs)
        if isinstance(m, ComplexRBM):
            self.flavour, self.names = N.RBM_COMPLEX, ("params_weights", "params_hidden_bias", "params_visible_bias")
        elif isinstance(m, RealRBM) and getattr(m, "rbm_type", "real") == "real":
            self.flavour, self.names = N.RBM_REAL, ("weights", "hidden_bias", "visible_bias")
        else:
            raise ValueError("FusedRbmGrad: the module is not a real (rbm_type 'real') or complex-parameter RBM")
        self.N, self.module, self.sorb = N, m, sorb
        self.params = [getattr(m, nm) for nm in self.names]
        if any(p.dtype != torch.float64 or not p.is_cuda for p in self.params):
            raise ValueError("FusedRbmGrad: float64 parameters on the GPU")
        dev = self.params[0].device
        self.H = self.params[0].size(0)
        # one buffer for the gradients AND the loss (its last element): one all-reduce per step over the ranks, not two
        self.flat = torch.zeros(sum(p.numel() for p in self.params) + 1, dtype=torch.float64, device=dev)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()
        self.loss = self.flat[o:o + 1]
        self.work = None
        self.events = None

    def __call__(self, onv: Tensor, state_prob: Tensor, eloc: Tensor, e_total, extra_psi_pow=1.0) -> Tensor:
        N, dev = self.N, self.flat.device
        n = onv.size(0)
        if onv.dtype != torch.uint8 or onv.dim() != 2 or onv.size(1) != 8 * ((self.sorb - 1) // 64 + 1) or onv.device != dev:
            raise ValueError("FusedRbmGrad: walkers as packed onv uint8[n, 8 len] on the parameters' device")
        cplx = eloc.is_complex()
        prob = (state_prob.real if state_prob.is_complex() else state_prob).to(torch.float64).contiguous()
        el = eloc.to(torch.complex128 if cplx else torch.float64).contiguous()
        et = (e_total if isinstance(e_total, Tensor) else torch.as_tensor(e_total)).to(device=dev, dtype=el.dtype).reshape(1).contiguous()
        pw = None
        if isinstance(extra_psi_pow, Tensor):
            if extra_psi_pow.is_complex():
                raise ValueError("FusedRbmGrad: extra_psi_pow must be real")
            pw = extra_psi_pow.to(torch.float64).contiguous()
        elif float(extra_psi_pow) != 1.0:
            pw = torch.full((n,), float(extra_psi_pow), dtype=torch.float64, device=dev)
        need = N.lib().pynqs_rbm_grad_workspace(n, self.sorb, self.H, self.flavour)
        if self.work is None or self.work.numel() * 8 < need:
            self.work = torch.empty(max(need // 8, 1), dtype=torch.float64, device=dev)
        W, hb, vb = (p.detach().contiguous() for p in self.params)
        gw, ghb, gvb = self.views
        N.check(N.lib().pynqs_rbm_grad(onv.contiguous().data_ptr(), n, self.sorb, W.data_ptr(), hb.data_ptr(), vb.data_ptr(), self.H, self.flavour,
                                       prob.data_ptr(), el.data_ptr(), int(cplx), et.data_ptr(), pw.data_ptr() if pw is not None else None,
                                       gw.data_ptr(), ghb.data_ptr(), gvb.data_ptr(), self.loss.data_ptr(), self.work.data_ptr(),
                                       torch.cuda.current_stream(dev).cuda_stream), "pynqs_rbm_grad")
        ws = get_world_size()
        ev = None
        if self.events is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
        if ws > 1:
            import torch.distributed as dist

            dist.all_reduce(self.flat, dist.ReduceOp.SUM)
            self.flat.div_(ws)  # gradients: mean over the ranks (DDP's convention); the loss is a sum: undone below
        if ev is not None:
            ev[1].record()
            self.events.append(ev)
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.loss * ws if ws > 1 else self.loss.clone()
