"""Minimal real RBM amplitude used by the benchmarks and tests as the psi(x) callable of the local-energy
path (same formula as PyNQS' vmc/ansatz/rbm/rbm.py:186-211, rbm_type "real"; parameters are supplied, not
re-initialised).  Ansatz families themselves are outside this package's scope: any nn.Module with
forward(x: +-1 float[n, sorb]) -> psi[n] works with pynqs_amd.energy."""
from __future__ import annotations

import torch
from torch import Tensor, nn


class _SkinnyLinear(torch.autograd.Function):
    """x @ w.T for x [n, k], w [m, k] with n >> m, k (walkers x orbitals against a few dozen hidden units).  The weight
    gradient gz.T @ x is an [m, n] x [n, k] product whose reduction runs over the walkers: rocBLAS picks a kernel without
    split-K for it (two workgroups walk all 8192 rows: 222 us of a 477 us gradient step, profiles/r02_grad_graph_kernels.txt).
    Here it is a batched product over slabs of rows followed by a sum: the same numbers up to the order of additions."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor) -> Tensor:
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, gz: Tensor):
        x, w = ctx.saved_tensors
        gx = gz @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            n = x.size(0)
            slabs = 1
            while slabs < 256 and n % (2 * slabs) == 0 and n // (2 * slabs) >= 32:
                slabs *= 2
            if slabs > 1:
                gw = torch.bmm(gz.reshape(slabs, n // slabs, -1).transpose(1, 2), x.reshape(slabs, n // slabs, -1)).sum(0)
            else:
                gw = gz.t() @ x
        return gx, gw


class RealRBM(nn.Module):
    """RBM amplitudes with real parameters, rbm.py:199-211: rbm_type "real" exp(a.x) prod 2cosh(theta), "tanh" tanh(a.x) prod
    2cosh(theta), "pRBM" exp(i (a.x + sum ln 2cosh(theta))) (complex-valued), "cos" prod cos(theta)."""

    def __init__(self, weights: Tensor, hidden_bias: Tensor, visible_bias: Tensor, rbm_type: str = "real") -> None:
        super().__init__()
        if rbm_type not in ("real", "tanh", "pRBM", "cos"):
            raise ValueError(f"rbm_type {rbm_type!r}")
        self.rbm_type = rbm_type
        self.weights = nn.Parameter(weights.clone())            # [num_hidden, sorb]
        self.hidden_bias = nn.Parameter(hidden_bias.clone())    # [num_hidden]
        self.visible_bias = nn.Parameter(visible_bias.clone())  # [sorb]

    def forward(self, x: Tensor) -> Tensor:
        x = x.to(self.weights.dtype)
        # one GEMM for theta and a.x (the reference calls mv + mm, rbm.py:186-211; rocBLAS' gemv on [M, sorb] costs a
        # third of this forward for M ~ 1e6 rows)
        wext = torch.cat([self.weights, self.visible_bias.unsqueeze(0)], 0)
        bext = torch.cat([self.hidden_bias, self.hidden_bias.new_zeros(1)])
        z = _SkinnyLinear.apply(x, wext) + bext
        if self.rbm_type == "cos":
            c = z[:, :-1].cos()
            # prod() without its host-synchronising backward: sign and magnitude separately
            return (1 - 2 * ((c < 0).sum(-1) % 2)).to(c.dtype) * c.abs().log().sum(-1).exp()
        lncosh = (2 * z[:, :-1].cosh()).log().sum(-1)
        if self.rbm_type == "tanh":
            return z[:, -1].tanh() * lncosh.exp()
        if self.rbm_type == "pRBM":
            return torch.exp(1j * (z[:, -1] + lncosh))
        # exp(a.x + sum ln 2cosh theta): the product of rbm.py:205-206 without prod(), whose backward synchronises with the host
        # (it looks for zeros with nonzero()) and therefore cannot be captured in a HIP graph (pynqs_amd.grad.GraphedGrad)
        return (z[:, -1] + lncosh).exp()


class ComplexRBM(nn.Module):
    """psi(x) = exp(a.x) prod_h 2 cosh(W x + b)_h with complex128 parameters stored as (re, im) pairs, the layout of the
    reference's rbm_type "complex" (rbm.py:61-73,147-168; its own psi() cannot run: it casts x back to float64 before
    the complex mv, rbm.py:198,205).  The complex-valued amplitude of the C4-type configurations (BDG-RNN amplitudes are
    complex128) for the generic local-energy path."""

    def __init__(self, weights: Tensor, hidden_bias: Tensor, visible_bias: Tensor) -> None:
        super().__init__()
        self.params_weights = nn.Parameter(weights.clone())            # [num_hidden, sorb, 2]
        self.params_hidden_bias = nn.Parameter(hidden_bias.clone())    # [num_hidden, 2]
        self.params_visible_bias = nn.Parameter(visible_bias.clone())  # [sorb, 2]

    def forward(self, x: Tensor) -> Tensor:
        # x is real (+-1): theta = x W^T and a.x are REAL GEMMs on the stacked (re, im) parts, [n, sorb] x [sorb, 2H + 2].
        # (As complex128 mm / mv the backward's 40 x 8192 x 40 product runs in a rocBLAS zgemm kernel without split-K:
        # 0.885 ms of a 1.5 ms gradient step for 8192 walkers, profiles/r02_fe2s2_vmc_step_v1.txt; as dgemm it is ~20 us.)
        H = self.params_weights.size(0)
        w = torch.cat([self.params_weights.permute(2, 0, 1).reshape(2 * H, -1), self.params_visible_bias.t()], 0)  # [2H + 2, sorb]
        z = _SkinnyLinear.apply(x.to(w.dtype), w)
        b = torch.view_as_complex(self.params_hidden_bias)
        theta = torch.complex(z[:, :H], z[:, H:2 * H]) + b
        ax = torch.complex(z[:, 2 * H], z[:, 2 * H + 1])
        # exp(sum ln 2cosh) instead of prod: the same value (exp(ln z) = z on every branch), and its backward has no
        # data-dependent host synchronisation (prod's looks for zeros with nonzero()), so the gradient step can be
        # captured in a HIP graph (pynqs_amd.grad.GraphedGrad)
        return (ax + (2 * theta.cosh()).log().sum(-1)).exp()
