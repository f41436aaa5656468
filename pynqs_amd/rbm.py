"""Minimal real RBM amplitude used by the benchmarks and tests as the psi(x) callable of the local-energy
path (same formula as PyNQS' vmc/ansatz/rbm/rbm.py:186-211, rbm_type "real"; parameters are supplied, not
re-initialised).  Ansatz families themselves are outside this package's scope: any nn.Module with
forward(x: +-1 float[n, sorb]) -> psi[n] works with pynqs_amd.energy."""
from __future__ import annotations

import torch
from torch import Tensor, nn


class RealRBM(nn.Module):
    def __init__(self, weights: Tensor, hidden_bias: Tensor, visible_bias: Tensor) -> None:
        super().__init__()
        self.weights = nn.Parameter(weights.clone())            # [num_hidden, sorb]
        self.hidden_bias = nn.Parameter(hidden_bias.clone())    # [num_hidden]
        self.visible_bias = nn.Parameter(visible_bias.clone())  # [sorb]

    def forward(self, x: Tensor) -> Tensor:
        x = x.to(self.weights.dtype)
        ax = torch.mv(x, self.visible_bias).exp()
        amp = (2 * (torch.mm(x, self.weights.T) + self.hidden_bias).cosh()).prod(-1)
        return ax * amp
