#!/usr/bin/env python3
"""Stage 1 of the semi-stochastic REDUCE fixtures (runs on the GPU box: `gpurun -- python tests/golden/dump_reduce_draws.py`).

The reference draws the sub-eps columns with torch.multinomial (vmc/energy/eloc.py:270); pynqs_amd draws them inside the kernel with a
counter-based generator, so the two never pick the same columns by themselves.  To pin the whole semi-stochastic path to the reference
at 1e-8 Ha, the draws of OUR kernel for a fixed seed are written out here (walker, column, hits), and tests/golden/make_golden_r3.py
(development container) feeds exactly these draws to the reference's _reduce_psi / _reduce_psi_flip in place of torch.multinomial.
The GPU tests then re-run local_energy with the same seed and must reproduce the reference's numbers.

Writes gpurun_out/reduce_draws_fe2s2.npz and reduce_draws_bdg_rnn_fe2s2.npz (copy them to tests/golden/): x, eps, eps_sample, torch_seed, kernel_seed, draw_walker, draw_col, draw_hits."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pynqs_amd import energy as E  # noqa: E402

TORCH_SEED, EPS, N = 20240, 1e-2, 200

f = np.load(os.path.join(ROOT, "tests", "golden", "fe2s2_inputs.npz"))
dev = torch.device("cuda")
h1, h2 = torch.from_numpy(f["h1e"]).to(dev), torch.from_numpy(f["h2e"]).to(dev)
sorb, nele, noA, noB = int(f["sorb"]), int(f["nele"]), int(f["noA"]), int(f["noB"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for src, dst in (("eloc_e2e_fe2s2.npz", "reduce_draws_fe2s2.npz"), ("bdg_rnn_walkers_fe2s2.npz", "reduce_draws_bdg_rnn_fe2s2.npz")):
    xs = np.load(os.path.join(ROOT, "tests", "golden", src))["x"]
    x = torch.from_numpy(xs).to(dev)
    torch.manual_seed(TORCH_SEED)
    seed = E._draw_seed()  # what local_energy draws first after torch.manual_seed(TORCH_SEED)
    E._FRONTS.clear()
    fe, nu = E.reduce_front(x, h1, h2, sorb, nele, noA, noB, EPS, N, None, seed=seed)
    walker, col, w, link, onv, drawn = fe.records()
    S = fe.row_sum[: x.size(0)]
    hits = (w[drawn].abs() * N / S[walker[drawn]]).round().long()
    assert bool((torch.zeros(x.size(0), dtype=torch.long, device=dev).index_add_(0, walker[drawn], hits) == N).all())
    out = os.path.join(ROOT, "gpurun_out", dst)
    np.savez_compressed(out, x=xs, eps=EPS, eps_sample=N, torch_seed=TORCH_SEED, kernel_seed=np.int64(seed),
                        draw_walker=walker[drawn].cpu().numpy(), draw_col=col[drawn].cpu().numpy(), draw_hits=hits.cpu().numpy(),
                        n_kept=int((~drawn).sum()), row_sum=S.cpu().numpy())
    print("wrote", out, "draw records", int(drawn.sum()), "kept", int((~drawn).sum()), "seed", seed)
