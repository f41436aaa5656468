"""A complete (tiny) VMC optimisation on the GPU through the drop-in API: H4-sized synthetic problem (sorb = 8,
2 alpha + 2 beta electrons, all 36 determinants enumerated, i.e. exact sampling p(x) = |psi(x)|^2 / sum), real RBM
ansatz, local energies from the fused kernel (pynqs_amd.energy.local_energy -> pynqs_eloc_rbm), statistics by one
kernel + packed all-reduce, gradient by pynqs_amd.grad.grad, Adam.  Run under torchrun for several GPUs (walkers are
sharded with distributed.shard_bounds; the gradient is all-reduced by DDP, the energy by the packed all-reduce).

    python examples/vmc_rbm_exact_sampling.py [steps]
"""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import C_extension as cx, energy, public_function as pf  # noqa: E402
from pynqs_amd.distributed import get_rank, get_world_size, shard_bounds  # noqa: E402
from pynqs_amd.grad import grad  # noqa: E402
from pynqs_amd.rbm import RealRBM  # noqa: E402
from pynqs_amd.stats import dist_stats_moments  # noqa: E402


def all_determinants(sorb, noA, noB):
    occ = []
    for a in itertools.combinations(range(0, sorb, 2), noA):
        for b in itertools.combinations(range(1, sorb, 2), noB):
            v = np.zeros(sorb, dtype=np.uint8)
            v[list(a) + list(b)] = 1
            occ.append(v)
    return np.stack(occ)


def synth_integrals(sorb, seed=1234):
    g = torch.Generator().manual_seed(seed)
    h1 = torch.rand(sorb, sorb, generator=g, dtype=torch.float64) - 0.5
    h1 = (h1 + h1.T).reshape(-1)
    pair = sorb * (sorb - 1) // 2
    h2 = torch.rand(pair * (pair + 1) // 2, generator=g, dtype=torch.float64) - 0.5
    return h1, h2


def run(steps=60, sorb=8, noA=2, noB=2, alpha=2, lr=0.05, log=print):
    torch.set_default_dtype(torch.float64)
    dev = torch.device("cuda", torch.cuda.current_device())
    h1e, h2e = (t.to(dev) for t in synth_integrals(sorb))
    x_all = cx.tensor_to_onv(torch.from_numpy(all_determinants(sorb, noA, noB)).to(dev), sorb)
    b, e = shard_bounds(x_all.size(0), get_world_size(), get_rank())
    x = x_all[b:e].contiguous()
    g = torch.Generator().manual_seed(7)
    model = RealRBM(0.05 * (torch.rand(alpha * sorb, sorb, generator=g) - 0.5), 0.05 * (torch.rand(alpha * sorb, generator=g) - 0.5),
                    0.05 * (torch.rand(sorb, generator=g) - 0.5)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    ab = lambda xx, func: pf.ansatz_batch(func, xx, 1 << 20, sorb, dev, torch.double)
    # exact ground state of the same Hamiltonian in the same determinant space, for reference
    hmat = cx.get_hij_torch(x_all, x_all, h1e, h2e, sorb, noA + noB)
    e0 = float(torch.linalg.eigvalsh(hmat)[0])
    hist = []
    for it in range(steps):
        eloc, _, psi, _ = energy.local_energy(x, h1e, h2e, model, ab, sorb, noA + noB, noA, noB)
        w = psi.abs() ** 2
        norm = w.sum()
        if get_world_size() > 1:
            torch.distributed.all_reduce(norm)
        prob = w / norm * get_world_size()  # pre-scaled by world_size like vmc/sample.py:772
        mean, var, sd, se = dist_stats_moments(eloc, prob, counts=x_all.size(0), world_size=get_world_size())
        opt.zero_grad()
        grad(model, pf.onv_to_tensor(x, sorb), prob, eloc, mean, 1.0, torch.double)
        opt.step()
        hist.append(float(mean))
        if it % 10 == 0 or it == steps - 1:
            log(f"step {it:3d}  <E> = {float(mean):+.8f}  var = {float(var):.3e}   (exact ground state {e0:+.8f})")
    return hist, e0


if __name__ == "__main__":
    if "RANK" in os.environ:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 60)
