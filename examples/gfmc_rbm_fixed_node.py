"""Fixed-node Green's-function Monte Carlo on the GPU with the fused step (pynqs_amd.gfmc: pynqs_green_rbm + pynqs_gfmc_sample_rank,
branching by all-gather): H4-sized synthetic problem (sorb = 8, 2 alpha + 2 beta electrons), trial function = a real RBM.

Walkers start from |psi_T|^2 (exact sampling over the 36 determinants), every generation applies G = Lambda - H_FN in the
importance-sampled form (weights w *= beta, move x' ~ G(. <- x) / beta) and is then resampled in proportion to the weights.  The
mixed estimator sum w E_loc / sum w converges to the lowest eigenvalue of the fixed-node Hamiltonian H_FN, which this script also
obtains by diagonalising H_FN in the full determinant space:  E_exact <= E_FN <= E_VMC(psi_T).

    python examples/gfmc_rbm_fixed_node.py [generations] [walkers]        (also under torchrun: walkers are sharded over the ranks)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from examples.vmc_rbm_exact_sampling import all_determinants, synth_integrals  # noqa: E402
from pynqs_amd import C_extension as cx, gfmc, public_function as pf  # noqa: E402
from pynqs_amd.distributed import get_rank, get_world_size  # noqa: E402
from pynqs_amd.rbm import RealRBM  # noqa: E402


def fixed_node_reference(hmat, psi):
    """(E_exact, E_FN, E_VMC) in the full determinant space: H_FN keeps the off-diagonal elements with H_xx' psi(x') / psi(x) < 0 and moves
    the others, weighted with psi(x') / psi(x), onto the diagonal (the sign-flip potential of gfmc/walker.py:199-211)."""
    ratio = psi.unsqueeze(0) / psi.unsqueeze(1)  # [x, x'] = psi(x') / psi(x)
    off = ~torch.eye(hmat.size(0), dtype=torch.bool, device=hmat.device)
    keep = (hmat * ratio < 0) & off
    h_fn = torch.where(keep, hmat, torch.zeros_like(hmat))
    v_sf = (torch.where(off & ~keep, hmat * ratio, torch.zeros_like(hmat))).sum(1)
    h_fn = h_fn + torch.diag(torch.diagonal(hmat) + v_sf)
    e_fn = float(torch.linalg.eigvalsh(h_fn)[0])  # (symmetric: H is, and so is the criterion H_xx' psi(x) psi(x') < 0)
    e_exact = float(torch.linalg.eigvalsh(hmat)[0])
    e_vmc = float((psi @ (hmat @ psi)) / (psi @ psi))
    return e_exact, e_fn, e_vmc, float((torch.diagonal(hmat) + v_sf).max())


def run(generations=120, walkers=8192, burn_in=40, sorb=8, noA=2, noB=2, seed=3, log=print):
    torch.set_default_dtype(torch.float64)
    dev = torch.device("cuda", torch.cuda.current_device())
    nele = noA + noB
    h1e, h2e = (t.to(dev) for t in synth_integrals(sorb))
    x_all = cx.tensor_to_onv(torch.from_numpy(all_determinants(sorb, noA, noB)).to(dev), sorb)
    g = torch.Generator().manual_seed(7)
    trial = RealRBM(0.3 * (torch.rand(2 * sorb, sorb, generator=g) - 0.5), 0.3 * (torch.rand(2 * sorb, generator=g) - 0.5),
                    0.3 * (torch.rand(sorb, generator=g) - 0.5)).to(dev)
    ab = lambda xx, func: pf.ansatz_batch(func, xx, 1 << 20, sorb, dev, torch.double)  # noqa: E731
    with torch.no_grad():
        psi = ab(x_all, trial)
    hmat = cx.get_hij_torch(x_all, x_all, h1e, h2e, sorb, nele)
    e_exact, e_fn, e_vmc, diag_max = fixed_node_reference(hmat, psi)
    Lambda = diag_max + 0.5  # every diagonal kernel Lambda - H_FN(x, x) stays positive
    # this rank's walkers, drawn from |psi_T|^2
    torch.manual_seed(seed + 1000 * get_rank())
    n = walkers // get_world_size()
    x = x_all[torch.multinomial(psi * psi, n, replacement=True)].contiguous()
    w = torch.ones(n, device=dev)
    num = den = 0.0
    for it in range(generations):
        eloc, gk, comb, _, clamped = gfmc.green_kernel(x, Lambda, h1e, h2e, trial, ab, sorb, nele, noA, noB, torch.double, None, True)
        assert not bool(clamped.any())
        if it >= burn_in:  # mixed estimator on the current population (weights are 1 after the resampling below)
            s = torch.stack([(w * eloc).sum(), w.sum()])
            if get_world_size() > 1:
                torch.distributed.all_reduce(s)
            num, den = num + float(s[0]), den + float(s[1])
        x, w, beta, _ = gfmc.sample_update(x, w, comb, gk)
        x = gfmc.branching(x, w)  # resample in proportion to the weights (all ranks together)
        w = torch.ones(n, device=dev)
        if it % 20 == 0 and get_rank() == 0:
            log(f"generation {it:4d}  <beta> = {float(beta.mean()):.5f}  (Lambda - E_FN = {Lambda - e_fn:.5f})")
    e_gfmc = num / den
    if get_rank() == 0:
        log(f"E_exact = {e_exact:+.6f}   E_FN = {e_fn:+.6f}   E_GFMC = {e_gfmc:+.6f}   E_VMC(psi_T) = {e_vmc:+.6f}")
    return e_exact, e_fn, e_gfmc, e_vmc


if __name__ == "__main__":
    if "RANK" in os.environ:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 8192)
