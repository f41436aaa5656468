import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def fe2s2():
    d = golden("fe2s2_inputs.npz")
    return {k: (d[k].item() if d[k].ndim == 0 else d[k]) for k in d.files}


def synth_integrals(sorb: int, seed: int = 1234):
    """SURVEY.md 8(d) synthetic dense integrals, packed layout (same recipe as tests/golden/make_golden.py)."""
    import torch

    g = torch.Generator().manual_seed(seed)
    h1 = torch.rand(sorb, sorb, generator=g, dtype=torch.float64) - 0.5
    h1 = (h1 + h1.T).reshape(-1)
    pair = sorb * (sorb - 1) // 2
    h2 = torch.rand(pair * (pair + 1) // 2, generator=g, dtype=torch.float64) - 0.5
    return h1.numpy(), h2.numpy()


def rand_occ(n, sorb, noA, noB, seed):
    g = np.random.default_rng(seed)
    occ = np.zeros((n, sorb), dtype=np.uint8)
    for i in range(n):
        occ[i, 2 * g.permutation(sorb // 2)[:noA]] = 1
        occ[i, 2 * g.permutation(sorb // 2)[:noB] + 1] = 1
    return occ


def pm1_from_onv(onv: np.ndarray, sorb: int) -> np.ndarray:
    """+1 (occupied) / -1 (empty) float64[n, sorb] from packed determinants uint8[n, 8*len] (numpy, for CPU-side tests)."""
    bits = np.unpackbits(np.ascontiguousarray(onv), axis=1, bitorder="little")[:, :sorb]
    return bits.astype(np.float64) * 2.0 - 1.0


GRAD_CASES = [("real", -1, 0), ("real", 5, 1), ("complex", -1, 0), ("complex", 5, 1)]


def grad_case(kind: str, amd: int, use_pow: int, device="cpu", rank: int = 0, world: int = 1):
    """Inputs of one case of tests/golden/grad_fe2s2.npz (vmc/grad/energy_grad.py:118-184 run by the reference under DDP):
    (module, states, prob * world, eloc, e_total, extra_psi_pow, dtype, AD_MAX_DIM, expected gradients by parameter name) for
    `rank`'s contiguous shard of the 32 walkers."""
    import torch

    from pynqs_amd.rbm import ComplexRBM, RealRBM

    g = golden("grad_fe2s2.npz")
    e0 = golden("eloc_e2e_fe2s2.npz")
    d = golden("eloc_flip_multipsi_fe2s2.npz")
    key = f"grad_{kind}_amd{amd}_pow{use_pow}"
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    if kind == "real":
        m = RealRBM(T(e0["W"]), T(e0["hb"]), T(e0["vb"])).to(device)
        names = {"weights": "params_weights", "hidden_bias": "params_hidden_bias", "visible_bias": "params_visible_bias"}
        dt = torch.double
    else:
        m = ComplexRBM(T(d["Wc"]), T(d["hbc"]), T(d["vbc"])).to(device)
        names = {k: k for k in ("params_weights", "params_hidden_bias", "params_visible_bias")}
        dt = torch.complex128
    n = 32
    k, res = divmod(n, world)
    b = rank * k + min(rank, res)
    e = b + k + (1 if rank < res else 0)
    states = T(pm1_from_onv(e0["x"], 40))[b:e]
    prob = T(g[key + "_prob"])[b:e] * world
    eloc = T(g[key + "_eloc"])[b:e]
    powr = T(g[key + "_pow"])[b:e] if use_pow else 1.0
    e_total = T(g[key + "_e_total"])
    want = {ours: g[f"{key}_ws{world}_{theirs}"] for ours, theirs in names.items()}
    return m, states, prob, eloc, e_total, powr, dt, amd, want
