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
