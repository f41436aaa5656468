"""Fused SAMPLE_SPACE local energy against the size of the sample space: 8192 Fe2S2 walkers, tables of K random determinants (plus the
walkers themselves and a share of their connected determinants, so that a few per cent of the x' are hits)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import C_extension as cx, energy, public_function as pf
from oracle import oracle as O
d = np.load("tests/golden/fe2s2_inputs.npz")
dev = torch.device("cuda")
h1e, h2e = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
n = 8192
x = torch.from_numpy(d["ci_space"][:n].copy()).to(dev)
g = torch.Generator(device="cpu").manual_seed(5)
comb, _ = cx.get_comb_tensor(x[:256].contiguous(), 40, 30, 15, 15)
connected = comb.reshape(-1, 8)[torch.randperm(comb.size(0) * comb.size(1), generator=g)[:200_000].to(dev)]
for logk in (10, 12, 13, 14, 16, 17, 18, 19, 20, 22):
    K = 1 << logk
    occ = torch.zeros((K, 40), dtype=torch.uint8)
    for s in (0, 1):
        idx = torch.rand(K, 20, generator=g).argsort(1)[:, :15]
        occ.scatter_(1, 2 * idx + s, 1)
    keys = torch.unique(torch.cat([cx.tensor_to_onv(occ.to(dev), 40), x, connected[: min(K // 4, connected.size(0))]]), dim=0)
    wf = torch.rand(keys.size(0), dtype=torch.float64, device=dev) + 0.1
    import time
    lut = pf.WavefunctionLUT(keys, wf, 40, device=dev); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        lut = pf.WavefunctionLUT(keys, wf, 40, device=dev)
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) / 3 * 1e3
    f = lambda: energy.local_energy(x, h1e, h2e, None, None, 40, 30, 15, 15, WF_LUT=lut, use_sample_space=True)
    times = {}
    for mode in (False, True):  # column-major (filter first), key-major
        energy.SS_KEYS = mode
        f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            e = f()[0]
        b.record(); torch.cuda.synchronize()
        times[mode] = a.elapsed_time(b) / 10
    ms = times[False]
    # parity of the first 8 walkers against the CPU oracle on the same (sorted) table
    e_ref, _ = O.eloc_sample_space(d["ci_space"][:8].copy(), d["h1e"], d["h2e"], 40, 30, 15, 15,
                                   lut.bra_key.cpu().numpy(), lut.wf_value.cpu().numpy())
    err = float(np.abs(e[:8].cpu().numpy() - e_ref).max())
    assert err < 1e-8, err
    print(f"keys 2^{logk} ({keys.size(0)}): column-major {ms:.3f} ms, key-major {times[True]:.3f} ms per 8192 walkers   (key-major: max |dE| vs oracle on 8 walkers {err:.1e}); table build (sort + hash + filters) {build_ms:.2f} ms", flush=True)
    del lut, keys, wf
