"""REDUCE on long rows (more than energy.FRONT_LONG_ROW columns): local_energy uses the one-launch front end while a segment's kept records
fit its LDS list and the multi-pass path once they do not (energy._FRONT_DENSE) -- the same local energies either way (the reference's
_reduce_psi, vmc/energy/eloc.py:205-324, through the generic tensor path as the check)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(sorb, no, n):
    import bench as B
    from pynqs_amd.rbm import RealRBM

    dev = torch.device("cuda")
    x = B.synth_walkers(n, sorb, no, no, 99).to(dev)
    h1, h2 = (t.to(dev) for t in B.synth_integrals(sorb))
    g = torch.Generator().manual_seed(1)
    r = lambda *s: 0.05 * (torch.rand(*s, generator=g, dtype=torch.float64) - 0.5)  # noqa: E731
    return x, h1, h2, RealRBM(r(sorb // 2, sorb), r(sorb // 2), r(sorb)).to(dev)


def test_long_rows_leave_the_front_end_when_the_list_overflows(monkeypatch):
    from pynqs_amd import energy as E, public_function as pf, reduce_front as RF

    sorb, no, n = 80, 20, 320
    x, h1, h2, m = _case(sorb, no, n)
    dev = x.device
    ab = lambda xx, func: pf.ansatz_batch(func, xx, 1 << 20, sorb, dev, torch.float64)  # noqa: E731
    monkeypatch.setattr(E, "_FRONT_DENSE", set())
    monkeypatch.setattr(E, "_FRONTS", {})
    assert E.get_Num_SinglesDoubles(sorb, no, no) + 1 > E.FRONT_LONG_ROW
    limit = RF.list_capacity(n, sorb, 2 * no, no, no, 0)
    assert limit > 0 and E._front_ok(x, h1, sorb, 2 * no, no, no, 0)
    calls = {"front": 0, "multi": 0}
    run, compact = RF.ReduceFrontEnd.run, E.reduce_compact
    monkeypatch.setattr(RF.ReduceFrontEnd, "run", lambda self, *a, **k: (calls.__setitem__("front", calls["front"] + 1), run(self, *a, **k))[1])
    monkeypatch.setattr(E, "reduce_compact", lambda *a, **k: (calls.__setitem__("multi", calls["multi"] + 1), compact(*a, **k))[1])

    def energies(eps, xs=x):
        return E.local_energy(xs, h1, h2, m, ab, sorb, 2 * no, no, no, reduce_psi=True, eps=eps)[0]

    def check(e, eps):
        monkeypatch.setattr(E, "FUSED", False)
        want = energies(eps, x[:6].contiguous())
        monkeypatch.setattr(E, "FUSED", True)
        ok = torch.isfinite(want)
        assert bool((torch.isfinite(e[:6]) == ok).all()) and bool(ok.any())
        assert float((e[:6] - want)[ok].abs().max()) < 1e-8  # Ha

    # sparse: the LIST form of the front end
    e1 = energies(0.4997)
    assert calls["front"] >= 1 and calls["multi"] == 0   # (the first call may repeat itself to grow its buffers)
    nf = calls["front"]
    fe = next(iter(E._FRONTS.values()))
    assert fe.cap_doubles <= limit and not E._FRONT_DENSE
    check(e1, 0.4997)
    # dense: the list overflows, the call is served by the multi-pass path and so are the following ones
    e2 = energies(0.3)
    assert calls["multi"] == 1 and calls["front"] == nf + 1 and len(E._FRONT_DENSE) == 1
    assert not E._front_ok(x, h1, sorb, 2 * no, no, no, 0)
    check(e2, 0.3)
    e3 = energies(0.3)
    assert calls["multi"] == 2 and calls["front"] == nf + 1 and torch.equal(torch.nan_to_num(e3), torch.nan_to_num(e2))
    # total_energy with look-ahead tickets takes the same turn (on walkers whose diagonal survives eps: it refuses NaN)
    fin = torch.isfinite(e2).nonzero().flatten()[:256]
    assert fin.numel() == 256
    monkeypatch.setattr(E, "_FRONT_DENSE", set())
    monkeypatch.setattr(E, "_FRONTS", {})
    nf, nm = calls["front"], calls["multi"]
    et = E.total_energy(x[fin].contiguous(), 128, -1, h1, h2, m, sorb, 2 * no, no, no, reduce_psi=True, eps=0.3)[0]
    assert float((et - e2[fin]).abs().max()) < 1e-9 and len(E._FRONT_DENSE) == 1
    assert calls["multi"] == nm + 2 and calls["front"] > nf   # chunk 0 overflowed its list; chunk 1's ticket was in flight and is dropped
    # forced (FRONT_ROUTE = False): the front end's other form, same numbers
    monkeypatch.setattr(E, "FRONT_ROUTE", False)
    monkeypatch.setattr(E, "_FRONTS", {})
    e4 = energies(0.3)
    both = torch.isfinite(e2) & torch.isfinite(e4)
    assert bool((torch.isfinite(e2) == torch.isfinite(e4)).all()) and float((e4 - e2)[both].abs().max()) < 1e-9


def test_short_rows_stay_on_the_front_end(fe2s2):
    from pynqs_amd import energy as E

    x = torch.from_numpy(fe2s2["ci_space"][:64].copy()).cuda()
    h1 = torch.from_numpy(fe2s2["h1e"]).cuda()
    assert E._long_row_cap(64, h1, 40, 30, 15, 15, 0) is None and E._front_ok(x, h1, 40, 30, 15, 15, 0) and E._front_ok(x, h1, 40, 30, 15, 15, 100)
