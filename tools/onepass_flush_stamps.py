"""Where a workgroup of the flushing LIST form spends its life (sums over its rounds): needs a library built with -DPYNQS_OP_STAMPS (see
tools/onepass_stamps.py).  usage: PYNQS_AMD_LIB=... python tools/onepass_flush_stamps.py [sorb n_alpha walkers eps [eps_sample]]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from pynqs_amd import C_extension as cx, energy as E, _native as N
sorb, no, n, eps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (80, 20, 4096, 0.49)
ns = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda")
x = B.synth_walkers(n, sorb, no, no, 4321).to(dev)
h1, h2 = (t.to(dev) for t in B.synth_integrals(sorb))
plan = cx.plan_for(h1, h2, sorb, dev).buf
fe, nu = E.reduce_front(x, h1, h2, sorb, 2 * no, no, no, eps, ns, seed=3, want_pm1=False)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3):
    fe.run(x, plan, eps, 3, None)
b.record(); torch.cuda.synchronize()
print(f"sorb {sorb}, {n} walkers, eps {eps}: {a.elapsed_time(b) / 3:.3f} ms per launch, {nu} distinct x', chunks per walker {fe.nchunks}, "
      f"kept per walker {float(fe.seg_count.view(n, -1).sum(1).float().mean()) + fe.fixed * fe.nchunks:.0f}")
lib = ctypes.CDLL(N.LIB_PATH)
out = np.zeros((8192, 16), dtype=np.uint64)
assert lib.pynqs_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
t = out[:min(n, 8192)].astype(np.float64) / 100.0  # us
for k, nm in ((10, "enumeration (incl. waiting for the slowest wave)"), (11, "padding + sort"), (12, "values into sorted order"), (13, "kets, probes, rows, links")):
    print(f"  {nm:50s} {t[:, k].mean():9.1f} us per workgroup")
print(f"  walker tables {np.mean(t[:, 1] - t[:, 0]):.1f} us")
if ns:
    for k, nm in ((5, "tile sums ready"), (6, "tile-level draws, scans"), (7, "draws inside the tiles (second visit of the drawn tiles) + emission"), (8, "drawn records: kets, probes, rows, links")):
        print(f"  {nm:70s} {np.mean(t[:, k + 1] - t[:, k]):9.1f} us per workgroup")
    print(f"  workgroup life {np.mean(t[:, 9] - t[:, 0]):.1f} us")
