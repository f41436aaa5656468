"""Per-workgroup phase timestamps of the semi-stochastic REDUCE front end: mean duration of every phase of a workgroup's life, 8192 Fe2S2
walkers.  Needs a library whose kernels_reduce_onepass.hip was compiled with -DPYNQS_OP_STAMPS, e.g.
  cd pynqs_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPYNQS_OP_STAMPS -c kernels_reduce_rowout.hip -o /tmp/s.o &&
  hipcc --offload-arch=gfx950 -fPIC -shared -o ../../build_ab/libpynqs_stamps.so $(ls build/*.o | grep -v kernels_reduce_rowout.o) /tmp/s.o
  PYNQS_AMD_LIB=$PWD/build_ab/libpynqs_stamps.so python tools/onepass_stamps.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import C_extension as cx, reduce_front as RF, _native as N
d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fe2s2_inputs.npz"))
dev = torch.device("cuda"); n = 8192
ci = d["ci_space"]
x = torch.from_numpy(np.ascontiguousarray(ci[np.arange(n) % ci.shape[0]])).to(dev)
h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
plan = cx.plan_for(h1, h2, 40, dev).buf
fe = RF.ReduceFrontEnd(n, 40, 30, 15, 15, 1000, torch.float64, dev, 246, 1900000, want_pm1=False)
for _ in range(3):
    fe.run(x, plan, 1e-2, 3, None)
torch.cuda.synchronize()
lib = ctypes.CDLL(N.LIB_PATH)
out = np.zeros((8192, 16), dtype=np.uint64)
assert lib.pynqs_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
t = out[:, :10].astype(np.float64)
names = ["walker tables", "phase A (enumeration, float32 row, kept list)", "record counts", "sort of the kept list (incl. barrier)", "kept records: kets, probes, rows, links",
         "S", "segment sums of the row read back, their scan", "draws located, hit counts by bitmap rank", "drawn records: kets, probes, rows, links"]
dt = np.diff(t, axis=1) / 100.0  # wall_clock64: 100 MHz
print("mean per workgroup (us):")
for k, nm in enumerate(names):
    print(f"  {nm:55s} {dt[:, k].mean():8.2f}   (median {np.median(dt[:, k]):.2f})")
life = (t[:, 9] - t[:, 0]) / 100.0
span = (t[:, 9].max() - t[:, 0].min()) / 100.0
print(f"  workgroup life {life.mean():.1f} us; kernel span {span:.1f} us; workgroups in flight on average {life.sum() / span:.0f}")
ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t))], 1), np.stack([t[:, 9], -np.ones(len(t))], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
alive = np.cumsum(ev[:, 1])
dur = np.diff(ev[:, 0], append=ev[-1, 0])
order = np.argsort(alive)
cum = np.cumsum(dur[order]) / dur.sum()
med = alive[order][np.searchsorted(cum, 0.5)]
print(f"  workgroups alive at once: peak {int(alive.max())} ({alive.max() / 256:.2f} per CU), time-weighted median {int(med)} ({med / 256:.2f} per CU)")
