"""Per-workgroup phase timestamps of the LDS-row semi-stochastic front end (kernels_reduce_rowlds.hip built with -DPYNQS_ROWLDS_STAMPS):
  cd pynqs_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DPYNQS_ROWLDS_STAMPS -c kernels_reduce_rowlds.hip -o /tmp/s.o &&
  hipcc --offload-arch=gfx950 -fPIC -shared -o ../../build_ab/libpynqs_rl_stamps.so $(ls build/*.o | grep -v kernels_reduce_rowlds.o) /tmp/s.o
  PYNQS_AMD_LIB=$PWD/build_ab/libpynqs_rl_stamps.so python tools/rowlds_stamps.py"""
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
assert lib.pynqs_debug_rowlds_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
t = out[:, :8].astype(np.float64)
names = ["walker tables", "enumeration (incl. the wait for the slowest wave)", "kept columns: ranks, values into place", "S, segment sums, starting sums",
         "draws located", "hits marked, drawn records placed", "all records: kets, probes, rows, links"]
dt = np.diff(t, axis=1) / 100.0  # wall_clock64: 100 MHz
print("mean per workgroup (us):")
for k, nm in enumerate(names):
    print(f"  {nm:55s} {dt[:, k].mean():8.2f}   (median {np.median(dt[:, k]):.2f})")
life = (t[:, 7] - t[:, 0]) / 100.0
span = (t[:, 7].max() - t[:, 0].min()) / 100.0
print(f"  workgroup life {life.mean():.1f} us; kernel span {span:.1f} us; workgroups in flight on average {life.sum() / span:.0f}")
