"""time of the fused SIMPLE + real-RBM kernel (bench.py's syn<sorb>_eloc_rbm / fe2s2_eloc_rbm workloads) -- for PYNQS_RBM_STOP ablation builds
(PYNQS_AMD_LIB=...) whose results are not local energies.  usage: python tools/eloc_rbm_time.py workload walkers [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
name, n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda")
w = B.make_workload(name, n, 0, dev)
tab = w.cx.RBMTable(w.W, w.hb, w.vb)
for _ in range(3):
    w.cx.eloc_rbm(w.x, w.h1, w.h2, tab, w.sorb, w.nele, w.noA, w.noB)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    w.cx.eloc_rbm(w.x, w.h1, w.h2, tab, w.sorb, w.nele, w.noA, w.noB)
b.record(); b.synchronize()
ms = a.elapsed_time(b) / reps
print(f"{name} x {n}: {ms:.4f} ms per call, {w.flops_per_walker * n / ms / 1e9:.2f} TFLOP/s of 3 ncomb H = {w.flops_per_walker * n / ms / 1e9 / 78.6:.3f} of 78.6 (lib {os.environ.get('PYNQS_AMD_LIB', 'default')})")
