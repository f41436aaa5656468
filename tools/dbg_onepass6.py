import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import rand_occ, synth_integrals
from pynqs_amd import energy as E, C_extension as cx, reduce_front as RF
sorb, no, n, eps = 120, 30, 3, 0.495
h1, h2 = synth_integrals(sorb)
h1e, h2e = torch.from_numpy(h1).cuda(), torch.from_numpy(h2).cuda()
x = cx.tensor_to_onv(torch.from_numpy(rand_occ(n, sorb, no, no, seed=sorb)).cuda(), sorb)
plan = cx.plan_for(h1e, h2e, sorb, x.device).buf
fe = RF.ReduceFrontEnd(n, sorb, 2 * no, no, no, 0, torch.float64, x.device, 64, 49152, torch.float32)
fe.run(x, plan, eps, 1, None)
nu = fe.counters_host()[0]
src = fe.uniq_onv[:nu]
torch.cuda.synchronize(); time.sleep(0.05)
outs = [cx.onv_to_tensor(src, sorb) for _ in range(4)]
torch.cuda.synchronize()
bits = torch.from_numpy(np.unpackbits(src.cpu().numpy(), axis=1, bitorder="little")[:, :sorb].astype(np.float32) * 2 - 1)
res = []
for o in outs:
    w = (o.cpu() != bits)
    res.append(int(w.sum()))
    if int(w.sum()):
        flat = torch.nonzero(w.reshape(-1)).squeeze(1)
        vals = o.cpu().reshape(-1)[flat]
        res.append(("mod4", sorted(set((flat % 4).tolist())), "all -1" if bool((vals == -1).all()) else "mixed", "span", int(flat.min()), int(flat.max())))
print("calls:", res)
