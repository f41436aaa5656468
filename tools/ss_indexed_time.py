"""INDEXED key-major SAMPLE_SPACE kernel against the streamed one: same results, time per call, index build time.
   python tools/ss_indexed_time.py [workload ...]   (default: the three sample-space workloads of bench.py)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench as B  # noqa: E402
from pynqs_amd import _native as N  # noqa: E402


def timed(fn, reps=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device("cuda:0")
    lib = N.lib()
    names = sys.argv[1:] or ["syn184_eloc_sample_space", "syn120_eloc_sample_space", "syn56_eloc_sample_space", "fe2s2_eloc_sample_space"]
    for name in names:
        w = B.make_workload(name, 8192, 0, dev)
        keys, wf = w.lut.bra_key, w.lut.wf_value
        nk = keys.size(0)
        st = torch.cuda.current_stream(dev).cuda_stream
        index = torch.empty(lib.pynqs_keys_index_bytes(nk, w.sorb), dtype=torch.uint8, device=dev)
        work = torch.empty(lib.pynqs_keys_index_workspace(nk, w.sorb), dtype=torch.uint8, device=dev)
        build = lambda: N.check(lib.pynqs_keys_index_build(keys.data_ptr(), nk, w.sorb, index.data_ptr(), work.data_ptr(), st), "index_build")
        t_build = timed(build, 10)
        out = {}
        for flip in (0, 1):
            e1, e2 = torch.empty_like(w.eloc), torch.empty_like(w.eloc)
            p1, p2 = torch.empty_like(w.psi0), torch.empty_like(w.psi0)
            args = (w.x.data_ptr(), w.n, w.sorb, w.nele, w.noA, w.noB, w.plan.data_ptr(), keys.data_ptr(), nk)
            if flip:  # psi0 is an input of the partner sum
                p1.copy_(out["p"]); p2.copy_(out["p"])
            streamed = lambda: N.check(lib.pynqs_eloc_sample_space_keys(*args, wf.data_ptr(), 1, flip, e1.data_ptr(), p1.data_ptr(), st), "keys")
            indexed = lambda: N.check(lib.pynqs_eloc_sample_space_indexed(*args, index.data_ptr(), wf.data_ptr(), 1, flip, e2.data_ptr(), p2.data_ptr(), st), "indexed")
            ts, ti = timed(streamed), timed(indexed)
            d = (e1 - e2).abs().max().item()
            again = e2.clone(); indexed(); torch.cuda.synchronize()
            out["p"] = p1.clone()
            print(f"{name} flip={flip}: {w.n} walkers, {nk} keys: streamed {ts:.4f} ms, indexed {ti:.4f} ms (index build {t_build:.3f} ms); "
                  f"max |diff| {d:.3e}, psi0 equal {torch.equal(p1, p2)}, indexed bit-reproducible {torch.equal(again, e2)}", flush=True)


if __name__ == "__main__":
    main()
