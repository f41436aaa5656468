"""cProfile of the host side of one REDUCE total_energy call at BASELINE configs[1]'s size (bench.py's syn56_reduce_vmc_step): where the time
between the kernels goes.  usage: python tools/profile_host_reduce.py [sorb no walkers eps]"""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
sorb, no, nw, eps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (56, 7, 4096, 0.47)
dev = torch.device("cuda")
w = B.ReduceTotalEnergy(f"syn{sorb}", sorb, no, nw, dev, eps)
torch.set_default_dtype(torch.float64)
for _ in range(5):
    w._call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    w._call()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per total_energy call ({w.n} walkers)")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    w._call()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
