"""pynqs_amd -- MI355X-native (gfx950) determinant / local-energy engine for PyNQS.

Scope: the VMC local-energy hot path only: excitation enumeration, Slater-Condon
matrix elements, wavefunction look-up and the local-energy contraction, as hand-written HIP kernels
behind the reference's `libs.C_extension` API (pynqs_amd.C_extension) and its `vmc/energy` functions
(pynqs_amd.energy).  There is no CPU fallback.

Modules (reference counterpart):
  C_extension      libs/C_extension (cpp_src/tensor/bind.cpp) + fused entry points (RBMTable, eloc_rbm, hash table)
  energy           vmc/energy/{eloc,flip,etot}.py: local_energy, total_energy, Func, REDUCE compaction and draws
  public_function  utils/public_function.py: WavefunctionLUT, ansatz_batch, sorting / unique of determinants
  stats, grad      utils/stats/{dist_stats,mc_stats}.py, vmc/grad/energy_grad.py
  distributed      utils/distributed/comm.py: packed all-reduce, all-gather of uneven shards, shard bounds
  sample_comm      Sampler.gather_scatter_sample (vmc/sample.py:627-772)
  gfmc             gfmc/walker.py: Green's-function row, move (one kernel), branching
  rbm              the real RBM amplitude of vmc/ansatz/rbm/rbm.py (used by tests, bench and the example)
"""
__version__ = "0.1.0"
