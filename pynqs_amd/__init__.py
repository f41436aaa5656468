"""pynqs_amd -- MI355X-native (gfx950) determinant / local-energy engine for PyNQS.

Scope: the VMC local-energy hot path only: excitation enumeration, Slater-Condon
matrix elements, wavefunction look-up and the local-energy contraction, as hand-written HIP kernels
behind the reference's `libs.C_extension` API (pynqs_amd.C_extension) and its `vmc/energy` functions
(pynqs_amd.energy).  There is no CPU fallback.
"""
__version__ = "0.1.0"
