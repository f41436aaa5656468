"""Drop-in `libs.C_extension` (PyNQS README.md:47 expects the compiled module under libs/).

Hot-path names come from pynqs_amd.C_extension.  Names of the reference module that are outside the
local-energy path are present so that `vmc/sample.py:19-26` and `utils/public_function.py:17`
import cleanly; the ones not implemented raise NotImplementedError when CALLED, never at import."""
from pynqs_amd.C_extension import (MAX_NELE, MAX_SORB, MAX_SORB_LEN, check_sorb, compress_h1e_h2e,  # noqa: F401
                                   decompress_h1e_h2e, get_comb_hij_fused, hash_build, hash_lookup, HashTable, get_comb_tensor, get_hij_torch,
                                   merge_rank_sample, onv_to_tensor, spin_flip_rand, tensor_to_onv, wavefunction_lut, permute_sgn, constrain_make_charts)


def _out_of_scope(name: str, where: str):
    def f(*a, **k):
        raise NotImplementedError(f"{name} ({where}) is outside the local-energy path implemented by pynqs_amd")
    f.__name__ = name
    return f


MCMC_sample = _out_of_scope("MCMC_sample", "dead code in the reference, vmc/sample.py:504")
convert_sites = _out_of_scope("convert_sites", "MPSWavefunction helper")
mps_vbatch = _out_of_scope("mps_vbatch", "MPSWavefunction helper")
wavefunction_lut_map = _out_of_scope("wavefunction_lut_map", "experimental unordered_map LUT")
