"""ctypes binding of libpynqs_amd.so (the C ABI declared in include/pynqs_amd.h).

There is no CPU fallback: if the HIP library is missing or no GPU is present, every compute entry
point raises.  Building: ``python -c "import __graft_entry__ as g; g.build()"`` or
``pynqs_amd.build.build_native()``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYNQS_AMD_LIB selects another build of the same library (e.g. a profiling variant); never a fallback
LIB_PATH = os.environ.get("PYNQS_AMD_LIB") or os.path.join(_HERE, "csrc", "libpynqs_amd.so")

PYNQS_F32, PYNQS_F64 = 0, 1
RBM_REAL, RBM_TANH, RBM_PHASE, RBM_COMPLEX = 0, 1, 2, 4
OK, EINVAL, ELAUNCH, ELENGTH, EOVERFLOW = 0, -1, -2, -3, -4

_i64, _int, _vp, _dbl = C.c_int64, C.c_int, C.c_void_p, C.c_double

# name -> (restype, argtypes); mirrors include/pynqs_amd.h one to one
SIGNATURES = {
    "pynqs_abi_version": (_int, []),
    "pynqs_last_error": (C.c_char_p, []),
    "pynqs_num_sd": (_i64, [_int, _int, _int]),
    "pynqs_check_sorb": (_int, [_int, _int]),
    "pynqs_comb_hij_fused": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, _int, _vp, _vp, _vp]),
    "pynqs_comb": (_int, [_vp, _i64, _int, _int, _int, _vp, _vp, _vp]),
    "pynqs_hij": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _vp, _int, _int, _int, _vp, _vp]),
    "pynqs_onv_to_pm1": (_int, [_vp, _i64, _int, _int, _vp, _vp]),
    "pynqs_pm01_to_onv": (_int, [_vp, _i64, _int, _vp, _vp]),
    "pynqs_wavefunction_lut": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _vp, _vp]),
    "pynqs_spin_flip_rand": (_int, [_vp, _i64, _int, _int, _int, C.c_uint64, C.c_uint64, _vp, _vp]),
    "pynqs_plan_bytes": (_i64, [_int, _int]),
    "pynqs_plan_build": (_int, [_vp, _vp, _int, _int, _vp, _vp]),
    "pynqs_comb_hij_fused_plan": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _int, _vp, _vp, _vp]),
    "pynqs_eloc_sample_space": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, _i64, _vp, _int, _vp, _vp, _vp]),
    "pynqs_hash_bytes": (_i64, [_i64, _int]),
    "pynqs_hash_build": (_int, [_vp, _i64, _int, _vp, _vp]),
    "pynqs_hash_lookup": (_int, [_vp, _i64, _vp, _i64, _int, _vp, _vp, _vp]),
    "pynqs_unique_workspace": (_i64, [_i64]),
    "pynqs_unique_first": (_int, [_vp, _i64, _int, _vp, _vp, _vp]),
    "pynqs_eloc_sample_space_hash": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, _i64, _vp, _int, _vp, _vp, _vp]),
    "pynqs_eloc_sample_space_flip": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, _i64, _vp, _int, _vp, _vp, _vp]),
    "pynqs_eloc_sample_space_hash_flip": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, _i64, _vp, _int, _vp, _vp, _vp]),
    "pynqs_eloc_sample_space_keys": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, _i64, _vp, _int, _int, _vp, _vp, _vp]),
    "pynqs_keys_index_bytes": (_i64, [_i64, _int]),
    "pynqs_keys_index_workspace": (_i64, [_i64, _int]),
    "pynqs_keys_index_build": (_int, [_vp, _i64, _int, _vp, _vp, _vp]),
    "pynqs_keys_index_density": (_int, [_vp, _i64, _int, _vp, _vp]),
    "pynqs_eloc_sample_space_indexed": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, _i64, _vp, _vp, _int, _int, _vp, _vp, _vp]),
    "pynqs_eloc_rbm_supported": (_int, [_int, _int, _int, _int, _int]),
    "pynqs_rbm_table_bytes": (_i64, [_int, _int]),
    "pynqs_rbm_table_build": (_int, [_vp, _vp, _vp, _int, _int, _vp, _vp]),
    "pynqs_eloc_rbm": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, _int, _vp, _vp, _vp]),
    "pynqs_eloc_rbm_flavour": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, _int, _int, _vp, _vp, _vp]),
    "pynqs_eloc_crbm_supported": (_int, [_int, _int, _int, _int, _int]),
    "pynqs_crbm_table_bytes": (_i64, [_int, _int]),
    "pynqs_crbm_table_build": (_int, [_vp, _vp, _vp, _int, _int, _vp, _vp]),
    "pynqs_eloc_crbm": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, _int, _dbl, _vp, _vp, _vp]),
    "pynqs_gfmc_sample": (_int, [_vp, _i64, _i64, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    "pynqs_green_rbm": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, _int, _int, _dbl, _vp, _vp, _vp, _vp, _vp]),
    "pynqs_gfmc_sample_rank": (_int, [_vp, _i64, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _vp, _vp]),
    "pynqs_moments_workspace": (_i64, []),
    "pynqs_stats_finish": (_int, [_vp, _dbl, _dbl, _vp, _vp]),
    "pynqs_weighted_moments": (_int, [_vp, _int, _vp, _i64, _vp, _vp]),
    "pynqs_reduce_tiles": (_i64, [_i64, _int, _int, _int, _int]),
    "pynqs_reduce_count": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _int, _dbl, _vp, _vp]),
    "pynqs_reduce_count_sums": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _int, _dbl, _vp, _vp, _vp]),
    "pynqs_reduce_sample": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _int, _dbl, _vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp, _vp]),
    "pynqs_reduce_emit": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _int, _dbl, _vp, _vp, _vp, _vp, _vp]),
}



class ReduceIO(C.Structure):
    """include/pynqs_amd.h: pynqs_reduce_io (buffers and capacities of the one-launch REDUCE front end)."""
    _fields_ = [("cap_doubles", _i64), ("cap_unique", _i64), ("dedup_slots", _i64),
                ("rec_col", _vp), ("rec_w", _vp), ("rec_onv", _vp), ("rec_link", _vp), ("seg_count", _vp),
                ("srec_col", _vp), ("srec_w", _vp), ("srec_onv", _vp), ("srec_link", _vp), ("row_sum", _vp),
                ("dedup_table", _vp), ("uniq_onv", _vp), ("uniq_pm1", _vp), ("pm1_dtype", C.c_int32), ("lut_is_hash", C.c_int32),
                ("lut_table", _vp), ("lut_nkeys", _i64), ("counters", _vp), ("seed_dev", _vp), ("row_cache", _vp), ("uniq_parent", _vp),
                ("tile_scratch", _vp), ("tile_scratch_bytes", _i64), ("row_f32", _vp)]


SIGNATURES.update({
    "pynqs_reduce_onepass_geometry": (_int, [_i64, _int, _int, _int, _int, _int, C.POINTER(_i64)]),
    "pynqs_reduce_onepass_tile_scratch_bytes": (_i64, [_i64, _int, _int, _int, _int, _int]),
    "pynqs_reduce_onepass_list_capacity": (_int, [_i64, _int, _int, _int, _int, _int, _int, _int, _int, C.POINTER(_i64)]),
    "pynqs_reduce_onepass_wants_row_f32": (_int, [_i64, _int, _int, _int, _int, _int, _int, _i64, _int, _int]),
    "pynqs_reduce_onepass_row_f32_elements": (_i64, [_i64, _int, _int, _int, _int]),
    "pynqs_reduce_onepass": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _int, _dbl, _int, C.c_uint64, C.POINTER(ReduceIO), _vp]),
    "pynqs_rbm_forward": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _int, _int, _vp, _vp]),
    "pynqs_rbm_children_table_bytes": (_i64, [_i64, _int, _int, _int]),
    "pynqs_rbm_children_prepare": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _int, _int, _vp, _vp]),
    "pynqs_rbm_forward_children": (_int, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _int, _vp, _vp, _vp, _int, _int, _vp, _vp]),
    "pynqs_rbm_forward_children_supported": (_int, [_int, _int, _int]),
    "pynqs_rbm_grad_workspace": (_i64, [_i64, _int, _int, _int]),
    "pynqs_rbm_grad": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _int, _int, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pynqs_reduce_contract": (_int, [_i64, _int, _int, _int, _int, _int, _int, C.POINTER(ReduceIO), _vp, _vp, _int, _int, _vp, _vp, _vp]),
})

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libpynqs_amd.so; raises NativeLibraryError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "pynqs_amd has no CPU fallback."
            )
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 missing
            raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            f = getattr(l, name)  # AttributeError -> header/library mismatch, loud
            f.restype = res
            f.argtypes = args
        if l.pynqs_abi_version() != 1:
            raise NativeLibraryError("libpynqs_amd.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int, what: str) -> None:
    if rc == OK:
        return
    msg = lib().pynqs_last_error().decode(errors="replace")
    if rc == ELENGTH:
        raise ValueError(msg)  # std::length_error -> ValueError in the reference (bind.cpp:290)
    if rc == EOVERFLOW:
        raise OverflowError(msg)  # std::overflow_error -> OverflowError (bind.cpp:294-299)
    raise RuntimeError(f"{what}: {msg} (code {rc})")
