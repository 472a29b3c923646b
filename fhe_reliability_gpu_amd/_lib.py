"""ctypes binding of libfhe_mi355x.so (include/fhe_mi355x.h).

The shared object is built in-tree by ``make -C fhe_reliability_gpu_amd/csrc`` (or
``__graft_entry__.build()``).  If it is missing the import FAILS LOUDLY: there is
no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfhe_mi355x.so")

u64 = C.c_uint64
p64 = C.POINTER(C.c_uint64)
vp = C.c_void_p
sz = C.c_size_t
ci = C.c_int


class FheError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(make -C fhe_reliability_gpu_amd/csrc, or python -c 'import __graft_entry__ as g; g.build()'). "
            "This package has no CPU fallback."
        )
    # When torch is already imported its bundled libamdhip64.so.7 is resident and is reused
    # (same SONAME); otherwise /opt/rocm's runtime is loaded through the library's RUNPATH.
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

_SIG = {
    "fhe_version": (ci, []),
    "fhe_last_error": (C.c_char_p, []),
    "fhe_ctx_create": (ci, [ci, C.POINTER(vp)]),
    "fhe_ctx_destroy": (ci, [vp]),
    "fhe_ctx_stream": (ci, [vp, C.POINTER(vp)]),
    "fhe_ctx_set_option": (ci, [vp, C.c_char_p, C.c_long]),
    "fhe_ctx_check": (ci, [vp]),
    "fhe_sync": (ci, [vp, vp]),
    "fhe_alloc": (ci, [vp, sz, C.POINTER(vp)]),
    "fhe_free": (ci, [vp, vp]),
    "fhe_h2d": (ci, [vp, vp, vp, sz, vp]),
    "fhe_d2h": (ci, [vp, vp, vp, sz, vp]),
    "fhe_d2d": (ci, [vp, vp, vp, sz, vp]),
    "fhe_memset": (ci, [vp, vp, ci, sz, vp]),
    "fhe_moduli_create": (ci, [u64, C.POINTER(ci), ci, p64]),
    "fhe_modulus_const_ratio": (ci, [u64, p64]),
    "fhe_min_primitive_root": (ci, [u64, u64, p64]),
    "fhe_root_powers": (ci, [u64, ci, p64, p64]),
    "fhe_ntt_tables_create": (ci, [vp, ci, p64, ci, C.POINTER(vp)]),
    "fhe_ntt_tables_create_from_roots": (ci, [vp, ci, p64, ci, p64, ci, C.POINTER(vp)]),
    "fhe_ntt_tables_destroy": (ci, [vp]),
    "fhe_ntt_tables_info": (ci, [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), p64]),
    "fhe_ntt_forward_inplace": (ci, [vp, vp, vp, sz, sz, vp]),
    "fhe_ntt_inverse_inplace": (ci, [vp, vp, vp, sz, sz, vp]),
    "fhe_ntt_forward_batch": (ci, [vp, vp, vp, sz, sz, sz, vp]),
    "fhe_ntt_inverse_batch": (ci, [vp, vp, vp, sz, sz, sz, vp]),
    "fhe_bitrev_permute": (ci, [vp, vp, vp, ci, sz, vp]),
    "fhe_ntt_cyclic": (ci, [vp, vp, vp, ci, sz, u64, u64, ci, ci, vp]),
    "fhe_fourstep_create": (ci, [vp, u64, u64, u64, u64, C.POINTER(vp)]),
    "fhe_fourstep_destroy": (ci, [vp]),
    "fhe_fourstep_ntt": (ci, [vp, vp, vp, vp, vp]),
    "fhe_fourstep_ntt_batch": (ci, [vp, vp, vp, vp, sz, vp]),
    "fhe_modmul": (ci, [vp, vp, vp, vp, vp, sz, sz, sz, vp]),
    "fhe_modmul_acc": (ci, [vp, vp, vp, vp, vp, sz, sz, sz, vp]),
    "fhe_polymul": (ci, [vp, vp, vp, vp, vp, sz, sz, sz, vp]),
    "fhe_baseconv_create": (ci, [vp, p64, ci, p64, ci, C.POINTER(vp)]),
    "fhe_baseconv_destroy": (ci, [vp]),
    "fhe_baseconv_exact": (ci, [vp, vp, vp, vp, sz, vp]),
    "fhe_baseconv_fast": (ci, [vp, vp, vp, vp, sz, vp]),
    "fhe_crt_garner": (ci, [vp, vp, vp, vp, p64, ci, sz, vp]),
    "fhe_bsgs_hadamard": (ci, [vp, vp, vp, vp, ci, ci, u64, vp]),
    "fhe_flip_bit": (ci, [vp, vp, u64, ci, vp]),
    "fhe_abft_create": (ci, [vp, vp, C.POINTER(vp)]),
    "fhe_abft_destroy": (ci, [vp]),
    "fhe_abft_checksum": (ci, [vp, vp, ci, vp, vp, sz, sz, sz, vp]),
    "fhe_ntt_forward_checked": (ci, [vp, vp, vp, vp, sz, sz, sz, vp, vp]),
    "fhe_ctx_inject_fault": (ci, [vp, C.c_longlong, ci]),
    "fhe_ntt_forward_checked_phases": (ci, [vp, vp, vp, vp, sz, sz, sz, vp, vp]),
    "fhe_ctx_inject_fault_in_pass": (ci, [vp, ci, C.c_uint32, C.c_uint32, ci]),
    "fhe_automorphism": (ci, [vp, vp, vp, vp, C.c_uint32, sz, sz, sz, vp]),
    "fhe_automorphism_ntt": (ci, [vp, vp, vp, ci, C.c_uint32, sz, vp]),
    "fhe_keyswitch_create": (ci, [vp, vp, ci, ci, ci, C.POINTER(vp)]),
    "fhe_keyswitch_destroy": (ci, [vp]),
    "fhe_keyswitch_shard_layout": (ci, [ci, ci, ci, ci, C.POINTER(ci)]),
    "fhe_keyswitch_create_sharded": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, C.POINTER(vp)]),
    "fhe_rescale_shard_info": (ci, [vp, C.POINTER(ci), C.POINTER(ci)]),
    "fhe_rescale_shard_begin": (ci, [vp, vp, vp, sz, vp]),
    "fhe_rescale_shard_finish": (ci, [vp, vp, vp, vp, sz, vp]),
    "fhe_keyswitch_shard_begin": (ci, [vp, vp, vp, vp]),
    "fhe_keyswitch_shard_inner": (ci, [vp, vp, vp, vp, vp]),
    "fhe_keyswitch_shard_finish": (ci, [vp, vp, vp, vp, vp, vp, vp]),
    "fhe_keyswitch_apply": (ci, [vp, vp, vp, vp, vp, vp, vp]),
    "fhe_rotate": (ci, [vp, vp, vp, vp, vp, vp, C.c_uint32, vp, vp]),
    "fhe_galois_key_prepare": (ci, [vp, vp, vp, vp, C.c_uint32, vp]),
    "fhe_rotate_hoisted": (ci, [vp, vp, C.POINTER(vp), C.POINTER(vp), vp, vp, C.POINTER(C.c_uint32), C.POINTER(vp), sz, vp]),
    "fhe_rotate_hoisted_shard_begin": (ci, [vp, vp, vp, vp]),
    "fhe_rotate_hoisted_shard_extend": (ci, [vp, vp, vp]),
    "fhe_rotate_hoisted_shard_inner": (ci, [vp, vp, vp, vp, C.c_uint32, vp]),
    "fhe_rotate_hoisted_shard_finish": (ci, [vp, vp, vp, vp, vp, C.c_uint32, vp]),
    "fhe_bsgs_matvec": (ci, [vp, vp, vp, vp, vp, vp, vp, sz, sz, C.POINTER(C.c_uint32), C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(vp), vp]),
    "fhe_rotate_shard_begin": (ci, [vp, vp, vp, C.c_uint32, vp]),
    "fhe_rotate_shard_inner": (ci, [vp, vp, vp, vp]),
    "fhe_rotate_shard_finish": (ci, [vp, vp, vp, vp, vp, C.c_uint32, vp]),
    "fhe_tensor_product": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, sz, vp]),
    "fhe_relinearize": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "fhe_rescale": (ci, [vp, vp, vp, vp, sz, vp]),
    "fhe_hmult": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, vp]),
    "fhe_hmult_shard_fusable": (ci, [vp, vp]),
    "fhe_hmult_shard_finish_begin": (ci, [vp, vp, vp, vp, vp]),
    "fhe_hmult_shard_finish_end": (ci, [vp, vp, vp, vp, vp, vp, vp]),
    "fhe_modadd": (ci, [vp, vp, vp, vp, vp, sz, sz, sz, vp]),
    "fhe_modsub": (ci, [vp, vp, vp, vp, vp, sz, sz, sz, vp]),
    "fhe_scalar_affine": (ci, [vp, vp, vp, p64, p64, vp, sz, sz, sz, vp]),
    "fhe_keyswitch_set_plain_modulus": (ci, [vp, u64]),
    "fhe_ctx_trace": (ci, [vp, ci]),
    "fhe_ctx_trace_read": (ci, [vp, C.c_char_p, sz, C.POINTER(sz)]),
}

EXPORTS = tuple(_SIG)

for _name, (_res, _args) in _SIG.items():
    _f = getattr(lib, _name)  # AttributeError here = the .so does not export a declared symbol
    _f.restype, _f.argtypes = _res, _args


def check(rc: int) -> None:
    if rc != 0:
        raise FheError(f"[fhe rc={rc}] {lib.fhe_last_error().decode(errors='replace')}")
