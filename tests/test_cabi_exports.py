"""The C-ABI library loads and exports every function include/fhe_mi355x.h declares
(no compute calls: runs without a GPU), and the host-side helpers that need no device
agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "fhe_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fhe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from fhe_reliability_gpu_amd import _lib
    names = _declared()
    assert len(names) >= 35
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in fhe_mi355x.h but not exported"
    # the ctypes binding covers the same set
    assert sorted(_lib.EXPORTS) == names
    assert _lib.lib.fhe_version() >= 100


def test_host_side_tables_match_oracle():
    import fhe_reliability_gpu_amd as F
    from oracle import cport as O
    for N, bits, cnt in ((16384, 50, 6), (4096, 61, 3), (65536, 50, 4), (2, 30, 2)):
        assert F.create_moduli(N, [bits] * cnt) == O.gen_primes(N, bits, cnt)
    # mixed bit sizes: every size gets its own descending-search pool, handed out smallest first
    got = F.create_moduli(8192, [50, 40, 50, 40, 30])
    p50, p40, p30 = O.gen_primes(8192, 50, 2), O.gen_primes(8192, 40, 2), O.gen_primes(8192, 30, 1)
    assert got == [p50[0], p40[0], p50[1], p40[1], p30[0]]
    q = 1125899903107073
    assert F.min_primitive_root(q, 32768) == 32853495844 == O.min_primitive_root(q, 32768)
    rp, sh = F.root_powers(q, 14, shoup=True)
    orp, osh = O.root_powers(q, 14, shoup=True)
    assert (rp == orp).all() and (sh == osh).all()
    assert int(rp[1]) == 163343304402113 and int(sh[1]) == 2676216708203466951      # SURVEY KAT-2


def test_const_ratio_is_floor_2_128_over_q():
    from fhe_reliability_gpu_amd._lib import lib
    for q in (17, 1125899903107073, 2305843009211596801, (1 << 61) - 1):
        out = (C.c_uint64 * 3)()
        assert lib.fhe_modulus_const_ratio(q, out) == 0
        assert (int(out[1]) << 64) | int(out[0]) == (1 << 128) // q
        assert int(out[2]) == (1 << 128) % q


def test_errors_do_not_cross_the_abi_as_exceptions():
    from fhe_reliability_gpu_amd._lib import lib
    out = (C.c_uint64 * 1)()
    bits = (C.c_int * 1)(70)
    assert lib.fhe_moduli_create(4096, bits, 1, out) != 0          # unsupported size -> status, message
    assert b"bit sizes" in lib.fhe_last_error()
    assert lib.fhe_min_primitive_root(17, 64, out) != 0             # 64 does not divide 16
