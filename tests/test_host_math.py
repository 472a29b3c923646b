"""Host-side number theory behind the C ABI (no GPU needed): prime rule, primitive roots, Barrett ratios.
Pins the values SURVEY appendix A4 lists and checks that bad moduli are refused quickly instead of searched for."""
import ctypes as C
import time

import pytest

from oracle import cport as O


@pytest.fixture(scope="module")
def lib():
    from fhe_reliability_gpu_amd._lib import lib as L
    return L


def test_prime_rule_matches_reference_log(lib):
    # reliability_test/data/bits1-16_num1.txt:10 -- the six 50-bit primes Phantom logged for N = 16384
    want = [1125899903107073, 1125899903500289, 1125899903795201, 1125899903827969, 1125899903991809, 1125899904679937]
    q = (C.c_uint64 * 6)()
    bits = (C.c_int * 6)(*([50] * 6))
    assert lib.fhe_moduli_create(C.c_uint64(16384), bits, 6, q) == 0
    assert sorted(q) == want
    assert list(q) == [int(x) for x in O.gen_primes(16384, 50, 6)]


def test_primitive_root_known_answer_and_bounded_search(lib):
    out = C.c_uint64()
    assert lib.fhe_min_primitive_root(C.c_uint64(1125899903107073), C.c_uint64(2 * 16384), C.byref(out)) == 0
    assert out.value == 32853495844
    assert out.value == O.min_primitive_root(1125899903107073, 2 * 16384)
    # a composite modulus that is 1 mod 2N, an even one, and a prime without a subgroup of that order: refused at once
    t = time.perf_counter()
    for q in ((1 << 62) + 1, (1 << 40) + (1 << 20) + 1, 1 << 50, 97):
        assert lib.fhe_min_primitive_root(C.c_uint64(q), C.c_uint64(2048), C.byref(out)) != 0
    assert time.perf_counter() - t < 2.0
    assert lib.fhe_last_error()


def test_barrett_ratio(lib):
    q = 1125899903107073
    r = (C.c_uint64 * 3)()
    assert lib.fhe_modulus_const_ratio(C.c_uint64(q), r) == 0
    full = (1 << 128) // q
    assert (r[1] << 64) | r[0] == full and r[2] == (1 << 128) - full * q
