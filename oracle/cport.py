"""ctypes wrapper over oracle/fhe_oracle.c (TEST INFRASTRUCTURE ONLY).

NumPy uint64 arrays in, NumPy uint64 arrays out.  The shared object is built by
``make -C oracle`` (done by ``__graft_entry__.build()``); if it is missing it is
built on first use with gcc.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfhe_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libref_baseconv.so")

u64 = C.c_uint64
p64 = C.POINTER(C.c_uint64)
pi64 = C.POINTER(C.c_int64)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fhe_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        sig = {
            "orc_mulmod": (u64, [u64, u64, u64]),
            "orc_powmod": (u64, [u64, u64, u64]),
            "orc_invmod": (u64, [u64, u64]),
            "orc_is_prime": (C.c_int, [u64]),
            "orc_gen_primes": (C.c_int, [u64, C.c_int, C.c_int, p64]),
            "orc_min_primitive_root": (u64, [u64, u64]),
            "orc_root_powers": (None, [u64, C.c_int, u64, p64, p64]),
            "orc_ntt_cyclic": (None, [p64, u64, u64, u64]),
            "orc_intt_cyclic": (None, [p64, u64, u64, u64]),
            "orc_ntt_cyclic_nthroot": (None, [p64, u64, u64, u64]),
            "orc_intt_cyclic_nthroot": (None, [p64, u64, u64, u64]),
            "orc_negacyclic_ntt_natural": (None, [p64, u64, u64, u64]),
            "orc_negacyclic_intt_natural": (None, [p64, u64, u64, u64]),
            "orc_nwt_forward": (None, [p64, C.c_int, u64, p64]),
            "orc_nwt_inverse": (None, [p64, C.c_int, u64, p64]),
            "orc_nwt_forward_batch": (None, [p64, C.c_int, C.c_int, p64, p64]),
            "orc_nwt_inverse_batch": (None, [p64, C.c_int, C.c_int, p64, p64]),
            "orc_modmul_batch": (None, [p64, p64, p64, u64, C.c_int, p64, C.c_int]),
            "orc_set_threads": (C.c_int, [C.c_int]),
            "orc_modmul": (None, [p64, p64, p64, u64, u64]),
            "orc_modmul_acc": (None, [p64, p64, p64, u64, u64]),
            "orc_polymul_naive_negacyclic": (None, [p64, p64, p64, u64, u64]),
            "orc_polymul_negacyclic_ntt": (None, [p64, p64, p64, u64, u64, u64]),
            "orc_ntt_direct": (None, [p64, p64, u64, u64, u64]),
            "orc_four_step_ntt": (None, [p64, p64, u64, u64, u64, u64]),
            "orc_barrett_ctx": (None, [u64, C.POINTER(C.c_int), p64, p64]),
            "orc_barrett_reduce": (u64, [u64, u64, u64, C.c_int, u64, u64]),
            "orc_crt_garner": (None, [p64, p64, p64, p64, C.c_int, u64]),
            "orc_baseconv_exact": (None, [p64, p64, p64, C.c_int, p64, C.c_int, u64]),
            "orc_bconv_fast": (None, [p64, p64, p64, C.c_int, p64, C.c_int, u64]),
            "orc_bsgs_hadamard": (None, [pi64, pi64, pi64, C.c_int, C.c_int]),
            "orc_bsgs_hadamard_mod": (None, [p64, p64, p64, C.c_int, C.c_int, u64]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
        # OpenMP would start one worker per visible CPU; a GPU box shows 256 of them to a job that may use 16 (cgroup quota), and every
        # parallel region of the batch helpers then costs milliseconds of oversubscription (a rotate_ref at N = 32 took 2.8 s there):
        # default to the cores this process may really use
        if "OMP_NUM_THREADS" not in os.environ:
            L.orc_set_threads(_usable_cores())
    return _lib


def _usable_cores() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                n = min(n, max(1, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.5)))
        except Exception:
            pass
    return max(1, min(n, 32))


def _a(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def _p(a: np.ndarray):
    return a.ctypes.data_as(p64)


# ---------------------------------------------------------------- tables
def gen_primes(N: int, bits: int, count: int) -> list:
    out = np.zeros(count, dtype=np.uint64)
    got = lib().orc_gen_primes(N, bits, count, _p(out))
    if got != count:
        raise ValueError(f"only {got} primes of {bits} bits with p = 1 mod {2 * N}")
    return [int(x) for x in out]


def min_primitive_root(q: int, order: int) -> int:
    r = lib().orc_min_primitive_root(q, order)
    if r == 0:
        raise ValueError("no root of that order")
    return int(r)


def root_powers(q: int, logn: int, psi: int | None = None, shoup: bool = False):
    if psi is None:
        psi = min_primitive_root(q, 2 << logn)
    rp = np.zeros(1 << logn, dtype=np.uint64)
    sh = np.zeros(1 << logn, dtype=np.uint64)
    lib().orc_root_powers(q, logn, psi, _p(rp), _p(sh))
    return (rp, sh) if shoup else rp


# ------------------------------------------------------------ transforms
def ntt_cyclic(a, mod: int, root: int) -> np.ndarray:
    a = _a(a).copy()
    lib().orc_ntt_cyclic(_p(a), a.size, mod, root)
    return a


def intt_cyclic(a, mod: int, root: int) -> np.ndarray:
    a = _a(a).copy()
    lib().orc_intt_cyclic(_p(a), a.size, mod, root)
    return a


def ntt_nthroot(a, root: int, mod: int) -> np.ndarray:
    a = _a(a).copy()
    lib().orc_ntt_cyclic_nthroot(_p(a), a.size, root, mod)
    return a


def negacyclic_ntt(a, psi: int, mod: int) -> np.ndarray:
    a = _a(a).copy()
    lib().orc_negacyclic_ntt_natural(_p(a), a.size, psi, mod)
    return a


def negacyclic_intt(a, psi: int, mod: int) -> np.ndarray:
    a = _a(a).copy()
    lib().orc_negacyclic_intt_natural(_p(a), a.size, psi, mod)
    return a


def nwt_forward(a, q: int, rp) -> np.ndarray:
    a = _a(a).copy()
    rp = _a(rp)
    lib().orc_nwt_forward(_p(a), int(a.size).bit_length() - 1, q, _p(rp))
    return a


def nwt_inverse(a, q: int, rp) -> np.ndarray:
    a = _a(a).copy()
    rp = _a(rp)
    lib().orc_nwt_inverse(_p(a), int(a.size).bit_length() - 1, q, _p(rp))
    return a


def nwt_forward_batch(a, qs, rps) -> np.ndarray:
    """a: (L, N); qs: L moduli; rps: (L, N) tables."""
    a = _a(a).copy()
    qs = _a(qs)
    rps = _a(rps)
    L, N = a.shape
    lib().orc_nwt_forward_batch(_p(a), int(N).bit_length() - 1, L, _p(qs), _p(rps))
    return a


def nwt_inverse_batch(a, qs, rps) -> np.ndarray:
    a = _a(a).copy()
    qs = _a(qs)
    rps = _a(rps)
    L, N = a.shape
    lib().orc_nwt_inverse_batch(_p(a), int(N).bit_length() - 1, L, _p(qs), _p(rps))
    return a


def modmul_batch(a, b, qs, acc=None) -> np.ndarray:
    """(L, n) limb-wise products mod qs[l]; with ``acc`` (L, n) returns acc + a*b."""
    a, b, qs = _a(a), _a(b), _a(qs)
    c = np.zeros_like(a) if acc is None else _a(acc).copy()
    lib().orc_modmul_batch(_p(c), _p(a), _p(b), a.shape[1], a.shape[0], _p(qs), 0 if acc is None else 1)
    return c


def set_threads(n: int = 0) -> int:
    """Worker threads of the batch helpers (limb-parallel, OpenMP); 0 = query.  1 = the scalar port."""
    return int(lib().orc_set_threads(n))


def modmul(a, b, mod: int) -> np.ndarray:
    a, b = _a(a), _a(b)
    c = np.zeros_like(a)
    lib().orc_modmul(_p(c), _p(a), _p(b), a.size, mod)
    return c


def modmul_acc(c, a, b, mod: int) -> np.ndarray:
    c = _a(c).copy()
    a, b = _a(a), _a(b)
    lib().orc_modmul_acc(_p(c), _p(a), _p(b), a.size, mod)
    return c


def polymul_naive(a, b, mod: int) -> np.ndarray:
    a, b = _a(a), _a(b)
    r = np.zeros_like(a)
    lib().orc_polymul_naive_negacyclic(_p(r), _p(a), _p(b), a.size, mod)
    return r


def polymul_ntt(a, b, psi: int, mod: int) -> np.ndarray:
    a, b = _a(a), _a(b)
    r = np.zeros_like(a)
    lib().orc_polymul_negacyclic_ntt(_p(r), _p(a), _p(b), a.size, psi, mod)
    return r


def ntt_direct(a, mod: int, g: int) -> np.ndarray:
    a = _a(a)
    y = np.zeros_like(a)
    lib().orc_ntt_direct(_p(y), _p(a), a.size, mod, g)
    return y


def four_step_ntt(a, n1: int, n2: int, mod: int, g: int) -> np.ndarray:
    a = _a(a)
    assert a.size == n1 * n2
    y = np.zeros_like(a)
    lib().orc_four_step_ntt(_p(y), _p(a), n1, n2, mod, g)
    return y


def barrett_ctx(q: int):
    K = C.c_int()
    lo, hi = u64(), u64()
    lib().orc_barrett_ctx(q, C.byref(K), C.byref(lo), C.byref(hi))
    return K.value, (hi.value << 64) | lo.value


def barrett_reduce(t: int, q: int, K: int, mu: int) -> int:
    M = (1 << 64) - 1
    return int(lib().orc_barrett_reduce(t & M, t >> 64, q, K, mu & M, mu >> 64))


# -------------------------------------------------------- base conversion
def crt_garner(residues, moduli):
    r = _a(residues)
    m, N = r.shape
    mod = _a(moduli)
    lo = np.zeros(N, dtype=np.uint64)
    hi = np.zeros(N, dtype=np.uint64)
    lib().orc_crt_garner(_p(lo), _p(hi), _p(r), _p(mod), m, N)
    return lo, hi


def baseconv_exact(residues, mod_in, mod_out) -> np.ndarray:
    r = _a(residues)
    m, N = r.shape
    mi, mo = _a(mod_in), _a(mod_out)
    out = np.zeros((mo.size, N), dtype=np.uint64)
    lib().orc_baseconv_exact(_p(out), _p(r), _p(mi), m, _p(mo), mo.size, N)
    return out


def bconv_fast(residues, mod_in, mod_out) -> np.ndarray:
    r = _a(residues)
    m, N = r.shape
    mi, mo = _a(mod_in), _a(mod_out)
    out = np.zeros((N, mo.size), dtype=np.uint64)
    lib().orc_bconv_fast(_p(out), _p(r), _p(mi), m, _p(mo), mo.size, N)
    return out


def bsgs_hadamard(M_blocks, v) -> np.ndarray:
    M = np.ascontiguousarray(np.asarray(M_blocks, dtype=np.int64))
    v = np.ascontiguousarray(np.asarray(v, dtype=np.int64))
    k, bs = M.shape
    y = np.zeros(k * bs, dtype=np.int64)
    lib().orc_bsgs_hadamard(y.ctypes.data_as(pi64), M.ctypes.data_as(pi64), v.ctypes.data_as(pi64), k, bs)
    return y


def bsgs_hadamard_mod(M_blocks, v, q: int) -> np.ndarray:
    M = _a(M_blocks)
    v = _a(v)
    k, bs = M.shape
    y = np.zeros(k * bs, dtype=np.uint64)
    lib().orc_bsgs_hadamard_mod(_p(y), _p(M), _p(v), k, bs, q)
    return y


# --------------------------------------------- compiled reference (oracle/_ref)
def ref_crt_reconstruct(residues, moduli):
    """crt_reconstruct of the reference's rfhe_framewk/src/baseConv.cpp:146-173,
    compiled by ``make -C oracle ref``; returns None when oracle/_ref is absent."""
    if not os.path.exists(_REF_SO):
        return None
    L = C.CDLL(_REF_SO)
    L.ref_crt_reconstruct.restype = C.c_int
    L.ref_crt_reconstruct.argtypes = [p64, p64, C.c_int, C.c_int, p64, p64]
    r = _a(residues)
    m, N = r.shape
    mod = _a(moduli)
    lo = np.zeros(N, dtype=np.uint64)
    hi = np.zeros(N, dtype=np.uint64)
    L.ref_crt_reconstruct(_p(r), _p(mod), m, N, _p(lo), _p(hi))
    return lo, hi
