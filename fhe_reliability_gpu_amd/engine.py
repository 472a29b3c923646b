"""Python host over the C ABI (``_lib``): device buffers, table sets and the
reference's Python-level operator names, so that code written against
``motivation/{ntt,baseConv,bsgs}.py``, ``rfhe_framewk/src/{ntt,negaclic_ntt,baseConv}.py``
and ``reliability_test/four_step_ntt_prot.py`` can switch to the GPU engine by
changing an import.  Every function here ends in HIP kernel launches; nothing is
computed on the CPU apart from packing/unpacking lists.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import check, lib, p64, u64, vp

_U64 = np.uint64


def _arr(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=_U64))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(p64)


class Engine:
    """One HIP device + stream (``phantom::util::cuda_stream_wrapper``, ntt_test.cu:40-41)."""

    def __init__(self, device: int = 0):
        h = vp()
        check(lib.fhe_ctx_create(device, C.byref(h)))
        self._h = h
        self.device = device
        s = vp()
        check(lib.fhe_ctx_stream(h, C.byref(s)))
        self.stream = s

    def set_option(self, name: str, value: int):
        """``ntt_mode`` (0 two launches / 1 fused), ``fused_dist``, ``fused_wgs`` -- tuning only."""
        check(lib.fhe_ctx_set_option(self._h, name.encode(), int(value)))

    def trace(self, enable: bool = True):
        """Start (and clear) / stop the operation trace (profile_framewk trace-line format)."""
        check(lib.fhe_ctx_trace(self._h, 1 if enable else 0))

    def trace_text(self) -> str:
        n = C.c_size_t()
        check(lib.fhe_ctx_trace_read(self._h, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value + 1)
        check(lib.fhe_ctx_trace_read(self._h, buf, n.value + 1, None))
        return buf.value.decode("utf-8")

    def check(self):
        """Synchronise and raise if a fused-NTT launch reported a timed-out wait."""
        check(lib.fhe_ctx_check(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib.fhe_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory ------------------------------------------------------------
    def alloc(self, n_words: int) -> "DeviceArray":
        return DeviceArray(self, n_words)

    def upload(self, host) -> "DeviceArray":
        a = _arr(host)
        d = DeviceArray(self, a.size)
        d.shape = a.shape
        check(lib.fhe_h2d(self._h, d.ptr, a.ctypes.data, a.nbytes, None))
        self.sync()  # the NumPy temporary must outlive the async copy
        return d

    def sync(self, stream=None):
        check(lib.fhe_sync(self._h, stream))

    # -- tables ------------------------------------------------------------
    def tables(self, log_n: int, moduli: Sequence[int]) -> "NttTables":
        return NttTables(self, log_n, moduli)

    def tables_from_roots(self, log_n: int, moduli: Sequence[int], root_powers, force_path: int = -1) -> "NttTables":
        return NttTables(self, log_n, moduli, root_powers=root_powers, force_path=force_path)


class DeviceArray:
    """uint64 words in HBM, owned through fhe_alloc/fhe_free (make_cuda_auto_ptr, ntt_test.cu:88)."""

    def __init__(self, eng: Engine, n_words: int):
        self.eng = eng
        self.size = int(n_words)
        self.shape = (self.size,)
        p = vp()
        check(lib.fhe_alloc(eng._h, self.size * 8, C.byref(p)))
        self.ptr = p

    def download(self) -> np.ndarray:
        out = np.empty(self.size, dtype=_U64)
        check(lib.fhe_d2h(self.eng._h, out.ctypes.data, self.ptr, out.nbytes, None))
        self.eng.sync()
        return out.reshape(self.shape)

    def copy_from(self, other: "DeviceArray"):
        check(lib.fhe_d2d(self.eng._h, self.ptr, other.ptr, min(self.size, other.size) * 8, None))

    def free(self):
        if self.ptr:
            lib.fhe_free(self.eng._h, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            if self.eng._h:
                self.free()
        except Exception:
            pass


def create_moduli(N: int, bits: Sequence[int]) -> List[int]:
    """``CoeffModulus::Create(N, {bits...})`` (reliability_test/ntt_test.cu:44)."""
    b = (C.c_int * len(bits))(*bits)
    out = (u64 * len(bits))()
    check(lib.fhe_moduli_create(N, b, len(bits), out))
    return [int(x) for x in out]


def min_primitive_root(q: int, order: int) -> int:
    r = u64()
    check(lib.fhe_min_primitive_root(q, order, C.byref(r)))
    return int(r.value)


def root_powers(q: int, log_n: int, shoup: bool = False):
    """``NTT::get_from_root_powers[_shoup]`` (ntt_test.cu:60-64)."""
    rp = np.zeros(1 << log_n, dtype=_U64)
    sh = np.zeros(1 << log_n, dtype=_U64)
    check(lib.fhe_root_powers(q, log_n, _ptr(rp), _ptr(sh)))
    return (rp, sh) if shoup else rp


class NttTables:
    """``DModulus[]`` + ``DNTTTable`` of a limb set (ntt_test.cu:47-69)."""

    def __init__(self, eng: Engine, log_n: int, moduli: Sequence[int], root_powers=None, force_path: int = -1):
        self.eng = eng
        self.log_n = log_n
        self.N = 1 << log_n
        self.moduli = [int(q) for q in moduli]
        q = _arr(self.moduli)
        h = vp()
        if root_powers is None:
            check(lib.fhe_ntt_tables_create(eng._h, log_n, _ptr(q), q.size, C.byref(h)))
        else:
            rp = _arr(root_powers).reshape(q.size, self.N)
            check(lib.fhe_ntt_tables_create_from_roots(eng._h, log_n, _ptr(q), q.size, _ptr(rp), force_path, C.byref(h)))
        self._h = h
        paths = (C.c_int * q.size)()
        psi = (u64 * q.size)()
        check(lib.fhe_ntt_tables_info(h, None, None, paths, psi))
        self.paths = list(paths)
        self.psi = [int(x) for x in psi]

    def __len__(self):
        return len(self.moduli)

    # nwt_2d_radix8_forward_inplace (ntt_test.cu:95): d is [n_poly][limbs][N]
    def forward(self, d: DeviceArray, limbs: Optional[int] = None, start: int = 0, n_poly: int = 1, stream=None):
        limbs = len(self) - start if limbs is None else limbs
        check(lib.fhe_ntt_forward_batch(self.eng._h, d.ptr, self._h, n_poly, limbs, start, stream))

    def inverse(self, d: DeviceArray, limbs: Optional[int] = None, start: int = 0, n_poly: int = 1, stream=None):
        limbs = len(self) - start if limbs is None else limbs
        check(lib.fhe_ntt_inverse_batch(self.eng._h, d.ptr, self._h, n_poly, limbs, start, stream))

    def modmul(self, c: DeviceArray, a: DeviceArray, b: DeviceArray, limbs=None, start=0, n_poly=1, acc=False, stream=None):
        limbs = len(self) - start if limbs is None else limbs
        f = lib.fhe_modmul_acc if acc else lib.fhe_modmul
        check(f(self.eng._h, c.ptr, a.ptr, b.ptr, self._h, n_poly, limbs, start, stream))

    def polymul(self, c: DeviceArray, a: DeviceArray, b: DeviceArray, limbs=None, start=0, n_poly=1, stream=None):
        limbs = len(self) - start if limbs is None else limbs
        check(lib.fhe_polymul(self.eng._h, c.ptr, a.ptr, b.ptr, self._h, n_poly, limbs, start, stream))

    def close(self):
        if getattr(self, "_h", None) and self.eng._h:
            lib.fhe_ntt_tables_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------
# Reference-named operators (lists in, lists out)
# ---------------------------------------------------------------------------
_default_engine: Optional[Engine] = None


def default_engine() -> Engine:
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


def _log2(n: int) -> int:
    if n < 2 or n & (n - 1):
        raise ValueError("length must be a power of two >= 2")
    return n.bit_length() - 1


def _cyclic(a, mod, root, convention, inverse, eng):
    eng = eng or default_engine()
    h = _arr(a)
    vecs = h.reshape(-1, h.shape[-1])
    d = eng.upload(vecs)
    s = eng.alloc(vecs.size)
    check(lib.fhe_ntt_cyclic(eng._h, d.ptr, s.ptr, _log2(vecs.shape[1]), vecs.shape[0], mod, root, convention, inverse, None))
    out = d.download().reshape(h.shape)
    return out


def ntt(a, mod: int, root: int, eng: Optional[Engine] = None) -> List[int]:
    """``ntt(a, mod, root)`` of motivation/ntt.py:8-32 (``root`` generates Z_mod*)."""
    return [int(x) for x in _cyclic(a, mod, root, 0, 0, eng)]


def intt(a, mod: int, root: int, eng: Optional[Engine] = None) -> List[int]:
    """``intt(a, mod, root)`` of motivation/bsgs.py:31-36."""
    return [int(x) for x in _cyclic(a, mod, root, 0, 1, eng)]


def ntt_nthroot(a, root: int, mod: int, eng: Optional[Engine] = None) -> List[int]:
    """``ntt(a, root, mod)`` of rfhe_framewk/src/negaclic_ntt.py:38-57 (``root`` is an n-th root)."""
    return [int(x) for x in _cyclic(a, mod, root, 1, 0, eng)]


def intt_nthroot(a, root: int, mod: int, eng: Optional[Engine] = None) -> List[int]:
    """``intt(a, root, mod)`` of rfhe_framewk/src/negaclic_ntt.py:77-83."""
    return [int(x) for x in _cyclic(a, mod, root, 1, 1, eng)]


def _psi_tables(eng: Engine, n: int, psi: int, mod: int) -> NttTables:
    log_n = _log2(n)
    rp = np.zeros(n, dtype=_U64)
    p = 1
    for i in range(n):
        rp[int(f"{i:0{log_n}b}"[::-1], 2)] = p
        p = p * psi % mod
    return eng.tables_from_roots(log_n, [mod], rp)


def negacyclic_ntt(a, psi: int, mod: int, eng: Optional[Engine] = None) -> List[int]:
    """``negacyclic_ntt(a, psi, mod)`` of rfhe_framewk/src/negaclic_ntt.py:86-92 (natural-order output)."""
    eng = eng or default_engine()
    n = len(a)
    t = _psi_tables(eng, n, psi, mod)
    d = eng.upload(a)
    t.forward(d)
    out = eng.alloc(n)
    check(lib.fhe_bitrev_permute(eng._h, out.ptr, d.ptr, t.log_n, 1, None))
    return [int(x) for x in out.download()]


def negacyclic_intt(A, psi: int, mod: int, eng: Optional[Engine] = None) -> List[int]:
    """``negacyclic_intt(A, psi, mod)`` of rfhe_framewk/src/negaclic_ntt.py:102-109."""
    eng = eng or default_engine()
    n = len(A)
    t = _psi_tables(eng, n, psi, mod)
    src = eng.upload(A)
    d = eng.alloc(n)
    check(lib.fhe_bitrev_permute(eng._h, d.ptr, src.ptr, t.log_n, 1, None))
    t.inverse(d)
    return [int(x) for x in d.download()]


def poly_mul_negacyclic_ntt(a, b, psi: int, mod: int, eng: Optional[Engine] = None) -> List[int]:
    """``poly_mul_negacyclic_ntt`` of rfhe_framewk/src/negaclic_ntt.py:123-127."""
    eng = eng or default_engine()
    t = _psi_tables(eng, len(a), psi, mod)
    da, db = eng.upload(a), eng.upload(b)
    t.polymul(da, da, db)
    return [int(x) for x in da.download()]


def four_step_ntt(a, N: int, mod: int = 998244353, g: int = 3, n1: Optional[int] = None, eng: Optional[Engine] = None):
    """``four_step_ntt(a, N, mod, g)`` of reliability_test/four_step_ntt_prot.py:71-109.
    ``n1`` defaults to sqrt(N) as in the reference (:73-75); any power-of-two split is accepted.  ``a`` may be a batch
    ([n_vec][N]): one call transforms all vectors (two launches in total) and a list of lists comes back."""
    eng = eng or default_engine()
    log_n = _log2(N)
    arr = _arr(a)
    if arr.ndim == 2:
        if n1 is None:
            n1 = 1 << (log_n // 2)
        h = vp()
        check(lib.fhe_fourstep_create(eng._h, n1, N // n1, mod, g, C.byref(h)))
        try:
            src = eng.upload(arr)
            dst = eng.alloc(arr.size)
            check(lib.fhe_fourstep_ntt_batch(eng._h, dst.ptr, src.ptr, h, arr.shape[0], None))
            return dst.download().reshape(arr.shape).tolist()
        finally:
            lib.fhe_fourstep_destroy(h)
    if n1 is None:
        n1 = 1 << (log_n // 2)
        if n1 * n1 != N:
            raise AssertionError("N must be a perfect square unless n1 is given")  # four_step_ntt_prot.py:74
    n2 = N // n1
    h = vp()
    check(lib.fhe_fourstep_create(eng._h, n1, n2, mod, g, C.byref(h)))
    try:
        src = eng.upload(a)
        dst = eng.alloc(N)
        check(lib.fhe_fourstep_ntt(eng._h, dst.ptr, src.ptr, h, None))
        return [int(x) for x in dst.download()]
    finally:
        lib.fhe_fourstep_destroy(h)


class BaseConv:
    """Base-conversion plan for (moduli_in -> moduli_out)."""

    def __init__(self, eng: Engine, moduli_in: Sequence[int], moduli_out: Sequence[int]):
        self.eng, self.m, self.k = eng, len(moduli_in), len(moduli_out)
        mi, mo = _arr(moduli_in), _arr(moduli_out)
        h = vp()
        check(lib.fhe_baseconv_create(eng._h, _ptr(mi), self.m, _ptr(mo), self.k, C.byref(h)))
        self._h = h

    def exact(self, out: DeviceArray, inp: DeviceArray, N: int, stream=None):
        check(lib.fhe_baseconv_exact(self.eng._h, out.ptr, inp.ptr, self._h, N, stream))

    def fast(self, out: DeviceArray, inp: DeviceArray, N: int, stream=None):
        check(lib.fhe_baseconv_fast(self.eng._h, out.ptr, inp.ptr, self._h, N, stream))

    def __del__(self):
        try:
            if self._h and self.eng._h:
                lib.fhe_baseconv_destroy(self._h)
        except Exception:
            pass


def base_conv_fixed(residue_arrays, moduli_in, moduli_out, eng: Optional[Engine] = None) -> List[List[int]]:
    """``base_conv_fixed`` of motivation/baseConv.py:67-83; returns ``[k][i]``."""
    eng = eng or default_engine()
    r = _arr(residue_arrays)
    N = r.shape[1]
    bc = BaseConv(eng, moduli_in, moduli_out)
    d = eng.upload(r)
    o = eng.alloc(len(moduli_out) * N)
    bc.exact(o, d, N)
    return o.download().reshape(len(moduli_out), N).tolist()


def bConv(residue_arrays, moduli, moduli_out, eng: Optional[Engine] = None) -> List[List[int]]:
    """``bConv`` of rfhe_framewk/src/baseConv.py:10-40; returns ``[i][k]`` like the reference."""
    eng = eng or default_engine()
    r = _arr(residue_arrays)
    N = r.shape[1]
    bc = BaseConv(eng, moduli, moduli_out)
    d = eng.upload(r)
    o = eng.alloc(len(moduli_out) * N)
    bc.fast(o, d, N)
    return o.download().reshape(len(moduli_out), N).T.tolist()


def crt_garner(residues, moduli, eng: Optional[Engine] = None):
    """``crt_kernel`` of rfhe_framewk/src/baseConv.cu:85-120; returns (x_lo, x_hi) arrays."""
    eng = eng or default_engine()
    r = _arr(residues)
    m, N = r.shape
    mod = _arr(moduli)
    d = eng.upload(r)
    lo, hi = eng.alloc(N), eng.alloc(N)
    check(lib.fhe_crt_garner(eng._h, lo.ptr, hi.ptr, d.ptr, _ptr(mod), m, N, None))
    return lo.download(), hi.download()


def diag_block_hadamard_matvec(M_blocks, v, mod: int = 0, eng: Optional[Engine] = None) -> np.ndarray:
    """``diag_block_hadamard_matvec`` of motivation/bsgs.py:39-52 (``mod=0``: the reference's
    unreduced int64 arithmetic)."""
    eng = eng or default_engine()
    M = np.ascontiguousarray(np.asarray(M_blocks, dtype=np.int64)).view(_U64)
    vv = np.ascontiguousarray(np.asarray(v, dtype=np.int64)).view(_U64)
    k, bs = M.shape
    dM, dv = eng.upload(M), eng.upload(vv)
    y = eng.alloc(k * bs)
    check(lib.fhe_bsgs_hadamard(eng._h, y.ptr, dM.ptr, dv.ptr, k, bs, mod, None))
    out = y.download()
    return out.view(np.int64) if mod == 0 else out


# ---------------------------------------------------------------------------
# Rotation / key switching
# ---------------------------------------------------------------------------
def automorphism(eng: Engine, t: NttTables, src: DeviceArray, galois_elt: int, n_poly: int = 1, limbs: Optional[int] = None,
                 start: int = 0, ntt_domain: bool = False) -> DeviceArray:
    """x -> x^galois_elt on every limb (coefficient domain, or NTT domain when ``ntt_domain``)."""
    limbs = len(t) - start if limbs is None else limbs
    dst = eng.alloc(src.size)
    dst.shape = src.shape
    if ntt_domain:
        check(lib.fhe_automorphism_ntt(eng._h, dst.ptr, src.ptr, t.log_n, galois_elt, n_poly * limbs, None))
    else:
        check(lib.fhe_automorphism(eng._h, dst.ptr, src.ptr, t._h, galois_elt, n_poly, limbs, start, None))
    return dst


class KeySwitch:
    """Hybrid RNS key switching over the primes of ``t`` (L ciphertext primes then K special primes,
    ``dnum`` digits); operation sequence of the reference's SEAL trace
    (profile_framewk/build/data/ckks/16384_4:466-539)."""

    def __init__(self, eng: Engine, t: NttTables, L: int, K: int, dnum: int):
        self.eng, self.t, self.L, self.K, self.dnum = eng, t, L, K, dnum
        h = vp()
        check(lib.fhe_keyswitch_create(eng._h, t._h, L, K, dnum, C.byref(h)))
        self._h = h

    def apply(self, c: DeviceArray, evk: DeviceArray, stream=None):
        """c: [L][N] NTT domain; evk: [dnum][2][L+K][N] NTT domain -> (out0, out1), each [L][N] NTT domain."""
        n = self.L * self.t.N
        o0, o1 = self.eng.alloc(n), self.eng.alloc(n)
        o0.shape = o1.shape = (self.L, self.t.N)
        check(lib.fhe_keyswitch_apply(self.eng._h, self._h, o0.ptr, o1.ptr, c.ptr, evk.ptr, stream))
        return o0, o1

    def rotate(self, c0: DeviceArray, c1: DeviceArray, galois_elt: int, galois_key: DeviceArray, stream=None):
        """``rotate_inplace`` (dotprod_test.cu:146) / frontend ROTATE of the SEAL traces: (c0, c1) -> (out0, out1)."""
        n = self.L * self.t.N
        o0, o1 = self.eng.alloc(n), self.eng.alloc(n)
        o0.shape = o1.shape = (self.L, self.t.N)
        check(lib.fhe_rotate(self.eng._h, self._h, o0.ptr, o1.ptr, c0.ptr, c1.ptr, galois_elt, galois_key.ptr, stream))
        return o0, o1

    def prepare_galois_key(self, galois_key: DeviceArray, galois_elt: int, stream=None) -> DeviceArray:
        """The key in the un-rotated frame (sigma^-1 of every key row): what ``rotate_hoisted`` takes; once per key."""
        out = self.eng.alloc(self.dnum * 2 * (self.L + self.K) * self.t.N)
        out.shape = (self.dnum, 2, self.L + self.K, self.t.N)
        check(lib.fhe_galois_key_prepare(self.eng._h, self._h, out.ptr, galois_key.ptr, galois_elt, stream))
        return out

    def rotate_hoisted(self, c0: DeviceArray, c1: DeviceArray, galois_elts, prepared_keys, stream=None):
        """Rotations of ONE ciphertext by several Galois elements with the decomposition of c1 shared (the baby steps of
        profile_framewk/src/matmul_ckks.cpp:45-113): a list of (out0, out1)."""
        n = len(galois_elts)
        outs = []
        for _ in range(n):
            o0, o1 = self.eng.alloc(self.L * self.t.N), self.eng.alloc(self.L * self.t.N)
            o0.shape = o1.shape = (self.L, self.t.N)
            outs.append((o0, o1))
        a0 = (vp * n)(*[o[0].ptr for o in outs])
        a1 = (vp * n)(*[o[1].ptr for o in outs])
        ks = (vp * n)(*[k.ptr for k in prepared_keys])
        ge = (C.c_uint32 * n)(*[int(g) for g in galois_elts])
        check(lib.fhe_rotate_hoisted(self.eng._h, self._h, a0, a1, c0.ptr, c1.ptr, ge, ks, n, stream))
        return outs

    def bsgs_matvec(self, c0: DeviceArray, c1: DeviceArray, diags: DeviceArray, n1: int, n2: int, baby_elts, baby_keys_prepared,
                    giant_elts, giant_keys, stream=None):
        """Baby-step / giant-step matrix-vector product (profile_framewk/src/matmul_ckks.cpp:45-113): diags = [n2][n1][L][N]."""
        o0, o1 = self._out(self.L), self._out(self.L)
        be = (C.c_uint32 * max(1, n1 - 1))(*[int(g) for g in baby_elts])
        ge = (C.c_uint32 * max(1, n2 - 1))(*[int(g) for g in giant_elts])
        bk = (vp * max(1, n1 - 1))(*[k.ptr for k in baby_keys_prepared])
        gk = (vp * max(1, n2 - 1))(*[k.ptr for k in giant_keys])
        check(lib.fhe_bsgs_matvec(self.eng._h, self._h, o0.ptr, o1.ptr, c0.ptr, c1.ptr, diags.ptr, n1, n2, be, bk, ge, gk, stream))
        return o0, o1

    def set_plain_modulus(self, t: int):
        """BGV form of the mod-down and of the rescale (0 = CKKS-style flooring)."""
        check(lib.fhe_keyswitch_set_plain_modulus(self._h, t))

    def _out(self, limbs: int) -> DeviceArray:
        o = self.eng.alloc(limbs * self.t.N)
        o.shape = (limbs, self.t.N)
        return o

    def tensor(self, a0: DeviceArray, a1: DeviceArray, b0: DeviceArray, b1: DeviceArray, stream=None):
        """``phantom::multiply`` (dotprod_test.cu:113): (d0, d1, d2), each [L][N], NTT domain."""
        d = [self._out(self.L) for _ in range(3)]
        check(lib.fhe_tensor_product(self.eng._h, d[0].ptr, d[1].ptr, d[2].ptr, a0.ptr, a1.ptr, b0.ptr, b1.ptr, self.t._h, self.L, 0, stream))
        return tuple(d)

    def relinearize(self, d0: DeviceArray, d1: DeviceArray, d2: DeviceArray, relin_key: DeviceArray, stream=None):
        """``relinearize_inplace`` (dotprod_test.cu:114)."""
        o0, o1 = self._out(self.L), self._out(self.L)
        check(lib.fhe_relinearize(self.eng._h, self._h, o0.ptr, o1.ptr, d0.ptr, d1.ptr, d2.ptr, relin_key.ptr, stream))
        return o0, o1

    def rescale(self, c: DeviceArray, n_parts: int = 2, stream=None) -> DeviceArray:
        """``mod_switch_to_next_inplace`` (dotprod_test.cu:115): [n_parts][L][N] -> [n_parts][L-1][N]."""
        o = self.eng.alloc(n_parts * (self.L - 1) * self.t.N)
        o.shape = (n_parts, self.L - 1, self.t.N)
        check(lib.fhe_rescale(self.eng._h, self._h, o.ptr, c.ptr, n_parts, stream))
        return o

    def hmult(self, a0: DeviceArray, a1: DeviceArray, b0: DeviceArray, b1: DeviceArray, relin_key: DeviceArray, rescale: bool = True,
              stream=None):
        """multiply -> relinearize -> mod_switch_to_next (dotprod_test.cu:113-115) in one call."""
        limbs = self.L - 1 if rescale else self.L
        o0, o1 = self._out(limbs), self._out(limbs)
        check(lib.fhe_hmult(self.eng._h, self._h, o0.ptr, o1.ptr, a0.ptr, a1.ptr, b0.ptr, b1.ptr, relin_key.ptr, 1 if rescale else 0, stream))
        return o0, o1

    def __del__(self):
        try:
            if self._h and self.eng._h:
                lib.fhe_keyswitch_destroy(self._h)
        except Exception:
            pass


# ---------------------------------------------------------------------------
# ABFT detector
# ---------------------------------------------------------------------------
class Abft:
    """Weighted-checksum ECC around the forward NTT (rfhe_framewk/src/negaclic_ntt.py:130-149)."""

    def __init__(self, eng: Engine, t: NttTables):
        self.eng, self.t = eng, t
        h = vp()
        check(lib.fhe_abft_create(eng._h, t._h, C.byref(h)))
        self._h = h

    def checksum(self, d: DeviceArray, side: int, n_poly: int = 1, limbs: Optional[int] = None, start: int = 0) -> np.ndarray:
        limbs = len(self.t) - start if limbs is None else limbs
        out = self.eng.alloc(n_poly * limbs)
        check(lib.fhe_abft_checksum(self.eng._h, self._h, side, d.ptr, out.ptr, n_poly, limbs, start, None))
        return out.download()

    def forward_checked(self, d: DeviceArray, n_poly: int = 1, limbs: Optional[int] = None, start: int = 0) -> np.ndarray:
        """In-place forward NTT; returns the per-limb-polynomial fault flags."""
        limbs = len(self.t) - start if limbs is None else limbs
        flags = self.eng.alloc((n_poly * limbs + 1) // 2)      # uint32 flags packed in a u64 buffer
        check(lib.fhe_ntt_forward_checked(self.eng._h, d.ptr, self.t._h, self._h, n_poly, limbs, start, flags.ptr, None))
        return flags.download().view(np.uint32)[: n_poly * limbs]

    def forward_checked_phases(self, d: DeviceArray, n_poly: int = 1, limbs: Optional[int] = None, start: int = 0) -> np.ndarray:
        """In-place forward NTT with the per-phase detector; returns flags[unit, 3] = (column pass, hand-off, row pass)."""
        limbs = len(self.t) - start if limbs is None else limbs
        n = n_poly * limbs * 3
        flags = self.eng.alloc((n + 1) // 2)
        check(lib.fhe_ntt_forward_checked_phases(self.eng._h, d.ptr, self.t._h, self._h, n_poly, limbs, start, flags.ptr, None))
        return flags.download().view(np.uint32)[:n].reshape(n_poly * limbs, 3)

    def __del__(self):
        try:
            if self._h and self.eng._h:
                lib.fhe_abft_destroy(self._h)
        except Exception:
            pass
