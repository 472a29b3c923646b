"""CPU emulation of the HIP NTT passes (tests/emu/emu_ntt.cpp compiles the very same
template code the kernels instantiate) against the oracle: checks tile/twiddle
indexing, the LDS exchange addressing and the FP64 lazy-range schedule for every
supported size, without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import cport as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "fhe_reliability_gpu_amd", "csrc")
p64 = C.POINTER(C.c_uint64)


@pytest.fixture(scope="module")
def emu():
    # FHE_EMU_SANITIZE=1 (with LD_PRELOAD=$(gcc -print-file-name=libasan.so)): AddressSanitizer + UBSan build of the
    # emulation -- the CPU-side sanitizer run of the kernels' indexing (GPU sanitizers are not available on the pool)
    san = os.environ.get("FHE_EMU_SANITIZE") == "1"
    so = os.path.join(EMU_DIR, "libemu_ntt_san.so" if san else "libemu_ntt.so")
    srcs = [os.path.join(EMU_DIR, "emu_ntt.cpp")] + [os.path.join(CSRC, f) for f in ("modarith.hpp", "ntt_core.hpp", "ntt_plan.hpp", "ntt_fused.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] if san else ["-O2"]
        subprocess.check_call(["g++"] + flags + ["-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-I" + CSRC, srcs[0], "-o", so])
    L = C.CDLL(so)
    L.emu_ntt.restype = C.c_int
    L.emu_ntt.argtypes = [p64, C.c_int, C.c_int, C.c_int, C.c_int, p64, p64, C.c_int, C.c_int]
    L.emu_max_ratio.restype = C.c_double
    return L


def _run(emu, data, logn, inverse, qs, rps, path, fused_dist=0):
    d = np.ascontiguousarray(data, dtype=np.uint64).copy()
    n_poly, limbs, _ = d.shape
    q = np.asarray(qs, dtype=np.uint64)
    rp = np.ascontiguousarray(rps, dtype=np.uint64)
    rc = emu.emu_ntt(d.ctypes.data_as(p64), logn, inverse, n_poly, limbs, q.ctypes.data_as(p64), rp.ctypes.data_as(p64), path, fused_dist)
    assert rc == 0, rc
    return d


def _tables(logn, bits, limbs):
    N = 1 << logn
    qs = O.gen_primes(max(N, 2), bits, limbs)
    rps = np.stack([O.root_powers(q, logn) for q in qs])
    return qs, rps


@pytest.mark.parametrize("logn", list(range(1, 21)))
@pytest.mark.parametrize("path,bits", [(0, 50), (1, 61), (1, 50), (0, 30)])
def test_emulated_passes_match_oracle(emu, logn, path, bits):
    if logn >= 15 and (path, bits) in ((1, 50), (0, 30)):
        pytest.skip("covered by the smaller sizes")
    limbs = 2 if logn <= 14 else 1
    n_poly = 2 if logn <= 12 else 1
    N = 1 << logn
    qs, rps = _tables(logn, bits, limbs)
    rng = np.random.default_rng(logn * 10 + path)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])
    # worst-case-ish inputs for the lazy ranges: all q-1 in one polynomial
    data[0, 0, :] = qs[0] - 1
    fwd = _run(emu, data, logn, 0, qs, rps, path)
    if path == 0:
        assert emu.emu_max_ratio() < 8.0
    for p in range(n_poly):
        for l in range(limbs):
            want = O.nwt_forward(data[p, l], qs[l], rps[l])
            assert (fwd[p, l] == want).all(), f"forward mismatch poly {p} limb {l}"
    back = _run(emu, fwd, logn, 1, qs, rps, path)
    if path == 0:
        assert emu.emu_max_ratio() < 8.0
    assert (back == data).all()


@pytest.mark.parametrize("path", [0, 1])
def test_out_of_range_words_are_reduced_first(emu, path):
    # bit-flipped symbols (reliability_test/ntt_test.cu:104-135) may exceed q
    logn, N = 10, 1024
    qs, rps = _tables(logn, 50, 1)
    rng = np.random.default_rng(7)
    data = rng.integers(0, qs[0], (1, 1, N), dtype=np.uint64)
    data[0, 0, 5] ^= np.uint64(1 << 63)
    data[0, 0, 17] ^= np.uint64(1 << 51)
    data[0, 0, 99] = np.uint64(qs[0])          # == q
    data[0, 0, 100] = np.uint64(2**64 - 1)
    fwd = _run(emu, data, logn, 0, qs, rps, path)
    assert (fwd[0, 0] == O.nwt_forward(data[0, 0], qs[0], rps[0])).all()
    inv = _run(emu, data, logn, 1, qs, rps, path)
    assert (inv[0, 0] == O.nwt_inverse(data[0, 0], qs[0], rps[0])).all()


@pytest.mark.parametrize("logn", [13, 14, 15, 16, 17])
@pytest.mark.parametrize("path,bits", [(0, 50), (1, 61)])
@pytest.mark.parametrize("dist", [1, 3, 9])
def test_fused_schedule_matches_oracle(emu, logn, path, bits, dist):
    # the single-launch kernel's ticket schedule and tile geometry, run by one sequential team
    # (dist = 9 > number of limbs: the team runs out of limbs before its first second pass)
    N = 1 << logn
    limbs, n_poly = (2, 3) if logn <= 15 else (2, 2)
    qs, rps = _tables(logn, bits, limbs)
    rng = np.random.default_rng(logn + 100 * path + dist)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])
    fwd = _run(emu, data, logn, 0, qs, rps, path, fused_dist=dist)
    for p in range(n_poly):
        for l in range(limbs):
            assert (fwd[p, l] == O.nwt_forward(data[p, l], qs[l], rps[l])).all()
    back = _run(emu, fwd, logn, 1, qs, rps, path, fused_dist=dist)
    assert (back == data).all()


@pytest.mark.parametrize("logn", [8, 12, 13, 16])
@pytest.mark.parametrize("path,bits", [(0, 50), (1, 61)])
def test_explicit_unit_list_addressing(emu, logn, path, bits):
    # PassArgs::map (the key switch transforms "every limb of every digit except its own" through it): the same batch
    # listed explicitly, in reverse order, must give the same transform
    N = 1 << logn
    limbs, n_poly = 3, 2
    qs, rps = _tables(logn, bits, limbs)
    rng = np.random.default_rng(7 * logn + path)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])
    fwd = _run(emu, data, logn, 0, qs, rps, path, fused_dist=-1)
    for p in range(n_poly):
        for l in range(limbs):
            assert (fwd[p, l] == O.nwt_forward(data[p, l], qs[l], rps[l])).all()
    assert (_run(emu, fwd, logn, 1, qs, rps, path, fused_dist=-1) == data).all()


@pytest.mark.parametrize("logn", [13, 14])
@pytest.mark.parametrize("path,bits", [(0, 50), (1, 61), (0, 30)])
def test_lds_resident_single_pass(emu, logn, path, bits):
    # 2^13 / 2^14: the whole limb in one workgroup's LDS, radix-32 register steps (ntt_plan.hpp ResidentPlan)
    N = 1 << logn
    limbs, n_poly = 2, 2
    qs, rps = _tables(logn, bits, limbs)
    rng = np.random.default_rng(logn + path)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])
    data[0, 0, :] = qs[0] - 1
    fwd = _run(emu, data, logn, 0, qs, rps, path, fused_dist=-2)
    for p in range(n_poly):
        for l in range(limbs):
            assert (fwd[p, l] == O.nwt_forward(data[p, l], qs[l], rps[l])).all()
    assert (_run(emu, fwd, logn, 1, qs, rps, path, fused_dist=-2) == data).all()
    inv = _run(emu, data, logn, 1, qs, rps, path, fused_dist=-2)
    for l in range(limbs):
        assert (inv[0, l] == O.nwt_inverse(data[0, l], qs[l], rps[l])).all()


@pytest.mark.parametrize("bits", [50, 30, 49])
def test_packed_handoff_forward(emu, bits):
    # forward 2^16 with the 50-bit packed hand-off between the passes (ntt_core.hpp): same words as the oracle; extreme
    # residues (q - 1 everywhere) exercise every bit of the 50-bit fields
    logn, N, path = 16, 1 << 16, 0
    limbs, n_poly = 2, 2
    qs, rps = _tables(logn, bits, limbs)
    rng = np.random.default_rng(bits)
    data = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(n_poly)])
    data[0, 0, :] = qs[0] - 1
    data[1, 1, ::3] = 0
    fwd = _run(emu, data, logn, 0, qs, rps, path, fused_dist=-3)
    for p in range(n_poly):
        for l in range(limbs):
            assert (fwd[p, l] == O.nwt_forward(data[p, l], qs[l], rps[l])).all()


def test_inverse_lazy_range_plan_is_sound(emu):
    """The per-register lazy-range plan of the FP64 inverse passes (ntt_core.hpp inv_lazy_plan, evaluated by the kernels' templates at
    compile time): an independent restatement with exact fractions follows the plan's fold masks and checks that no butterfly ever sees
    |X| + |Y| above 8 q (q < 2^50: every sum, difference and product input below 2^53, an exact integer in a double), that the registers
    a step hands over are within the bound it declares, and that the bounds chain through the steps of a pass as Passes<> assumes.
    Model of one Gentleman-Sande stage: (bX, bY) -> (bX + bY, 1/2 + (bX + bY) / 4 + slack); a fold leaves 1/2 + slack
    (ArithF64::mulmod / reduce, modarith.hpp)."""
    from fractions import Fraction as Fr
    u32p = C.POINTER(C.c_uint32)
    emu.emu_inv_lazy_plan.restype = C.c_int
    emu.emu_inv_lazy_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u32p, u32p, C.POINTER(C.c_int)]
    slack = Fr(1, 1 << 40)
    fewest = {}
    for K in range(1, 6):
        R = 1 << K
        for in8 in (4, 5, 8, 9, 12, 16, 17, 24, 32, 64):
            for exit8 in (16, 64):
                for fold in (0, 1):
                    before = (C.c_uint32 * K)()
                    at_exit, out8 = C.c_uint32(), C.c_int()
                    assert emu.emu_inv_lazy_plan(K, in8, exit8, fold, before, C.byref(at_exit), C.byref(out8)) == 0
                    b = [Fr(in8, 8)] * R
                    folds = 0
                    for v in range(K):
                        u = K - 1 - v
                        half = R >> (u + 1)
                        for r in range(R):
                            if (before[v] >> r) & 1:
                                b[r] = Fr(1, 2) + slack
                                folds += 1
                        for blk in range(1 << u):
                            for j in range(half):
                                i0 = blk * 2 * half + j
                                i1 = i0 + half
                                ssum = b[i0] + b[i1]
                                assert ssum <= 8, (K, in8, exit8, fold, v, i0, i1, float(ssum))
                                prod = Fr(1, 2) + ssum / 4 + slack
                                b[i0] = prod if (fold and u == 0) else ssum
                                b[i1] = prod
                    for r in range(R):
                        if (at_exit.value >> r) & 1:
                            b[r] = Fr(1, 2) + slack
                            folds += 1
                    assert max(b) <= Fr(out8.value, 8), (K, in8, exit8, fold, float(max(b)), out8.value)
                    assert max(b) <= max(Fr(exit8, 8), Fr(out8.value, 8)) and (exit8 == 64 or out8.value <= max(exit8, 8) or fold)
                    fewest[(K, in8, exit8, fold)] = Fr(folds, R)
    # the point of it: N = 2^16 (Steps<4,4> twice) folds fewer than 3 registers per point where the uniform schedule folds 5
    emu.emu_inv_lazy_chain.restype = C.c_int
    chain = [8]
    for se in (1, 2):
        chain.append(emu.emu_inv_lazy_chain(4, 4, 0, 8, se))
    assert chain[1] <= 16 and chain[2] <= 16
    second = [chain[2], emu.emu_inv_lazy_chain(4, 4, 0, chain[2], 1)]
    total = Fr(0)
    for in8, (exit8, fold) in zip([chain[0], chain[1]] + second, [(16, 0), (16, 0), (16, 0), (64, 1)]):
        key = (4, in8, exit8, fold)
        if key not in fewest:          # bounds off the sampled grid: count through the plan again
            before = (C.c_uint32 * 4)()
            at_exit, out8 = C.c_uint32(), C.c_int()
            emu.emu_inv_lazy_plan(4, in8, exit8, fold, before, C.byref(at_exit), C.byref(out8))
            fewest[key] = Fr(sum(bin(m).count("1") for m in before) + bin(at_exit.value).count("1"), 16)
        total += fewest[key]
    assert total < 3, float(total)
