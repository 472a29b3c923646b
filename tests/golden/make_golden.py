#!/usr/bin/env python3
"""Generate tests/golden/*.json by IMPORTING the reference's own Python.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py [/root/reference]

The fixtures are data (inputs + the reference's outputs); no reference source text
is stored.  Modules imported (read-only, Agg backend, scratch cwd because
``motivation/ntt.py`` and ``bsgs.py`` plot at import time):

    motivation/ntt.py            ntt                                   (a1)
    motivation/bsgs.py           ntt, intt, diag_block_hadamard_matvec  (a1, a3, a9)
    motivation/baseConv.py       base_conv_fixed                        (a8 exact)
    rfhe_framewk/src/ntt.py      ntt, intt, poly_mul_ntt, poly_mul_naive(a1, a3)
    rfhe_framewk/src/negaclic_ntt.py  negacyclic_ntt/intt, poly_mul_*   (a2, a3, a5)
    rfhe_framewk/src/baseConv.py bConv                                  (a8 fast)
    reliability_test/four_step_ntt_prot.py  ntt_direct, four_step_ntt   (a6)

Large cases store a SHA-256 of the little-endian u64 output plus head/tail words
instead of the whole vector.
"""
import hashlib
import importlib.util
import json
import os
import random
import sys
import tempfile

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

os.environ["MPLBACKEND"] = "Agg"
_scratch = tempfile.mkdtemp(prefix="golden_")
os.chdir(_scratch)


def load(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha_u64(vals):
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v).to_bytes(8, "little"))
    return h.hexdigest()


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(OUT, name)), "bytes")


Q61 = 2305843009211596801  # largest 61-bit prime = 1 mod 2^18; generator 37
Q50 = 1125899903107073     # first logged Phantom prime (bits1-16_num1.txt:10)
PSI_16384 = 32853495844    # its minimal primitive 32768-th root (SURVEY KAT-2)


def main():
    import warnings
    warnings.filterwarnings("ignore")

    mot_ntt = load("motivation/ntt.py", "ref_mot_ntt")
    mot_bsgs = load("motivation/bsgs.py", "ref_mot_bsgs")
    mot_bc = load("motivation/baseConv.py", "ref_mot_bc")
    r_ntt = load("rfhe_framewk/src/ntt.py", "ref_r_ntt")
    r_neg = load("rfhe_framewk/src/negaclic_ntt.py", "ref_r_neg")
    r_bc = load("rfhe_framewk/src/baseConv.py", "ref_r_bc")
    four = load("reliability_test/four_step_ntt_prot.py", "ref_four")

    # ---------------- a1: cyclic NTT (motivation/ntt.py:8-32)
    cyc = {"cases": []}
    # module-level demo (composite modulus 15728641, root 3, a[i] = i % 16): ntt.py:35-43
    cyc["demo"] = {"mod": mot_ntt.mod, "root": mot_ntt.root, "n": mot_ntt.n,
                   "a": list(mot_ntt.a), "A": [int(x) for x in mot_ntt.A]}
    for lg, (mod, root) in ((4, (97, 5)), (8, (Q61, 37)), (10, (Q61, 37)), (12, (Q61, 37)), (16, (Q61, 37))):
        random.seed(lg)
        a = [random.randrange(mod) for _ in range(1 << lg)]
        out = mot_ntt.ntt(list(a), mod, root)
        case = {"lg": lg, "mod": mod, "root": root, "seed": lg, "sha256": sha_u64(out),
                "head": out[:8], "tail": out[-8:]}
        if lg <= 10:
            case["a"], case["out"] = a, out
        cyc["cases"].append(case)
    # bsgs.py twin + its inverse (a3)
    random.seed(77)
    a = [random.randrange(Q61) for _ in range(256)]
    f = mot_bsgs.ntt(list(a), Q61, 37)
    cyc["bsgs_twin"] = {"mod": Q61, "root": 37, "a": a, "fwd": [int(x) for x in f],
                        "inv_of_fwd": [int(x) for x in mot_bsgs.intt(list(f), Q61, 37)]}
    # rfhe_framewk/src/ntt.py:38-62 -- same generator convention as motivation/ntt.py,
    # argument order (a, root, mod)
    mod, n = 998244353, 128
    random.seed(5)
    a = [random.randrange(mod) for _ in range(n)]
    cyc["rfhe_twin"] = {"mod": mod, "root": 3, "a": a, "fwd": r_ntt.ntt(list(a), 3, mod),
                        "inv": r_ntt.intt(list(a), 3, mod)}
    # rfhe_framewk/src/negaclic_ntt.py:38-57,77-83 -- root is a primitive n-th root
    root = pow(3, (mod - 1) // n, mod)
    cyc["nthroot"] = {"mod": mod, "root": root, "a": a, "fwd": r_neg.ntt(list(a), root, mod),
                      "inv": r_neg.intt(list(a), root, mod)}
    dump("cyclic_ntt.json", cyc)

    # ---------------- a2/a3/a5: negacyclic (rfhe_framewk/src/negaclic_ntt.py)
    neg = {"cases": []}
    for n, q, psi in ((8, 17, 3), (64, Q50, pow(PSI_16384, 16384 // 64, Q50)),
                      (1024, Q50, pow(PSI_16384, 16, Q50)), (16384, Q50, PSI_16384)):
        random.seed(n)
        a = [random.randrange(q) for _ in range(n)]
        fwd = r_neg.negacyclic_ntt(list(a), psi, q)
        back = r_neg.negacyclic_intt(list(fwd), psi, q)
        assert back == a
        case = {"n": n, "q": q, "psi": psi, "seed": n, "sha256_fwd": sha_u64(fwd),
                "head": fwd[:8], "tail": fwd[-8:]}
        if n <= 1024:
            case["a"], case["fwd"] = a, fwd
        neg["cases"].append(case)
    pm = []
    for n, q, psi in ((8, 17, 3), (32, Q50, pow(PSI_16384, 16384 // 32, Q50)), (64, Q50, pow(PSI_16384, 256, Q50))):
        random.seed(1000 + n)
        a = [random.randrange(q) for _ in range(n)]
        b = [random.randrange(q) for _ in range(n)]
        naive = r_neg.poly_mul_naive_negacyclic(a, b, q)
        viantt = r_neg.poly_mul_negacyclic_ntt(a, b, psi, q)
        assert naive == viantt
        pm.append({"n": n, "q": q, "psi": psi, "a": a, "b": b, "c": naive})
    neg["polymul"] = pm
    dump("negacyclic.json", neg)

    # ---------------- a6: four-step (reliability_test/four_step_ntt_prot.py)
    fs = {"mod": four.MOD, "g": four.G, "cases": []}
    for N in (16, 64, 256):
        random.seed(N)
        a = [random.randrange(four.MOD) for _ in range(N)]
        d = four.ntt_direct(a, N)
        y = four.four_step_ntt(a, N)
        assert d == y
        fs["cases"].append({"N": N, "a": a, "y": [int(v) for v in y]})
    dump("four_step.json", fs)

    # ---------------- a8: base conversion
    bc = {}
    random.seed(9)
    mi = [1073741827, 1073741831, 1073741833]          # 31-bit primes
    mo = [1099511627791, 1099511627803, 2305843009213693951, 1125899903107073]
    N = 48
    P = 1
    for p in mi:
        P *= p
    vals = [random.randrange(P) for _ in range(N)]
    res = mot_bc.values_to_rns(vals, mi)
    bc["exact"] = {"mod_in": mi, "mod_out": mo, "values": [str(v) for v in vals], "res": res,
                   "out": mot_bc.base_conv_fixed(res, mi, mo)}
    mi2 = [1125899903107073, 1125899903500289, 1125899903795201, 1125899903827969]
    mo2 = [1125899903991809, 1125899904679937, 2305843009211596801]
    res2 = [[random.randrange(p) for _ in range(N)] for p in mi2]
    bc["exact50"] = {"mod_in": mi2, "mod_out": mo2, "res": res2,
                     "out": mot_bc.base_conv_fixed(res2, mi2, mo2)}
    bc["fast"] = {"mod_in": mi2, "mod_out": mo2, "res": res2,
                  "out": [[int(x) for x in row] for row in r_bc.bConv(res2, mi2, mo2)]}
    bc["fast31"] = {"mod_in": mi, "mod_out": mo, "res": res,
                    "out": [[int(x) for x in row] for row in r_bc.bConv(res, mi, mo)]}
    dump("baseconv.json", bc)

    # ---------------- a9: BSGS Hadamard (motivation/bsgs.py:39-52) -- module-level data
    import numpy as np
    M_blocks = [[int(x) for x in blk] for blk in mot_bsgs.M_blocks]
    v = [int(x) for x in mot_bsgs.v]
    y = [int(x) for x in mot_bsgs.diag_block_hadamard_matvec(mot_bsgs.M_blocks, mot_bsgs.v)]
    assert y == [int(x) for x in mot_bsgs.y_normal]
    np.random.seed(3)
    Ms = [np.random.randint(0, 1 << 20, size=8) for _ in range(4)]
    vs = np.random.randint(0, 1 << 20, size=32)
    ys = mot_bsgs.diag_block_hadamard_matvec(Ms, vs)
    dump("bsgs.json", {"mod": mot_bsgs.mod, "block_size": mot_bsgs.block_size, "k": mot_bsgs.k,
                       "sha256_M": sha_u64([x for b in M_blocks for x in b]), "sha256_v": sha_u64(v),
                       "numpy_seed": 0, "y_head": y[:32], "sha256_y": sha_u64([x & ((1 << 64) - 1) for x in y]),
                       "small": {"M": [[int(x) for x in b] for b in Ms], "v": [int(x) for x in vs],
                                 "y": [int(x) for x in ys]}})


if __name__ == "__main__":
    main()
