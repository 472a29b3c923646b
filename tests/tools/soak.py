"""Soak run: structured worst-case-ish inputs for the lazy ranges (all q-1, alternating 0 / q-1 at every period, q/2, sparse
spikes, random) through every size and both arithmetic paths; forward checked against the oracle, inverse by round trip.
    python tests/tools/soak.py [seconds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # repo root
import fhe_reliability_gpu_amd as F  # noqa: E402
from oracle import cport as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
eng = F.Engine(0)
rng = np.random.default_rng(12345)
t_end = time.time() + budget
cases = bad = 0
while time.time() < t_end:
    logn = int(rng.integers(1, 18))
    N = 1 << logn
    bits = int(rng.choice([30, 50, 61]))
    qs = F.create_moduli(max(N, 2), [bits, bits])
    t = eng.tables(logn, qs)
    pats = []
    for q in qs:
        kind = int(rng.integers(0, 6))
        if kind == 0:
            v = np.full(N, q - 1, dtype=np.uint64)
        elif kind == 1:
            period = 1 << int(rng.integers(0, logn + 1))
            v = np.where((np.arange(N) // period) % 2 == 0, 0, q - 1).astype(np.uint64)
        elif kind == 2:
            v = np.full(N, q // 2, dtype=np.uint64)
        elif kind == 3:
            v = np.zeros(N, dtype=np.uint64)
            v[rng.integers(0, N, size=max(1, N // 64))] = q - 1
        elif kind == 4:
            v = rng.integers(q - min(q, 1000), q, N, dtype=np.uint64)
        else:
            v = rng.integers(0, q, N, dtype=np.uint64)
        pats.append(v)
    data = np.stack(pats)[None]
    d = eng.upload(data)
    t.forward(d)
    fwd = d.download()
    for l, q in enumerate(qs):
        if not (fwd[0, l] == O.nwt_forward(data[0, l], q, O.root_powers(q, logn))).all():
            bad += 1
            print("MISMATCH forward", logn, bits, l)
    t.inverse(d)
    if not (d.download() == data).all():
        bad += 1
        print("MISMATCH round trip", logn, bits)
    # every 8th case: the fused negacyclic product and the checked transform on the same tables
    if cases % 8 == 0 and logn >= 5:
        b = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs])[None]
        da, db = eng.upload(data), eng.upload(b)
        t.polymul(da, da, db)
        got = da.download()
        for l, q in enumerate(qs):
            if not (got[0, l] == O.polymul_ntt(data[0, l], b[0, l], t.psi[l], q)).all():
                bad += 1
                print("MISMATCH polymul", logn, bits, l)
        ab = F.Abft(eng, t)
        d2 = eng.upload(data)
        if ab.forward_checked(d2).any() or not (d2.download() == fwd).all():
            bad += 1
            print("MISMATCH checked transform", logn, bits)
    cases += 1
print(f"soak: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
