"""Where the wall time of a small key-switch test goes (set-up vs calls): python3 fhe_reliability_gpu_amd/tools/time_setup.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
T0 = time.time()
import fhe_reliability_gpu_amd as F  # noqa: E402

def lap(msg, t=[T0]):
    now = time.time()
    print(f"{now - t[0]:8.3f} s  {msg}", flush=True)
    t[0] = now

lap("import")
eng = F.default_engine()
lap("engine")
for logn, L, K, dnum in ((5, 3, 1, 3), (13, 4, 1, 4)):
    N = 1 << logn
    qs = F.create_moduli(N, [50] * (L + K))
    lap(f"moduli 2^{logn}")
    t = eng.tables(logn, qs)
    lap("tables")
    ks = F.KeySwitch(eng, t, L, K, dnum)
    lap("plan")
    rng = np.random.default_rng(1)
    c0 = eng.upload(np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]]))
    c1 = eng.upload(np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]]))
    key = eng.upload(np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(dnum)]))
    lap("uploads")
    o = ks.rotate(c0, c1, 3, key)
    eng.sync()
    lap("first rotate")
    o = ks.rotate(c0, c1, 3, key)
    eng.sync()
    lap("second rotate")
    pk = ks.prepare_galois_key(key, 3)
    outs = ks.rotate_hoisted(c0, c1, [3], [pk])
    eng.sync()
    lap("first hoisted")
    from oracle.keyswitch_ref import rotate_hoisted_ref, rotate_ref
    a0, a1, kk = c0.download(), c1.download(), key.download().reshape(dnum, 2, L + K, N)
    lap("downloads")
    rotate_ref(a0, a1, 3, kk, qs, L, K, dnum, logn)
    lap("oracle rotate_ref")
    rotate_hoisted_ref(a0, a1, 3, kk, qs, L, K, dnum, logn)
    lap("oracle rotate_hoisted_ref")
