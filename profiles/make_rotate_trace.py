"""Writes a ROTATE trace at the shape of the reference's SEAL trace (N = 16384, L = 4 ciphertext primes,
one special prime, one digit per prime; profile_framewk/build/data/ckks/16384_4:466-539) in the line format
profile_framewk/build/{analyze_trace,sum_trace}.py parse.  Random NTT-form inputs: timing and format only.

    python profiles/make_rotate_trace.py gpurun_out/rotate_trace.txt      (on the GPU box)
    python <reference>/profile_framewk/build/sum_trace.py rotate_trace.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fhe_reliability_gpu_amd as F  # noqa: E402

logn, L, K, dnum = 14, 4, 1, 4
n = 1 << logn
eng = F.Engine(0)
qs = F.create_moduli(n, [50] * (L + K))
t = eng.tables(logn, qs)
ks = F.KeySwitch(eng, t, L, K, dnum)
rng = np.random.default_rng(1)
rand = lambda shape: eng.upload(rng.integers(0, qs[0], size=shape, dtype=np.uint64))
c0, c1, gk = rand((L, n)), rand((L, n)), rand((dnum, 2, L + K, n))
for _ in range(3):
    ks.rotate(c0, c1, 3, gk)           # warm: code objects, allocations
eng.sync()
eng.trace(True)
ks.rotate(c0, c1, 3, gk)
eng.trace(False)
with open(sys.argv[1], "w") as fh:
    fh.write(eng.trace_text())
print(eng.trace_text())
