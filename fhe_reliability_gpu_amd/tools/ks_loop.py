"""Runs fhe_keyswitch_apply (or rotate / hmult / eight hoisted rotations) in a loop (random NTT-form inputs) so that rocprofv3 can break a key switch down
by kernel:  rocprofv3 --kernel-trace --stats -d out -- python3 fhe_reliability_gpu_amd/tools/ks_loop.py 16 16 4 4 20 [rotate|hmult|hoisted]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fhe_reliability_gpu_amd as F  # noqa: E402

logn, L, K, dnum, reps = (int(x) for x in sys.argv[1:6])
rotate = len(sys.argv) > 6 and sys.argv[6] == "rotate"
hmult = len(sys.argv) > 6 and sys.argv[6] == "hmult"
hoisted = len(sys.argv) > 6 and sys.argv[6] == "hoisted"
n = 1 << logn
eng = F.Engine(0)
if os.environ.get("FHE_KS_FUSED"):
    eng.set_option("ks_fused", int(os.environ["FHE_KS_FUSED"]))
qs = F.create_moduli(n, [50] * (L + K))
t = eng.tables(logn, qs)
ks = F.KeySwitch(eng, t, L, K, dnum)
rng = np.random.default_rng(1)
rand = lambda shape: eng.upload(rng.integers(0, qs[0], size=shape, dtype=np.uint64))
c, evk = rand((L, n)), rand((dnum, 2, L + K, n))
c0 = rand((L, n))
if hoisted:
    elts = [pow(3, b + 1, 2 * n) for b in range(8)]
    pk = ks.prepare_galois_key(evk, 3)
for _ in range(reps):
    if hoisted:
        ks.rotate_hoisted(c0, c, elts, [pk] * 8)         # eight baby rotations of one ciphertext, one decomposition
    elif rotate:
        ks.rotate(c0, c, 3, evk)
    elif hmult:
        ks.hmult(c0, c, c, c0, evk)
    else:
        ks.apply(c, evk)
eng.sync()
