"""Debug: hoisted rotations with one-limb digits (alpha = 1, N >= 2^13: the extension rides on the column pass's load) against the oracle,
single call and through the sharded phases with one rank.  python3 fhe_reliability_gpu_amd/tools/hoist_trivial_check.py [logn L K dnum]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fhe_reliability_gpu_amd as F  # noqa: E402
from fhe_reliability_gpu_amd.dist import ShardedKeySwitch, sharded_rotate_hoisted  # noqa: E402
from oracle.keyswitch_ref import rotate_hoisted_ref  # noqa: E402

logn, L, K, dnum = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (14, 7, 2, 7)
N = 1 << logn
eng = F.Engine(0)
qs = F.create_moduli(N, [50] * (L + K))
t = eng.tables(logn, qs)
rng = np.random.default_rng(3)
lim = lambda n: np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:n]])
c0, c1 = lim(L), lim(L)
key = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(dnum)])
elts = [3, 2 * N - 1]
ks = F.KeySwitch(eng, t, L, K, dnum)
pk = [ks.prepare_galois_key(eng.upload(key), e) for e in elts]
got = ks.rotate_hoisted(eng.upload(c0), eng.upload(c1), elts, pk)
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy()).cuda()
plan = ShardedKeySwitch(eng, t, L, K, dnum)
pkl = [plan.prepare_galois_key(to(key), e) for e in elts]
sh = sharded_rotate_hoisted(plan, to(c0), to(c1), elts, pkl)
torch.cuda.synchronize()
for i, e in enumerate(elts):
    w0, w1 = rotate_hoisted_ref(c0, c1, e, key, qs, L, K, dnum, logn)
    s0, s1 = sh[i][0].cpu().numpy().view(np.uint64), sh[i][1].cpu().numpy().view(np.uint64)
    print(f"elt {e}: single == oracle: {bool((got[i][0].download() == w0).all() and (got[i][1].download() == w1).all())}; "
          f"sharded(1 rank) == oracle: {bool((s0 == w0).all() and (s1 == w1).all())}; bad limbs sharded: {[j for j in range(L) if not (s0[j] == w0[j]).all()]} / {[j for j in range(L) if not (s1[j] == w1[j]).all()]}")
