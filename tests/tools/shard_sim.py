"""Debug / soak tool: runs every rank of a limb-sharded key switch in ONE process on one GPU (the all-gathers become slot
copies between the ranks' gather buffers) and compares the concatenated result with the oracle composite.
python tests/tools/shard_sim.py world logn L K dnum [bits]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fhe_reliability_gpu_amd as F  # noqa: E402
from fhe_reliability_gpu_amd._lib import check, lib, vp  # noqa: E402
from fhe_reliability_gpu_amd.dist import ks_layout, own_ct_rows, own_rows  # noqa: E402
from oracle.keyswitch_ref import keyswitch_ref  # noqa: E402

world, logn, L, K, dnum = (int(x) for x in sys.argv[1:6])
bits = int(sys.argv[6]) if len(sys.argv) > 6 else 50
N = 1 << logn
eng = F.Engine(0)
qs = F.create_moduli(N, [bits] * L + [61 if bits == 61 else 50] * K)
t = eng.tables(logn, qs)
rng = np.random.default_rng(1)
c = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs[:L]])
evk = np.stack([np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in qs]) for _ in range(2)]) for _ in range(dnum)])
to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy()).cuda()
P = lambda x: C.c_void_p(x.data_ptr() if x.numel() else 0)
lays = [ks_layout(L, K, world, r) for r in range(world)]
cmax, smax = lays[0]["cmax"], lays[0]["smax"]
g1 = [torch.zeros((world * cmax, N), dtype=torch.int64, device="cuda") for _ in range(world)]
g2 = [torch.zeros((world * 2 * smax, N), dtype=torch.int64, device="cuda") for _ in range(world)]
plans, cl, el = [], [], []
for r in range(world):
    h = vp()
    check(lib.fhe_keyswitch_create_sharded(eng._h, t._h, L, K, dnum, world, r, P(g1[r]), P(g2[r]), C.byref(h)))
    plans.append(h)
    cl.append(to(c[own_ct_rows(lays[r])]))
    el.append(to(evk[:, :, own_rows(lays[r])]))
for r in range(world):
    check(lib.fhe_keyswitch_shard_begin(eng._h, plans[r], P(cl[r]), None))
eng.sync()
for r in range(world):
    for s in range(world):
        g1[r][s * cmax:(s + 1) * cmax] = g1[s][s * cmax:(s + 1) * cmax]
torch.cuda.synchronize()
for r in range(world):
    check(lib.fhe_keyswitch_shard_inner(eng._h, plans[r], P(cl[r]), P(el[r]), None))
eng.sync()
for r in range(world):
    for s in range(world):
        g2[r][s * 2 * smax:(s + 1) * 2 * smax] = g2[s][s * 2 * smax:(s + 1) * 2 * smax]
torch.cuda.synchronize()
outs = []
for r in range(world):
    o0 = torch.zeros((lays[r]["cn"], N), dtype=torch.int64, device="cuda")
    o1 = torch.zeros_like(o0)
    check(lib.fhe_keyswitch_shard_finish(eng._h, plans[r], P(o0), P(o1), None, None, None))
    eng.sync()
    outs.append((o0.cpu().numpy().view(np.uint64), o1.cpu().numpy().view(np.uint64)))
w0, w1 = keyswitch_ref(c, evk, qs, L, K, dnum, logn)
bad = 0
for r in range(world):
    for h, w in ((0, w0), (1, w1)):
        for j, l in enumerate(own_ct_rows(lays[r])):
            ok = bool((outs[r][h][j] == w[l]).all())
            if not ok:
                bad += 1
                nz = np.nonzero(outs[r][h][j] != w[l])[0]
                print(f"rank {r} half {h} limb {l}: {nz.size} words differ, first at {nz[:4]}, lay {lays[r]}")
print("mismatching (rank, half, limb) rows:", bad)
sys.exit(1 if bad else 0)
