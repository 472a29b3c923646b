"""Per-rotation time of n hoisted rotations against n plain ones (device time, one call per step): python3 .../hoist_rate.py [n_rot]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fhe_reliability_gpu_amd as F  # noqa: E402
from fhe_reliability_gpu_amd._lib import check, lib  # noqa: E402

n_rot = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = F.Engine(0)
stream = torch.cuda.Stream()
sptr = C.c_void_p(stream.cuda_stream)
P = lambda x: C.c_void_p(x.data_ptr())
for logn, L, K, dnum in ((16, 44, 11, 4), (17, 32, 8, 4), (16, 16, 4, 4)):
    n = 1 << logn
    qk = F.create_moduli(n, [50] * (L + K))
    tk = eng.tables(logn, qk)
    ks = F.KeySwitch(eng, tk, L, K, dnum)
    mk = lambda *shape: torch.randint(0, qk[0], shape, device="cuda", dtype=torch.int64)
    c0, c1, key = mk(L, n), mk(L, n), mk(dnum, 2, L + K, n)
    o0 = [torch.empty((L, n), dtype=torch.int64, device="cuda") for _ in range(n_rot)]
    o1 = [torch.empty((L, n), dtype=torch.int64, device="cuda") for _ in range(n_rot)]
    elts = [pow(3, b + 1, 2 * n) for b in range(n_rot)]
    vp_t = C.c_void_p * n_rot
    a0, a1, kk, ge = vp_t(*[x.data_ptr() for x in o0]), vp_t(*[x.data_ptr() for x in o1]), vp_t(*[key.data_ptr()] * n_rot), (C.c_uint32 * n_rot)(*elts)
    hoisted = lambda: check(lib.fhe_rotate_hoisted(eng._h, ks._h, a0, a1, P(c0), P(c1), ge, kk, n_rot, sptr))

    def plain():
        for r in range(n_rot):
            check(lib.fhe_rotate(eng._h, ks._h, P(o0[r]), P(o1[r]), P(c0), P(c1), elts[r], P(key), sptr))
    for split in (0, -1):
        eng.set_option("ntt_split", split)
        for name, fn in (("hoisted", hoisted), ("plain", plain)):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(30):
                fn()
            e1.record(stream)
            torch.cuda.synchronize()
            print(f"N=2^{logn} L={L} K={K} dnum={dnum} side_stream={'off' if split == 0 else 'on'} {name}: {e0.elapsed_time(e1) / 30 / n_rot * 1e3:.1f} us per rotation", flush=True)
    del ks, key, o0, o1
