"""Measurement tool: one key switch per setting of "ks_fused" (0 = separate row pass + inner product, 1 = fused) at several shapes.  python -m fhe_reliability_gpu_amd.tools.ks_variants"""
import ctypes as C

import torch

import fhe_reliability_gpu_amd as F
from fhe_reliability_gpu_amd._lib import check, lib

eng = F.Engine(0)
st = torch.cuda.Stream()
sp = C.c_void_p(st.cuda_stream)
P = lambda x: C.c_void_p(x.data_ptr())
for logn, L, K, dnum in ((16, 16, 4, 4), (14, 4, 1, 4), (16, 44, 11, 4), (16, 44, 4, 11), (17, 32, 8, 4), (15, 24, 6, 4)):
    n = 1 << logn
    qs = F.create_moduli(n, [50] * (L + K))
    t = eng.tables(logn, qs)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    c = torch.randint(0, qs[0], (L, n), device="cuda", dtype=torch.int64)
    evk = torch.randint(0, qs[0], (dnum, 2, L + K, n), device="cuda", dtype=torch.int64)
    o0, o1 = torch.empty_like(c), torch.empty_like(c)
    ref = None
    for mode in (0, 1):
        eng.set_option("ks_fused", mode)
        call = lambda: check(lib.fhe_keyswitch_apply(eng._h, ks._h, P(o0), P(o1), P(c), P(evk), sp))
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(30):
            call()
        e1.record(st)
        torch.cuda.synchronize()
        same = True if ref is None else bool((o0 == ref[0]).all() and (o1 == ref[1]).all())
        if ref is None:
            ref = (o0.clone(), o1.clone())
        print(f"N=2^{logn} L={L} K={K} dnum={dnum} ks_fused={mode}: {e0.elapsed_time(e1) / 30 * 1e3:8.1f} us  identical={same}", flush=True)
    eng.set_option("ks_fused", -1)
    del ks, evk
