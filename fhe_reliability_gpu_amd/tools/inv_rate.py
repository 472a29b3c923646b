"""Inverse / forward NTT rate at N = 2^16 on the headline batch (one library call per step): python3 fhe_reliability_gpu_amd/tools/inv_rate.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fhe_reliability_gpu_amd as F  # noqa: E402
from fhe_reliability_gpu_amd._lib import check, lib  # noqa: E402

logn, polys = 16, 1024
n = 1 << logn
eng = F.Engine(0)
stream = torch.cuda.Stream()
sptr = C.c_void_p(stream.cuda_stream)
for bits, limbs, pl in ((50, 1, polys), (50, 16, 64), (50, 1, 256), (61, 1, polys)):
    qs = F.create_moduli(n, [bits] * limbs)
    t = eng.tables(logn, qs)
    data = torch.randint(0, qs[0], (pl, limbs, n), device="cuda", dtype=torch.int64)
    for name, fn in (("forward", lib.fhe_ntt_forward_batch), ("inverse", lib.fhe_ntt_inverse_batch)):
        call = lambda: check(fn(eng._h, C.c_void_p(data.data_ptr()), t._h, pl, limbs, 0, sptr))
        for _ in range(30):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(100):
            call()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        print(f"{bits}-bit x{limbs} limbs x{pl} polys {name}: {ms:.4f} ms/step, {16.0 * n * limbs * pl / (ms * 1e-3) / 8e12:.4f} of the roofline", flush=True)
    del data
