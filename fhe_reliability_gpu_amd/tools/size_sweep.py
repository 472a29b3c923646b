"""Measurement tool: forward and inverse negacyclic NTT at every size from 2^10 to 2^17 (one 50-bit prime), one library call per
step with the library's defaults, on a 128 MiB batch (stays in the Infinity Cache) and on a 512 MiB batch (streams from HBM).
python -m fhe_reliability_gpu_amd.tools.size_sweep"""
import ctypes as C

import torch

import fhe_reliability_gpu_amd as F
from fhe_reliability_gpu_amd._lib import check, lib

eng = F.Engine(0)
s = torch.cuda.Stream()
sp = C.c_void_p(s.cuda_stream)


def measure(fn, reps):
    for _ in range(max(5, reps // 10)):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps):
        fn()
    b.record(s)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for logn in range(10, 18):
    N = 1 << logn
    q = F.create_moduli(N, [50])
    t = eng.tables(logn, q)
    for mib in (128, 512):
        polys = (mib << 20) // (N * 8)
        data = torch.randint(0, q[0], (polys, N), device="cuda", dtype=torch.int64)
        out = [f"N=2^{logn:2d} {mib:3d} MiB ({polys:6d} polynomials):"]
        for inv, f in ((0, lib.fhe_ntt_forward_batch), (1, lib.fhe_ntt_inverse_batch)):
            ms = measure(lambda: check(f(eng._h, C.c_void_p(data.data_ptr()), t._h, polys, 1, 0, sp)), 200 if mib == 128 else 60)
            out.append(f"{'inverse' if inv else 'forward'} {ms * 1e3:7.1f} us {polys / (ms * 1e-3) / 1e6:8.2f} M NTT/s {16.0 * N * polys / (ms * 1e-3) / 8e12:.3f}")
        print("  ".join(out), flush=True)
        del data
