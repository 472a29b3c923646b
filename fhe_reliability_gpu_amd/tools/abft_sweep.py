"""Measurement tool: the checked forward transform (whole-transform and per-phase checks) on the headline batch against the unchecked
one, one call per step, with the pieces' non-temporal accesses and scratch hand-off on / off; A/B interleaved.
python -m fhe_reliability_gpu_amd.tools.abft_sweep"""
import ctypes as C

import torch

import fhe_reliability_gpu_amd as F
from fhe_reliability_gpu_amd._lib import check, lib

N, polys = 1 << 16, 1024
eng = F.Engine(0)
q = F.create_moduli(N, [50])
t = eng.tables(16, q)
ab = F.Abft(eng, t)
data = torch.randint(0, q[0], (polys, N), device="cuda", dtype=torch.int64)
flags = torch.zeros(polys * 3, dtype=torch.int32, device="cuda")
s = torch.cuda.Stream()
sp = C.c_void_p(s.cuda_stream)
P = lambda x: C.c_void_p(x.data_ptr())


def timed(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps):
        fn()
    b.record(s)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


plain = lambda: check(lib.fhe_ntt_forward_batch(eng._h, P(data), t._h, polys, 1, 0, sp))
chk = lambda: check(lib.fhe_ntt_forward_checked(eng._h, P(data), t._h, ab._h, polys, 1, 0, P(flags), sp))
phs = lambda: check(lib.fhe_ntt_forward_checked_phases(eng._h, P(data), t._h, ab._h, polys, 1, 0, P(flags), sp))
for rnd in range(2):
    for nt in (-1, 0):
        for pp in (-1, 1):
            eng.set_option("ntt_stream", nt)
            eng.set_option("ntt_pingpong", pp)
            u, c, p3 = timed(plain), timed(chk), timed(phs)
            print(f"round {rnd} ntt_stream {nt:2d} ntt_pingpong {pp:2d}: unchecked {u:7.1f} us, checked {c:7.1f} us (+{(c / u - 1) * 100:4.1f} %), per-phase {p3:7.1f} us (+{(p3 / u - 1) * 100:4.1f} %)", flush=True)
