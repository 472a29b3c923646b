"""Measurement tool: non-temporal accesses on the external side of the two launches ("ntt_stream" 1) against plain ones, on the
headline batch (N = 2^16, one prime, 1024 polynomials = 512 MiB) and on a cache-resident one; A/B interleaved in one process.
python -m fhe_reliability_gpu_amd.tools.nt_sweep"""
import ctypes as C

import torch

import fhe_reliability_gpu_amd as F
from fhe_reliability_gpu_amd._lib import check, lib

N = 1 << 16
eng = F.Engine(0)
q = F.create_moduli(N, [50])
t = eng.tables(16, q)


def measure(data, streams, reps):
    polys = data.shape[0]
    per = polys // len(streams)
    calls = [(C.c_void_p(data.data_ptr() + i * per * N * 8), per, C.c_void_p(s.cuda_stream)) for i, s in enumerate(streams)]

    def step():
        for ptr, cnt, sp in calls:
            check(lib.fhe_ntt_forward_batch(eng._h, ptr, t._h, cnt, 1, 0, sp))
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    ee = [torch.cuda.Event(enable_timing=True) for _ in streams]
    e0.record(streams[0])
    for s in streams[1:]:
        s.wait_event(e0)
    for _ in range(reps):
        step()
    for e, s in zip(ee, streams):
        e.record(s)
    torch.cuda.synchronize()
    ms = max(e0.elapsed_time(e) for e in ee) / reps
    return ms, 16.0 * N * polys / (ms * 1e-3) / 8e12


big = torch.randint(0, q[0], (1024, N), device="cuda", dtype=torch.int64)
small = big[:256]
s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
for rnd in range(3):
    for nt in (0, 1):
        eng.set_option("ntt_stream", nt)
        eng.set_option("ntt_split", -1)
        ms, fr = measure(big, s2[:1], 60)
        out = [f"round {rnd} nt {nt}: one call 512 MiB {ms * 1e3:7.1f} us {fr:.3f}"]
        eng.set_option("ntt_split", 0)
        ms, fr = measure(big, s2, 60)
        out.append(f"two streams 512 MiB {ms * 1e3:7.1f} us {fr:.3f}")
        ms, fr = measure(small, s2, 300)
        out.append(f"two streams 128 MiB {ms * 1e3:7.1f} us {fr:.3f}")
        print("; ".join(out), flush=True)
