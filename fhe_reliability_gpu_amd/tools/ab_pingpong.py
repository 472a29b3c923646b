"""A/B in one process, interleaved rounds: the bench's headline loop (N = 2^16, 1024 polynomials, two streams, 64 MiB sub-batches) with
the in-place hand-off and with the ping-pong hand-off.  python -m fhe_reliability_gpu_amd.tools.ab_pingpong"""
import ctypes as C
import statistics

import torch

import fhe_reliability_gpu_amd as F
from fhe_reliability_gpu_amd._lib import check, lib

N, polys = 1 << 16, 1024
eng = F.Engine(0)
q = F.create_moduli(N, [50])
t = eng.tables(16, q)
data = torch.randint(0, q[0], (polys, N), device="cuda", dtype=torch.int64)
streams = [torch.cuda.Stream() for _ in range(2)]
calls = [(C.c_void_p(data.data_ptr() + i * 512 * N * 8), 512, C.c_void_p(s.cuda_stream)) for i, s in enumerate(streams)]


def run(reps=100):
    def step():
        for ptr, cnt, sp in calls:
            check(lib.fhe_ntt_forward_batch(eng._h, ptr, t._h, cnt, 1, 0, sp))
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    ee = [torch.cuda.Event(enable_timing=True) for _ in streams]
    e0.record(streams[0])
    for _ in range(reps):
        step()
    for e, s in zip(ee, streams):
        e.record(s)
    torch.cuda.synchronize()
    return max(e0.elapsed_time(e) for e in ee) / reps


res = {0: [], 1: []}
for chunk in (64, 32):
    eng.set_option("ntt_chunk_mib", chunk)
    res = {0: [], 1: []}
    for rnd in range(6):
        for pp in (0, 1):
            eng.set_option("ntt_pingpong", pp)
            res[pp].append(run())
    for pp in (0, 1):
        ms = res[pp]
        print(f"chunk {chunk} MiB, pingpong {pp}: median {statistics.median(ms) * 1e3:.1f} us  min {min(ms) * 1e3:.1f}  max {max(ms) * 1e3:.1f}   frac(median) {16.0 * N * polys / (statistics.median(ms) * 1e-3) / 8e12:.3f}", flush=True)
