"""Measurement tool: forward NTT of a batch larger than the Infinity Cache, issued as sub-batches of `chunk` polynomials
(column pass + row pass per sub-batch) so that a sub-batch's intermediate may stay in the 256 MiB Infinity Cache between
its two launches.  python -m fhe_reliability_gpu_amd.tools.chunk_sweep [polys] [streams]"""
import ctypes as C
import sys

import torch

import fhe_reliability_gpu_amd as F
from fhe_reliability_gpu_amd._lib import check, lib

N = 1 << 16
polys = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nstr = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = F.Engine(0)
q = F.create_moduli(N, [50])
t = eng.tables(16, q)
data = torch.randint(0, q[0], (polys, N), device="cuda", dtype=torch.int64)
streams = [torch.cuda.Stream() for _ in range(nstr)]
for chunk in (16, 32, 64, 128, 256, 512, polys):
    if chunk > polys:
        continue
    calls = []
    for i, lo in enumerate(range(0, polys, chunk)):
        calls.append((C.c_void_p(data.data_ptr() + lo * N * 8), min(chunk, polys - lo), C.c_void_p(streams[i % nstr].cuda_stream)))

    def step():
        for ptr, cnt, sp in calls:
            check(lib.fhe_ntt_forward_batch(eng._h, ptr, t._h, cnt, 1, 0, sp))
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    ee = [torch.cuda.Event(enable_timing=True) for _ in streams]
    e0.record(streams[0])
    reps = 50
    for _ in range(reps):
        step()
    for e, s in zip(ee, streams):
        e.record(s)
    torch.cuda.synchronize()
    ms = max(e0.elapsed_time(e) for e in ee) / reps
    print(f"polys {polys} chunk {chunk:5d} ({chunk // 2:4d} MiB) streams {nstr}: {ms * 1e3:8.1f} us/step  frac {16.0 * N * polys / (ms * 1e-3) / 8e12:.3f}", flush=True)
