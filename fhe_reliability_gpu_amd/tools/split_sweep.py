"""Measurement tool: one forward-NTT call per step on a batch that streams from HBM, with the library cutting it into sub-batches
("ntt_chunk_mib") and, with "ntt_split" 1, alternating the sub-batches between the caller's stream and a side stream of its own.
A/B interleaved in one process.  python -m fhe_reliability_gpu_amd.tools.split_sweep [polys] [caller streams]"""
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
per = polys // nstr
calls = [(C.c_void_p(data.data_ptr() + i * per * N * 8), per, C.c_void_p(s.cuda_stream)) for i, s in enumerate(streams)]


def measure(reps=60):
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
    return max(e0.elapsed_time(e) for e in ee) / reps


for rnd in range(2):
    for chunk in (80, 96, 112, 128, 160):
        for split in (0, 1):
            eng.set_option("ntt_chunk_mib", chunk)
            eng.set_option("ntt_split", split)
            ms = measure()
            print(f"round {rnd} polys {polys} caller streams {nstr} chunk {chunk:4d} MiB split {split}: {ms * 1e3:8.1f} us/step  "
                  f"frac {16.0 * N * polys / (ms * 1e-3) / 8e12:.3f}", flush=True)
