"""Measurement tool: forward NTT at N = 2^16 on a batch that streams from HBM, for the engine's variants of the two-launch
transform: sub-batch size (ntt_chunk_mib), packed 50-bit hand-off (ntt_packed), streams.
python -m fhe_reliability_gpu_amd.tools.variant_sweep [polys]"""
import ctypes as C
import sys

import torch

import fhe_reliability_gpu_amd as F
from fhe_reliability_gpu_amd._lib import check, lib

N = 1 << 16
polys = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = F.Engine(0)
q = F.create_moduli(N, [50])
t = eng.tables(16, q)
data = torch.randint(0, q[0], (polys, N), device="cuda", dtype=torch.int64)


def run(nstr, reps=60):
    streams = [torch.cuda.Stream() for _ in range(nstr)]
    per = polys // nstr
    calls = [(C.c_void_p(data.data_ptr() + i * per * N * 8), per, C.c_void_p(s.cuda_stream)) for i, s in enumerate(streams)]

    def step():
        for ptr, cnt, sp in calls:
            check(lib.fhe_ntt_forward_batch(eng._h, ptr, t._h, cnt, 1, 0, sp))
    for _ in range(10):
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


modes = (("plain", 0, 0), ("packed", 1, 0), ("pingpong", 0, 1))   # (name, ntt_packed, ntt_pingpong)
if len(sys.argv) > 2:
    modes = tuple(m for m in modes if m[0] in sys.argv[2:])
for name, packed, pingpong in modes:
    for chunk in (0, 32, 64, 128, 256):
        for nstr in (1, 2, 4):
            eng.set_option("ntt_packed", packed)
            eng.set_option("ntt_pingpong", pingpong)
            eng.set_option("ntt_chunk_mib", chunk)
            ms = run(nstr)
            print(f"polys {polys} {name:9s} chunk_mib {chunk:4d} streams {nstr}: {ms * 1e3:8.1f} us/step  frac {16.0 * N * polys / (ms * 1e-3) / 8e12:.3f}", flush=True)
