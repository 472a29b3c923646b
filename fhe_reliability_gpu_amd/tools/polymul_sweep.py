"""Measurement tool: negacyclic product of 16 primes x 16 polynomials at N = 2^16 (BASELINE configs[2]: operands + result = 384 MiB)
as whole-batch launches ("ntt_chunk_mib" 0) and cut into pieces of several sizes, with and without the side stream; A/B interleaved.
python -m fhe_reliability_gpu_amd.tools.polymul_sweep [polys]"""
import ctypes as C
import sys

import torch

import fhe_reliability_gpu_amd as F
from fhe_reliability_gpu_amd._lib import check, lib

N, L = 1 << 16, 16
polys = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eng = F.Engine(0)
qs = F.create_moduli(N, [50] * L)
t = eng.tables(16, qs)
g = torch.Generator(device="cuda")
g.manual_seed(3)
a0 = torch.randint(0, qs[0], (polys, L, N), generator=g, device="cuda", dtype=torch.int64)
b0 = torch.randint(0, qs[0], (polys, L, N), generator=g, device="cuda", dtype=torch.int64)
a, b, c = a0.clone(), b0.clone(), torch.empty_like(a0)
s = torch.cuda.Stream()
P = lambda x: C.c_void_p(x.data_ptr())


def measure(reps=40):
    def step():
        check(lib.fhe_polymul(eng._h, P(c), P(a), P(b), t._h, polys, L, 0, C.c_void_p(s.cuda_stream)))
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps):
        step()
    e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rnd in range(2):
    for chunk, split in ((0, 0), (64, 0), (64, 1), (128, 0), (128, 1), (192, 1), (256, 1)):
        eng.set_option("ntt_chunk_mib", chunk)
        eng.set_option("ntt_chunk_floor_mib", 0 if chunk else 192)
        eng.set_option("ntt_split", split)
        ms = measure()
        print(f"round {rnd} polys {polys} chunk {chunk:4d} MiB split {split}: {ms * 1e3:8.1f} us  frac {72.0 * N * L * polys / (ms * 1e-3) / 8e12:.3f}", flush=True)
