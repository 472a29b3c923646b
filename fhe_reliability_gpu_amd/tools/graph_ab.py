"""Measurement tool: a key switch issued call by call against the same launch chain captured once into a hipGraph and replayed
(the library's launches are capturable: tests/test_gpu_keyswitch.py).  python -m fhe_reliability_gpu_amd.tools.graph_ab"""
import ctypes as C

import torch

import fhe_reliability_gpu_amd as F
from fhe_reliability_gpu_amd._lib import check, lib

eng = F.Engine(0)
P = lambda x: C.c_void_p(x.data_ptr())


def timed(fn, stream, reps):
    for _ in range(max(3, reps // 10)):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps):
        fn()
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for logn, L, K, dnum in ((14, 4, 1, 4), (13, 4, 1, 4), (16, 16, 4, 4), (16, 44, 11, 4)):
    n = 1 << logn
    qs = F.create_moduli(n, [50] * (L + K))
    t = eng.tables(logn, qs)
    ks = F.KeySwitch(eng, t, L, K, dnum)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    c = torch.randint(0, qs[0], (L, n), generator=g, device="cuda", dtype=torch.int64)
    evk = torch.randint(0, qs[0], (dnum, 2, L + K, n), generator=g, device="cuda", dtype=torch.int64)
    o0, o1 = torch.empty_like(c), torch.empty_like(c)
    s = torch.cuda.Stream()

    def call(st=s):
        check(lib.fhe_keyswitch_apply(eng._h, ks._h, P(o0), P(o1), P(c), P(evk), C.c_void_p(st.cuda_stream)))

    torch.cuda.synchronize()
    eager = timed(call, s, 300)
    out = [f"N=2^{logn} L={L} K={K} dnum={dnum}: call by call {eager:7.1f} us"]
    for per_graph in (1, 8):
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=s):
            for _ in range(per_graph):
                call(s)
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            us = timed(graph.replay, s, 300 // per_graph) / per_graph
        out.append(f"graph of {per_graph}: {us:7.1f} us per key switch")
    print("; ".join(out), flush=True)
