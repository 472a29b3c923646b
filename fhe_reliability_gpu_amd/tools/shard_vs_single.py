"""One rank: the sharded phase path (dist.sharded_rotate / sharded_hmult) against the single-call composites, device time per call.
python3 fhe_reliability_gpu_amd/tools/shard_vs_single.py [rotate|hmult]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fhe_reliability_gpu_amd as F  # noqa: E402
from fhe_reliability_gpu_amd._lib import check, lib  # noqa: E402
from fhe_reliability_gpu_amd.dist import ShardedKeySwitch, sharded_hmult, sharded_rotate  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "rotate"
logn, L, K, dnum = (16, 44, 11, 4) if kind == "rotate" else (17, 32, 8, 4)
n = 1 << logn
eng = F.Engine(0)
qk = F.create_moduli(n, [50] * (L + K))
tk = eng.tables(logn, qk)
g = torch.Generator(device="cuda")
g.manual_seed(7)
mk = lambda *shape: torch.randint(0, qk[0], shape, generator=g, device="cuda", dtype=torch.int64)
c0, c1, b0, b1, gk = mk(L, n), mk(L, n), mk(L, n), mk(L, n), mk(dnum, 2, L + K, n)
plan = ShardedKeySwitch(eng, tk, L, K, dnum)
ks = F.KeySwitch(eng, tk, L, K, dnum)
stream = torch.cuda.Stream()
sptr = C.c_void_p(stream.cuda_stream)
P = lambda x: C.c_void_p(x.data_ptr())
lo = L - 1 if kind == "hmult" else L
o0, o1 = torch.empty((lo, n), dtype=torch.int64, device="cuda"), torch.empty((lo, n), dtype=torch.int64, device="cuda")
if kind == "rotate":
    shard = lambda: sharded_rotate(plan, c0, c1, 3, gk)
    single = lambda: check(lib.fhe_rotate(eng._h, ks._h, P(o0), P(o1), P(c0), P(c1), 3, P(gk), sptr))
else:
    shard = lambda: sharded_hmult(plan, c0, c1, b0, b1, gk, rescale=True)
    single = lambda: check(lib.fhe_hmult(eng._h, ks._h, P(o0), P(o1), P(c0), P(c1), P(b0), P(b1), P(gk), 1, sptr))
with torch.cuda.stream(stream):
    for name, fn in (("single", single), ("sharded", shard), ("single", single), ("sharded", shard)):
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{kind} {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call", flush=True)
