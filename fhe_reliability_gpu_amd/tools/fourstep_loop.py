"""Runs the batched four-step transform (256 x 256, MOD = 998244353) and, for comparison, the inverse negacyclic transform of a batch
of the same size in a loop, so that rocprofv3 can time their kernels side by side:
rocprofv3 --kernel-trace -d out -- python3 fhe_reliability_gpu_amd/tools/fourstep_loop.py [n_vec] [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import fhe_reliability_gpu_amd as F  # noqa: E402
from fhe_reliability_gpu_amd._lib import check, lib  # noqa: E402

n_vec = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, mod = 1 << 16, 998244353
eng = F.Engine(0)
h = C.c_void_p()
check(lib.fhe_fourstep_create(eng._h, 256, 256, mod, 3, C.byref(h)))
src = torch.randint(0, mod, (n_vec, N), device="cuda", dtype=torch.int64)
qs = F.create_moduli(N, [50])
t = eng.tables(16, qs)
neg = torch.randint(0, qs[0], (n_vec, N), device="cuda", dtype=torch.int64)
torch.cuda.synchronize()
P = lambda x: C.c_void_p(x.data_ptr())
for _ in range(reps):
    check(lib.fhe_fourstep_ntt_batch(eng._h, P(src), P(src), h, n_vec, None))
    check(lib.fhe_ntt_inverse_batch(eng._h, P(neg), t._h, n_vec, 1, 0, None))
eng.sync()
lib.fhe_fourstep_destroy(h)
