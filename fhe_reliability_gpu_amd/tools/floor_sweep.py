"""Measurement tool: forward NTT of batches around the Infinity Cache size (128 .. 256 MiB), one call per step, as one launch pair
("ntt_chunk_floor_mib" huge) and cut into sub-batches (floor 0).  A loop over the SAME batch keeps up to 256 MiB resident across steps, so
the whole-batch form looks better here than it would on data that arrives from HBM; the library cuts from 192 MiB.
python -m fhe_reliability_gpu_amd.tools.floor_sweep"""
import ctypes as C, sys, torch
import fhe_reliability_gpu_amd as F
from fhe_reliability_gpu_amd._lib import check, lib
N = 1 << 16
eng = F.Engine(0); q = F.create_moduli(N, [50]); t = eng.tables(16, q)
s = torch.cuda.Stream(); sp = C.c_void_p(s.cuda_stream)
def measure(data, reps=100):
    polys = data.shape[0]
    fn = lambda: check(lib.fhe_ntt_forward_batch(eng._h, C.c_void_p(data.data_ptr()), t._h, polys, 1, 0, sp))
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return ms * 1e3, 16.0 * N * polys / (ms * 1e-3) / 8e12
for polys in (256, 288, 320, 352, 384, 448, 512):
    data = torch.randint(0, q[0], (polys, N), device="cuda", dtype=torch.int64)
    out = [f"{polys} polys ({polys // 2} MiB):"]
    for floor in (4096, 0):      # never cut / cut whenever more than 1.5 pieces
        eng.set_option("ntt_chunk_floor_mib", floor)
        us, fr = measure(data)
        out.append(f"{'whole' if floor else 'cut  '} {us:7.1f} us {fr:.3f}")
    print("  ".join(out), flush=True)
    del data
