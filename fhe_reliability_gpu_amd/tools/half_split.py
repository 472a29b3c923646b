"""Measurement tool (negative result): one call on a batch that FITS the Infinity Cache, run whole and cut into two / four pieces on the
caller's stream and the context's side stream.  The fork / join around every call costs 8-15 us, more than the overlap buys: whole 0.36,
two halves 0.33 at 128 MiB -- so only batches past the cache are cut.  python -m fhe_reliability_gpu_amd.tools.half_split"""
import ctypes as C, torch
import fhe_reliability_gpu_amd as F
from fhe_reliability_gpu_amd._lib import check, lib
N = 1 << 16
eng = F.Engine(0); q = F.create_moduli(N, [50]); t = eng.tables(16, q)
s = torch.cuda.Stream(); sp = C.c_void_p(s.cuda_stream)
def measure(data, reps=300):
    polys = data.shape[0]
    fn = lambda: check(lib.fhe_ntt_forward_batch(eng._h, C.c_void_p(data.data_ptr()), t._h, polys, 1, 0, sp))
    for _ in range(30): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(reps): fn()
    b.record(s); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return ms * 1e3, 16.0 * N * polys / (ms * 1e-3) / 8e12
for rnd in range(2):
    for polys in (32, 64, 128, 256, 352):
        data = torch.randint(0, q[0], (polys, N), device="cuda", dtype=torch.int64)
        out = [f"round {rnd} {polys} polys ({polys // 2} MiB):"]
        for name, opts in (("whole", dict(ntt_chunk_floor_mib=4096)),
                           ("two halves on two streams", dict(ntt_chunk_floor_mib=0, ntt_chunk_mib=max(1, polys // 4), ntt_pingpong=0, ntt_stream=0, ntt_split=1)),
                           ("four quarters", dict(ntt_chunk_floor_mib=0, ntt_chunk_mib=max(1, polys // 8), ntt_pingpong=0, ntt_stream=0, ntt_split=1))):
            for k, v in dict(ntt_chunk_floor_mib=192, ntt_chunk_mib=96, ntt_pingpong=-1, ntt_stream=-1, ntt_split=-1).items(): eng.set_option(k, v)
            for k, v in opts.items(): eng.set_option(k, v)
            us, fr = measure(data)
            out.append(f"{name} {us:6.1f} us {fr:.3f};")
        print(" ".join(out), flush=True)
        del data
