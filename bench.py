#!/usr/bin/env python3
"""bench.py -- forward negacyclic NTT throughput at N = 2^16 on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward NTT pass (nwt_2d_radix8_forward_inplace semantics,
reliability_test/ntt_test.cu:95) over one batch of residue polynomials that is already
resident in HBM.  Default workload = BASELINE.json configs[1]: N = 2^16, a single
50-bit prime (the reference's prime size, ntt_test.cu:44), batched over --polys
polynomials so that the 256 CUs are filled.  --limbs L switches to L distinct primes
per polynomial (configs[2] shape).  Multi-GPU: residue polynomials are independent, so
every rank transforms its own batch (weak scaling, no collective on the data path).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(algorithmic bytes 16*N per limb-NTT / HIP-event time vs 8 TB/s) and `cpu_baseline`
(the oracle's C port timed on one host core).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOGN = 16
N = 1 << LOGN
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--polys", type=int, default=256, help="residue polynomials per limb in the batch")
    ap.add_argument("--limbs", type=int, default=1, help="distinct RNS primes per polynomial")
    ap.add_argument("--bits", type=int, default=50, help="prime size (50 = reference; 61 = integer path)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--check", action="store_true", help="verify one limb against the oracle before timing")
    ap.add_argument("--mode", choices=["fused", "twopass"], default="twopass")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (other batch shapes)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the batch's polynomials are sharded over inside one GPU (each step = one call per stream)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    import ctypes as C

    import numpy as np

    import fhe_reliability_gpu_amd as F
    from fhe_reliability_gpu_amd._lib import check, lib

    eng = F.Engine(local_rank)
    eng.set_option("ntt_mode", 1 if args.mode == "fused" else 0)
    qs = F.create_moduli(N, [args.bits] * args.limbs)
    tables = eng.tables(LOGN, qs)
    units = args.polys * args.limbs

    # synthetic residue polynomials: uniform in [0, q_l), seeded per rank, resident in HBM
    g = torch.Generator(device="cuda")
    g.manual_seed(2025 + rank)
    data = torch.empty((args.polys, args.limbs, N), dtype=torch.int64, device="cuda")
    for l, q in enumerate(qs):
        data[:, l, :] = torch.randint(0, q, (args.polys, N), generator=g, device="cuda", dtype=torch.int64)
    pristine = data.clone()
    torch.cuda.synchronize()
    # a non-default torch stream: its handle is non-null, so the library launches on it and
    # torch.cuda.Event timestamps see the kernels
    stream = torch.cuda.Stream()
    sptr = C.c_void_p(stream.cuda_stream)
    assert stream.cuda_stream != 0
    dptr = C.c_void_p(data.data_ptr())

    def step():
        check(lib.fhe_ntt_forward_batch(eng._h, dptr, tables._h, args.polys, args.limbs, 0, sptr))

    # The headline loop shards the batch's polynomials over `--streams` HIP streams (north_star: "RNS limbs shard
    # one-per-stream and then one-per-GPU"): polynomials are independent, every stream transforms its own slab in
    # place, one library call per stream per step, no synchronisation between steps -- so one slab's column pass
    # runs under another slab's row pass and the kernel tails overlap (measured: -8 % against one stream).
    n_str = max(1, min(args.streams, args.polys))
    streams = [stream] + [torch.cuda.Stream() for _ in range(n_str - 1)]
    bounds = [(i * args.polys // n_str, (i + 1) * args.polys // n_str) for i in range(n_str)]
    shard_args = [(C.c_void_p(data.data_ptr() + lo * args.limbs * N * 8), hi - lo, C.c_void_p(st.cuda_stream))
                  for (lo, hi), st in zip(bounds, streams)]

    def step_sharded():
        for ptr, cnt, sp in shard_args:
            check(lib.fhe_ntt_forward_batch(eng._h, ptr, tables._h, cnt, args.limbs, 0, sp))

    if args.check and rank == 0:
        from oracle import cport as O
        step()
        torch.cuda.synchronize()
        eng.check()
        got = data[0, 0].cpu().numpy().view(np.uint64)
        want = O.nwt_forward(pristine[0, 0].cpu().numpy().view(np.uint64), qs[0], O.root_powers(qs[0], LOGN))
        assert (got == want).all(), "GPU forward NTT differs from the oracle"
        data.copy_(pristine)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The transform is in place; iterating it on its own output is still a full-rate
    # forward NTT of canonical residues (outputs are in [0, q)), so no reset inside the loop.
    for _ in range(args.warmup):
        step_sharded()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev_end = [torch.cuda.Event(enable_timing=True) for _ in streams]
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step_sharded()
    for e, st in zip(ev_end, streams):
        e.record(st)
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = max(ev0.elapsed_time(e) for e in ev_end)    # first launch of the region -> last stream to finish
    if world > 1:
        tmax = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())

    ntts = units * args.steps * world
    value = ntts / wall
    alg_bytes_per_step = 16.0 * N * units                # 16*N bytes per limb-NTT
    step_ms_dev = dev_ms / args.steps
    achieved = alg_bytes_per_step / (step_ms_dev * 1e-3) / 1e9

    result = {
        "metric": "forward negacyclic NTT/s at N=2^16 (limb-polynomials per second)",
        "value": value,
        "unit": "NTT/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64" if args.bits <= 50 else "u64",
        "data": "synthetic",
        "config": {
            "workload": f"N=2^16 forward NTT, {args.limbs} x {args.bits}-bit prime(s), batch of {args.polys} residue polynomials per GPU "
                        f"({args.polys * args.limbs * N * 8 >> 20} MiB), in place, resident on the device before the timed region",
            "log_n": LOGN, "limbs": args.limbs, "polys_per_gpu": args.polys, "prime_bits": args.bits,
            "parallelism": f"limb-polynomials sharded over {world} GPU(s), no collective; {n_str} stream(s) per GPU",
            "streams_per_gpu": n_str,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "kernel": ("k_ntt_fused (one launch per step)" if args.mode == "fused" else "column pass + row pass (two launches per slab per step)")
                      + "; achieved = 16*N*units bytes / HIP-event time of the step (first launch to the last stream's end, / steps)",
            "ms_per_step_device": step_ms_dev,
        },
    }

    # each kernel of the step timed on its own, inside real transforms (the two launches issued as two calls with an event
    # between them): the column pass and the row pass each move the batch once in and once out, i.e. 16*N bytes per
    # limb-polynomial per LAUNCH; the transform needs both
    if args.mode == "twopass":
        marks = []
        n_pairs = min(args.steps, 100)
        for i in range(n_pairs + 5):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            eng.set_option("ntt_only_pass", 0)
            ev[0].record(stream)
            step()
            ev[1].record(stream)
            eng.set_option("ntt_only_pass", 1)
            step()
            ev[2].record(stream)
            if i >= 5:
                marks.append(ev)
        eng.set_option("ntt_only_pass", -1)
        torch.cuda.synchronize()
        per_launch = {}
        for k, name in ((0, "column_pass"), (1, "row_pass")):
            ms = sum(e[k].elapsed_time(e[k + 1]) for e in marks) / len(marks)
            per_launch[name] = {"avg_launch_ms": ms, "GBps_moved": alg_bytes_per_step / (ms * 1e-3) / 1e9,
                                "frac_of_peak": alg_bytes_per_step / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        result["roofline"]["per_launch"] = per_launch

    # HBM/fabric bytes per step from the PMC passes committed under profiles/ (same command,
    # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes and as calibrated on this access
    # pattern, plus WRITE_SIZE); only quoted for the configuration it was collected on
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            tr = json.load(fh)
        key = f"{args.mode}:limbs={args.limbs}:polys={args.polys}:bits={args.bits}"
        if key in tr:
            result["roofline"]["traffic"] = tr[key]["bytes_per_step"]
            result["roofline"]["traffic_note"] = tr[key]["note"]

    # secondary measurements on rank 0 (not the headline): same kernels, other batch shapes
    if rank == 0 and not args.no_extras and world == 1:
        def rate(limbs, polys, bits, inverse=False, steps=300):
            q2 = F.create_moduli(N, [bits] * limbs)
            t2 = eng.tables(LOGN, q2)
            buf = torch.empty((polys, limbs, N), dtype=torch.int64, device="cuda")
            for l, q in enumerate(q2):
                buf[:, l, :] = torch.randint(0, q, (polys, N), generator=g, device="cuda", dtype=torch.int64)
            torch.cuda.synchronize()
            fn = lib.fhe_ntt_inverse_batch if inverse else lib.fhe_ntt_forward_batch
            call = lambda: check(fn(eng._h, C.c_void_p(buf.data_ptr()), t2._h, polys, limbs, 0, sptr))
            for _ in range(max(3, steps // 4)):
                call()
            torch.cuda.synchronize()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(stream)
            for _ in range(steps):
                call()
            a1.record(stream)
            torch.cuda.synchronize()
            ms = a0.elapsed_time(a1) / steps
            del buf
            return {"ntt_per_s": limbs * polys / (ms * 1e-3), "ms_per_step_device": ms,
                    "frac_of_hbm_roofline": 16.0 * N * limbs * polys / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        def pointwise_rates():
            # coefficient-wise products and base conversion at the configs[2] shape (N = 2^16, L = 16)
            L2, P2 = 16, 16
            q2 = F.create_moduli(N, [args.bits] * L2)
            t2 = eng.tables(LOGN, q2)
            mk = lambda: torch.randint(0, q2[0], (P2, L2, N), generator=g, device="cuda", dtype=torch.int64)
            a, b, c = mk(), mk(), mk()
            pa, pb, pc = (C.c_void_p(x.data_ptr()) for x in (a, b, c))
            def timed(fn, steps=200):
                for _ in range(50):
                    fn()
                torch.cuda.synchronize()
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record(stream)
                for _ in range(steps):
                    fn()
                a1.record(stream)
                torch.cuda.synchronize()
                return a0.elapsed_time(a1) / steps
            units2 = L2 * P2
            ms_mul = timed(lambda: check(lib.fhe_modmul(eng._h, pc, pa, pb, t2._h, P2, L2, 0, sptr)))
            # fhe_polymul uses its factors as scratch: restore them (untimed) before every timed call
            a_keep, b_keep = a.clone(), b.clone()
            pairs = []
            with torch.cuda.stream(stream):
                for i in range(13):
                    a.copy_(a_keep)
                    b.copy_(b_keep)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    check(lib.fhe_polymul(eng._h, pc, pa, pb, t2._h, P2, L2, 0, sptr))
                    e1.record(stream)
                    if i >= 3:
                        pairs.append((e0, e1))
            torch.cuda.synchronize()
            ms_poly = sum(x.elapsed_time(y) for x, y in pairs) / len(pairs)
            out = {
                "modmul_L16x16": {"ms_per_step_device": ms_mul, "GBps_algorithmic": 24.0 * N * units2 / (ms_mul * 1e-3) / 1e9,
                                  "frac_of_hbm_roofline": 24.0 * N * units2 / (ms_mul * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "polymul_L16x16 (configs[2]; bytes counted as NTT+NTT+modmul+INTT = 72N per limb)": {
                    "ms_per_step_device": ms_poly, "limb_polymul_per_s": units2 / (ms_poly * 1e-3),
                    "frac_of_hbm_roofline": 72.0 * N * units2 / (ms_poly * 1e-3) / 1e9 / HBM_PEAK_GBS},
            }
            m_in, k_out = 4, 4
            bc = F.BaseConv(eng, q2[:m_in], q2[m_in:m_in + k_out])
            src = torch.randint(0, q2[0], (m_in, 64 * N), generator=g, device="cuda", dtype=torch.int64)
            dst = torch.empty((k_out, 64 * N), dtype=torch.int64, device="cuda")
            ms_bc = timed(lambda: check(lib.fhe_baseconv_exact(eng._h, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), bc._h, 64 * N, sptr)))
            out["baseconv_exact_4to4_N=2^22"] = {"ms_per_step_device": ms_bc,
                                                 "frac_of_hbm_roofline": 8.0 * 64 * N * (m_in + k_out) / (ms_bc * 1e-3) / 1e9 / HBM_PEAK_GBS}
            return out
        result["also"] = {
            "inverse_same_batch": rate(args.limbs, args.polys, args.bits, inverse=True),
            "L16_distinct_primes_x16_polys (configs[2] shape)": rate(16, 16, args.bits),
            "hbm_streaming_1024_polys_512MiB (exceeds the 256 MiB Infinity Cache)": rate(1, 1024, args.bits, steps=60),
            "61bit_prime_integer_path": rate(1, args.polys, 61),
        }
        result["also"].update(pointwise_rates())

        def keyswitch_rates():
            # hybrid key switching: random NTT-form input and key, timing only
            out = {}
            for name, logn, L, K, dnum, reps in (("keyswitch_N=2^16_L16_K4_dnum4", 16, 16, 4, 4, 10),
                                                 ("rotate_N=2^14_L4_K1_dnum4 (shape of the reference's SEAL trace, 2543 us on its CPU)", 14, 4, 1, 4, 50)):
                n = 1 << logn
                qk = F.create_moduli(n, [args.bits] * (L + K))
                tk = eng.tables(logn, qk)
                ks = F.KeySwitch(eng, tk, L, K, dnum)
                c0 = torch.randint(0, qk[0], (L, n), generator=g, device="cuda", dtype=torch.int64)
                c1 = torch.randint(0, qk[0], (L, n), generator=g, device="cuda", dtype=torch.int64)
                evk = torch.randint(0, qk[0], (dnum, 2, L + K, n), generator=g, device="cuda", dtype=torch.int64)
                o0, o1 = torch.empty_like(c0), torch.empty_like(c0)
                P = lambda x: C.c_void_p(x.data_ptr())
                if name.startswith("rotate"):
                    call = lambda: check(lib.fhe_rotate(eng._h, ks._h, P(o0), P(o1), P(c0), P(c1), 3, P(evk), sptr))
                else:
                    call = lambda: check(lib.fhe_keyswitch_apply(eng._h, ks._h, P(o0), P(o1), P(c0), P(evk), sptr))
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record(stream)
                for _ in range(reps):
                    call()
                e1.record(stream)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / reps
                dev = e0.elapsed_time(e1) / reps
                # bytes every key switch must move at least once: input, key, two outputs
                alg = 8.0 * n * (L + dnum * 2 * (L + K) + 2 * L)
                out[name] = {"us_per_call_device": dev * 1e3, "us_per_call_wall": wall * 1e6,
                             "frac_of_hbm_roofline (input + key + outputs once)": alg / (dev * 1e-3) / 1e9 / HBM_PEAK_GBS}
                del ks
            return out
        result["also"].update(keyswitch_rates())

        def abft_rate():
            # forward transform with the weighted-checksum detector on the headline batch
            ab = F.Abft(eng, tables)
            flags = torch.zeros(args.polys * args.limbs, dtype=torch.int32, device="cuda")
            call = lambda: check(lib.fhe_ntt_forward_checked(eng._h, C.c_void_p(data.data_ptr()), tables._h, ab._h, args.polys, args.limbs, 0,
                                                             C.c_void_p(flags.data_ptr()), sptr))
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(20):
                call()
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            return {"abft_checked_forward_same_batch": {"ms_per_step_device": ms, "overhead_vs_unchecked": ms / step_ms_dev - 1.0,
                                                        "flags_raised": int(flags.sum().item())}}
        result["also"].update(abft_rate())

        def pcie_inclusive():
            # the boundary can hand over HOST buffers (ntt_test does): pinned host -> device, transform, device -> host
            host = torch.empty((args.polys, args.limbs, N), dtype=torch.int64).pin_memory()
            host.copy_(pristine.cpu())
            def call():
                with torch.cuda.stream(stream):
                    data.copy_(host, non_blocking=True)
                    step()
                    host.copy_(data, non_blocking=True)
            call()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 5
            return {"pcie_inclusive_same_batch (pinned host in, host out; not the headline)": {
                "ms_per_step_wall": dt * 1e3, "ntt_per_s": units / dt, "GBps_each_way": args.polys * args.limbs * N * 8 / (dt / 2) / 1e9}}
        result["also"].update(pcie_inclusive())

    if rank == 0 and world == 1 and not args.no_cpu:      # the CPU leg runs at N = 1 only
        from oracle import cport as O
        rp = O.root_powers(qs[0], LOGN)
        a = pristine[0, 0].cpu().numpy().view(np.uint64)
        O.nwt_forward(a, qs[0], rp)                      # warm
        reps, t = 0, time.perf_counter()
        while time.perf_counter() - t < 10.0:
            O.nwt_forward(a, qs[0], rp)
            reps += 1
        cpu_s = (time.perf_counter() - t) / reps
        # the reference's own CPU path is pure Python (motivation/ntt.py, rfhe_framewk/src/negaclic_ntt.py):
        # the oracle's Python restatement of it on the same limb, a few repetitions
        from oracle import pyport as PY
        a_list, rp_list = [int(x) for x in a], [int(x) for x in rp]
        py_reps, t = 0, time.perf_counter()
        while py_reps < 2 or (time.perf_counter() - t < 6.0 and py_reps < 16):
            PY.nwt_forward(a_list, qs[0], rp_list)
            py_reps += 1
        py_s = (time.perf_counter() - t) / py_reps
        result["cpu_baseline"] = {
            "value": 1.0 / cpu_s, "unit": "NTT/s", "cores": 1, "kind": "port",
            "sample": f"{reps} forward NTTs of one N=2^16 limb (oracle/fhe_oracle.c orc_nwt_forward, u128 mulmod, gcc -O3), ~10 s",
            "pure_python": {"value": 1.0 / py_s, "unit": "NTT/s", "cores": 1,
                            "sample": f"{py_reps} forward NTTs of the same limb with oracle/pyport.py nwt_forward (Python integers, the reference's CPU form)"},
        }
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
