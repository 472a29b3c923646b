#!/usr/bin/env python3
"""bench.py -- forward negacyclic NTT throughput at N = 2^16 on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one forward NTT pass (nwt_2d_radix8_forward_inplace semantics,
reliability_test/ntt_test.cu:95) over one batch of residue polynomials that is already
resident in HBM.  Default workload = BASELINE.json configs[1]: N = 2^16, a single
50-bit prime (the reference's prime size, ntt_test.cu:44), batched over --polys
polynomials.  The default batch is 1024 polynomials = 512 MiB: larger than the 256 MiB
Infinity Cache, so every sweep of a step is HBM traffic and `roofline.frac` is a
fraction of the HBM roofline in the literal sense (the 128 MiB batch that stays in the
Infinity Cache is reported beside it as `frac_cache_resident`).  --limbs L switches to L
distinct primes per polynomial (configs[2] shape).

Multi-GPU: residue polynomials are independent, so every rank transforms its own batch
(weak scaling, no collective on the data path, `value` = all ranks' NTTs / max time).  The
path's one exchange -- the base-conversion join of key switching -- is measured by the
strong-scaling legs `also.strong_scaling_config5` (BASELINE configs[4]: a rotation at
N = 2^16, L = 44) and `also.strong_scaling_config4` (configs[3]: hmult at N = 2^17, L = 32), limbs
sharded over the ranks, two RCCL all-gathers per key switch and one broadcast per rescale,
compute and joins timed separately.

Timing: an untimed device pre-warm of --prewarm-s seconds of the same step (a freshly leased GPU sits at idle clocks; with the
driver's --steps 20 the whole timed region would otherwise lie on the clock ramp: 0.31 instead of 0.37), then W untimed warm-up
steps, then exactly K timed steps between barrier + synchronize on both sides; `config.prewarm_steps_untimed` reports the count.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`
(algorithmic bytes 16*N per limb-NTT / HIP-event time vs 8 TB/s) and `cpu_baseline`
(the oracle's C port on one core and on all host cores, and the pure-Python form).
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


def _py_ntt_task(args):
    """One pure-Python forward NTT of a limb (worker of the all-cores CPU leg; imports only the oracle's Python port)."""
    a, q, rp, reps = args
    from oracle import pyport as PY
    t = time.perf_counter()
    for _ in range(reps):
        PY.nwt_forward(a, q, rp)
    return time.perf_counter() - t


def host_cores():
    """Host cores this process may use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a job a share of its
    cores, not the whole socket)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.5)))
            break
        except Exception:
            continue
    return n


FABRIC_SUSTAINED_GBS = 6000.0     # what the transform passes sustain through the fabric (profiles/r02_bench_kt.txt)


def traffic_floor_us(kind, logn, L, K, dnum):
    """Time the launches of one composite need for their OWN bytes at the rate the NTT passes sustain (6.0 TB/s): every launch's
    reads + writes in limb-sweeps of N words, summed.  Key switch (capi_keyswitch.cpp): opening INTT (two launches, in + out each),
    digit extension, the extended limbs' column pass, the fused row pass + inner product (extended limbs, own limbs, key, two sums),
    INTT of the special limbs, mod-down conversion, its column pass, the row pass with the tail (converted limbs, sums, addends, result)."""
    M, two = L + K, (2 if logn >= 13 else 1)        # two-launch sizes sweep a transform's limbs twice
    ext = dnum * M - L                              # extended limbs of all digits (a digit's own limbs are not extended)
    sweeps = 2 * two * L                            # opening INTT
    sweeps += L + ext                               # digit extension: reads the input, writes the extended limbs
    sweeps += (2 * ext if two == 2 else 0)          # column pass of the extended limbs
    sweeps += ext + L + 2 * dnum * M + 2 * M        # row pass + inner product
    addends = L if kind == "rotate" else 2 * L if kind == "hmult" else 0
    if kind == "hmult" and two == 2 and K >= 2:
        # fhe_hmult's own launch list at these shapes: the mod-down and the rescale share one forward transform (capi_keyswitch.cpp
        # ks_finish_rescale_*): the last limb rides along with the special limbs' INTT, y costs two word-wise launches of a few limbs
        R = L - 1
        sweeps += 6 + 2 * two * 2 * (K + 1)         # acc P^-1 + d on the last limb; INTT of the special limbs and of that limb
        sweeps += 2 * K + 2 * L + 6                 # mod-down conversion; y
        sweeps += 2 * 2 * R + 2                     # column pass of the converted limbs (+ y)
        sweeps += 2 * R + 2 * R + 2 * R + 2 * R     # row pass with the tail (converted limbs, sums, addends, result)
        sweeps += 7 * L                             # tensor product: four operands in, three parts out
        return sweeps * (8 << logn) / (FABRIC_SUSTAINED_GBS * 1e9) * 1e6
    sweeps += 2 * two * 2 * K                       # INTT of the special limbs, both halves
    sweeps += 2 * K + 2 * L                         # mod-down conversion
    sweeps += (2 * 2 * L if two == 2 else 0)        # column pass of the converted limbs
    sweeps += 2 * L + 2 * L + addends + 2 * L       # row pass with the tail
    if kind == "rotate":
        sweeps += L                                 # sigma(c1) kept for the inner product
    if kind == "hmult":
        sweeps += 7 * L                             # tensor product: four operands in, three parts out
        R = L - 1
        sweeps += 2 * two * 2                       # INTT of the two last limbs
        sweeps += (2 + 2 * R if two == 2 else 0)    # column pass of the residues (reads the broadcast limbs)
        sweeps += 2 * R + 2 * R + 2 * R             # row pass with (c - delta) / q_last
    return sweeps * (8 << logn) / (FABRIC_SUSTAINED_GBS * 1e9) * 1e6


def launcher_command(n_gpus, argv, port):
    """The command a plain `python bench.py --gpus N` turns itself into (the driver's own form of the N > 1 launch)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus, argv):
    import socket
    import subprocess
    with socket.socket() as sk:                    # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = subprocess.Popen(launcher_command(n_gpus, argv, port), env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:                        # relay: the ranks' stdout passes through, the JSON line is the last one that parses
        sys.stdout.write(out)
        sys.stdout.flush()
        if out.lstrip().startswith("{"):
            line = out
    rc = proc.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks exited cleanly but printed no JSON line", file=sys.stderr)
        return 3
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--prewarm-s", type=float, default=0.5, dest="prewarm_s", help="untimed device pre-warm before the W warm-up steps, seconds of wall time (0 = none)")
    ap.add_argument("--polys", type=int, default=1024, help="residue polynomials per limb in the batch (1024 = 512 MiB: HBM streaming)")
    ap.add_argument("--limbs", type=int, default=1, help="distinct RNS primes per polynomial")
    ap.add_argument("--bits", type=int, default=50, help="prime size (50 = reference; 61 = integer path)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--check", action="store_true", help="verify one limb against the oracle before timing")
    ap.add_argument("--mode", choices=["fused", "twopass"], default="twopass")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (other batch shapes, composites)")
    ap.add_argument("--chunk-mib", type=int, default=96, help="sub-batch size of each library call (0 = whole slab per launch pair)")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the batch's polynomials are sharded over inside one GPU (each step = one call per stream)")
    args = ap.parse_args()

    # ---- N > 1 started as a plain `python bench.py --gpus N`: become the launcher.  Nothing has touched the GPU yet (torch is not
    # even imported), so the ranks are CHILD processes of torch.distributed.run; their one JSON line is relayed and their status
    # becomes ours.  Started under the launcher already (WORLD_SIZE set), the world must be the one that was asked for.
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(self_launch(args.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: refusing to report a line for another world size", file=sys.stderr)
        sys.exit(2)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FHE_BENCH_DRYRUN"):
        # launcher rehearsal (tests/test_bench_launcher.py, no GPU): every rank reports in, rank 0 prints the line's skeleton
        if rank == 0:
            print(json.dumps({"metric": "dry run of the launcher", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dryrun": True}), flush=True)
        sys.exit(int(os.environ.get("FHE_BENCH_DRYRUN_RC", "0")) if rank == world - 1 else 0)

    import torch

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL; FHE_BENCH_BACKEND=gloo + FHE_BENCH_ONE_GPU=1 rehearse the multi-rank path on a one-GPU box (not a measurement)
        backend = os.environ.get("FHE_BENCH_BACKEND", "nccl")
        if os.environ.get("FHE_BENCH_ONE_GPU"):
            local_rank = 0
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    torch.cuda.set_device(local_rank)

    import ctypes as C

    import numpy as np

    import fhe_reliability_gpu_amd as F
    from fhe_reliability_gpu_amd._lib import check, lib

    eng = F.Engine(local_rank)
    eng.set_option("ntt_mode", 1 if args.mode == "fused" else 0)
    # each library call transforms its slab as sub-batches of this size, so that a sub-batch's second launch finds the first
    # one's output in the 256 MiB Infinity Cache (profiles/r02_split_sweep.txt: 96-112 MiB is the best size once the pieces' external side uses non-temporal accesses)
    eng.set_option("ntt_chunk_mib", args.chunk_mib)
    # the headline loop feeds --streams streams of its own: the library's own side stream ("ntt_split") stays off there
    eng.set_option("ntt_split", 0 if args.streams > 1 else -1)
    qs = F.create_moduli(N, [args.bits] * args.limbs)
    tables = eng.tables(LOGN, qs)
    units = args.polys * args.limbs

    # synthetic residue polynomials: uniform in [0, q_l), seeded per rank, resident in HBM
    g = torch.Generator(device="cuda")
    g.manual_seed(2025 + rank)
    data = torch.empty((args.polys, args.limbs, N), dtype=torch.int64, device="cuda")
    for l, q in enumerate(qs):
        data[:, l, :] = torch.randint(0, q, (args.polys, N), generator=g, device="cuda", dtype=torch.int64)
    first_limb = data[0, 0].clone()
    torch.cuda.synchronize()
    # a non-default torch stream: its handle is non-null, so the library launches on it and
    # torch.cuda.Event timestamps see the kernels
    stream = torch.cuda.Stream()
    sptr = C.c_void_p(stream.cuda_stream)
    assert stream.cuda_stream != 0
    P = lambda x: C.c_void_p(x.data_ptr())

    def step():
        check(lib.fhe_ntt_forward_batch(eng._h, P(data), tables._h, args.polys, args.limbs, 0, sptr))

    # The headline loop shards the batch's polynomials over `--streams` HIP streams (north_star: "RNS limbs shard
    # one-per-stream and then one-per-GPU"): polynomials are independent, every stream transforms its own slab in
    # place, one library call per stream per step, no synchronisation between steps -- so one slab's column pass
    # runs under another slab's row pass and the kernel tails overlap.
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
        keep = data[0, 0].clone()
        step()
        torch.cuda.synchronize()
        eng.check()
        got = data[0, 0].cpu().numpy().view(np.uint64)
        want = O.nwt_forward(keep.cpu().numpy().view(np.uint64), qs[0], O.root_powers(qs[0], LOGN))
        assert (got == want).all(), "GPU forward NTT differs from the oracle"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The transform is in place; iterating it on its own output is still a full-rate
    # forward NTT of canonical residues (outputs are in [0, q)), so no reset inside the loop.
    # Device pre-warm (not part of W or K): a freshly leased GPU sits at idle clocks and a timed region of a few milliseconds
    # (the driver's --steps 20) would be measured on the ramp -- 0.31 instead of 0.37 of the roofline.  The same step, untimed,
    # for a fixed wall time; then the W warm-up steps and the K timed steps as the contract says.
    prewarm_steps = 0
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_s:
        for _ in range(20):
            step_sharded()
        torch.cuda.synchronize()
        prewarm_steps += 20
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
    batch_mib = args.polys * args.limbs * N * 8 >> 20
    streaming = batch_mib > 256

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
            "workload": f"BASELINE configs[1]: N=2^16 forward NTT, {args.limbs} x {args.bits}-bit prime(s) ({'ONE prime for the whole batch -- `value` and `roofline.frac` are this single-prime batch; the RNS-shaped batch of 16 distinct primes x 64 polynomials is `roofline.frac_L16`' if args.limbs == 1 else 'distinct primes per polynomial'}), batch of {args.polys} residue polynomials per GPU "
                        f"({batch_mib} MiB, {'exceeds' if streaming else 'fits'} the 256 MiB Infinity Cache), in place, resident on the device "
                        f"before the timed region",
            "log_n": LOGN, "limbs": args.limbs, "polys_per_gpu": args.polys, "prime_bits": args.bits,
            "parallelism": f"limb-polynomials sharded over {world} GPU(s), no collective; {n_str} stream(s) per GPU, "
                           f"each call in sub-batches of {args.chunk_mib} MiB",
            "streams_per_gpu": n_str, "chunk_mib": args.chunk_mib, "prewarm_steps_untimed": prewarm_steps,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "regime": "hbm_streaming" if streaming else "infinity_cache_resident",
            "kernel": ("k_ntt_fused (one launch per step)" if args.mode == "fused" else "column pass + row pass (two launches per slab per step)")
                      + "; achieved = 16*N*units bytes / HIP-event time of the step (first launch to the last stream's end, / steps)",
            "ms_per_step_device": step_ms_dev,
        },
    }
    if streaming:
        result["roofline"]["frac_hbm_streaming"] = achieved / HBM_PEAK_GBS
    else:
        result["roofline"]["frac_cache_resident"] = achieved / HBM_PEAK_GBS

    def timed_loop(fn, steps, warm, on=stream):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record(on)
        for _ in range(steps):
            fn()
        a1.record(on)
        torch.cuda.synchronize()
        return a0.elapsed_time(a1) / steps

    # each kernel of the step timed on its own, inside real transforms on ONE stream (column pass, event, row pass, event, ...:
    # the sequence the transform runs; each launch moves the batch once in and once out, 16*N bytes per limb-polynomial per
    # LAUNCH; the transform needs both).  Always 200 transforms, whatever --steps is; the rocprofv3 kernel trace of this command
    # (profiles/) must agree with these averages.
    if args.mode == "twopass" and rank == 0:
        eng.set_option("ntt_stream", 1)        # the kernels the headline's sub-batches run: non-temporal accesses on the external side
        marks = []
        for i in range(220):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            eng.set_option("ntt_only_pass", 0)
            ev[0].record(stream)
            step()
            ev[1].record(stream)
            eng.set_option("ntt_only_pass", 1)
            step()
            ev[2].record(stream)
            if i >= 20:
                marks.append(ev)
        eng.set_option("ntt_only_pass", -1)
        eng.set_option("ntt_stream", -1)
        torch.cuda.synchronize()
        per_launch = {}
        for k, name in ((0, "column_pass"), (1, "row_pass")):
            ms = sum(e[k].elapsed_time(e[k + 1]) for e in marks) / len(marks)
            per_launch[name] = {"avg_launch_ms": ms, "launches_timed": len(marks), "GBps_moved": alg_bytes_per_step / (ms * 1e-3) / 1e9,
                                "frac_of_peak": alg_bytes_per_step / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        result["roofline"]["per_launch"] = per_launch
        result["roofline"]["per_launch_note"] = ("the headline's two kernels (non-temporal accesses on the external side) as whole-batch launches (no sub-batching) inside real transforms on one stream, 200 transforms, an event "
                                                 "between the two launches; each launch moves the batch once in and once out")
        # ONE library call per step on one caller stream: whole-batch launches, then the library's defaults (sub-batches
        # alternating between the caller's stream and the context's side stream)
        eng.set_option("ntt_chunk_mib", 0)
        one_stream_ms = timed_loop(step, 100, 10)
        result["roofline"]["ms_per_step_one_call_whole_batch_launches"] = one_stream_ms
        eng.set_option("ntt_chunk_mib", 96)
        eng.set_option("ntt_split", -1)
        one_call_ms = timed_loop(step, 200, 20)
        result["roofline"]["ms_per_step_one_call_library_defaults"] = one_call_ms
        result["roofline"]["frac_one_call_library_defaults"] = alg_bytes_per_step / (one_call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    else:
        one_stream_ms = one_call_ms = None

    # HBM/fabric bytes per launch pair from the PMC passes committed under profiles/ (rocprofv3 --pmc on this very command
    # line, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes and as calibrated on this access pattern, plus WRITE_SIZE):
    # a per-launch property of the kernels on this batch shape, quoted only for the configuration it was collected on
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            tr = json.load(fh)
        key = f"{args.mode}:limbs={args.limbs}:polys={args.polys}:bits={args.bits}:chunk={args.chunk_mib}:streams={n_str}"
        if key in tr:
            result["roofline"]["traffic"] = tr[key]["bytes_per_step"]
            result["roofline"]["traffic_note"] = tr[key]["note"]

    extras = not args.no_extras
    also = {}
    # everything below is one library call per step with the library's defaults (96 MiB sub-batches, side stream)
    eng.set_option("ntt_chunk_mib", 96)
    eng.set_option("ntt_split", -1)

    # (the sharded legs run BEFORE the single-GPU legs below: measured after them -- after their gigabytes of allocations and frees -- the
    # same one-rank rotation read 740 us instead of 375, while the identical calls in a fresh process read 348: an artefact of this
    # process's allocation history, not of the path)
    # ------------------------------------------------------------------ strong scaling: config 5 with the limbs sharded
    if extras:
        def strong_scaling(kind):
            # BASELINE configs[4] (kind "rotate": one rotation at N = 2^16, L = 44, K = 11, dnum = 4) and configs[3] (kind "hmult":
            # multiply + relinearize + rescale at N = 2^17, L = 32, K = 8, dnum = 4) with the limbs sharded over the ranks: the
            # fhe_keyswitch_shard_* / fhe_rescale_shard_* phases on each GPU, two in-place RCCL all-gathers per key switch and one
            # broadcast per rescale (dist.sharded_rotate / sharded_hmult).  Total work is fixed as the rank count grows; per call:
            # the compute phases and the joins from CUDA events on this rank's stream, max over ranks.
            from fhe_reliability_gpu_amd.dist import ShardedKeySwitch, ks_layout, sharded_hmult, sharded_rotate
            logn, L, K, dnum, reps = (16, 44, 11, 4, 60) if kind == "rotate" else (17, 32, 8, 4, 40)
            n = 1 << logn
            qk = F.create_moduli(n, [args.bits] * (L + K))
            tk = eng.tables(logn, qk)
            lay = ks_layout(L, K, world, rank)
            # every rank draws the SAME full operands (one seed) and keeps its own rows: rank 0 can then run the single-device
            # composite on the full operands and compare it with the gathered result of the sharded run
            gg = torch.Generator(device="cuda")
            gg.manual_seed(7)
            mk = lambda *shape: torch.randint(0, qk[0], shape, generator=gg, device="cuda", dtype=torch.int64)
            full = [mk(L, n) for _ in range(4)]
            gk_full = mk(dnum, 2, L + K, n)
            ct_rows = slice(lay["clo"], lay["clo"] + lay["cn"])
            key_rows = list(range(lay["clo"], lay["clo"] + lay["cn"])) + list(range(lay["slo"], lay["slo"] + lay["sn"]))
            c0, c1, b0, b1 = (x[ct_rows].contiguous() for x in full)
            gk = gk_full[:, :, key_rows, :].contiguous()
            if not (world > 1 and rank == 0):
                del full, gk_full
            # every rank finishes its own set-up before the first collective: a rank that failed here must not leave the others
            # waiting inside an all-gather
            err = None
            try:
                plan = ShardedKeySwitch(eng, tk, L, K, dnum)
            except Exception as ex:
                err = ex
            if world > 1:
                bad = torch.tensor([1 if err else 0], device="cuda", dtype=torch.int32)
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
                if int(bad.item()) and err is None:
                    err = RuntimeError("another rank could not build its sharded plan")
            if err:
                raise err
            if kind == "rotate":
                call = lambda tm=None: sharded_rotate(plan, c0, c1, 3, gk, timings=tm)
            else:
                call = lambda tm=None: sharded_hmult(plan, c0, c1, b0, b1, gk, rescale=True, timings=tm)
            equal = None
            with torch.cuda.stream(stream):
                if world > 1:
                    # untimed self-check of the first real multi-rank run: gather every rank's rows of the result on all ranks
                    # (padded slabs, one in-place all-gather per part) and compare on rank 0 with fhe_rotate / fhe_hmult on one device
                    from fhe_reliability_gpu_amd.dist import all_gather_slots
                    outs = call()
                    rows_max = lay["cmax"]
                    got = []
                    for part in outs:
                        buf = torch.zeros((world * rows_max, n), dtype=torch.int64, device="cuda")
                        buf[rank * rows_max:rank * rows_max + part.shape[0]] = part
                        all_gather_slots(buf, rows_max)
                        got.append(buf)
                    torch.cuda.synchronize()
                    if rank == 0:
                        ks1 = F.KeySwitch(eng, tk, L, K, dnum)
                        lo = L - 1 if kind == "hmult" else L
                        w0, w1 = torch.empty((lo, n), dtype=torch.int64, device="cuda"), torch.empty((lo, n), dtype=torch.int64, device="cuda")
                        if kind == "rotate":
                            check(lib.fhe_rotate(eng._h, ks1._h, P(w0), P(w1), P(full[0]), P(full[1]), 3, P(gk_full), sptr))
                        else:
                            check(lib.fhe_hmult(eng._h, ks1._h, P(w0), P(w1), P(full[0]), P(full[1]), P(full[2]), P(full[3]), P(gk_full), 1, sptr))
                        torch.cuda.synchronize()
                        equal = True
                        for r in range(world):
                            lr = ks_layout(L, K, world, r)
                            rows = [l for l in range(lr["clo"], lr["clo"] + lr["cn"]) if l < lo]
                            for want, have in ((w0, got[0]), (w1, got[1])):
                                if rows and not torch.equal(have[r * rows_max:r * rows_max + len(rows)], want[rows[0]:rows[0] + len(rows)]):
                                    equal = False
                        del ks1, w0, w1, full, gk_full
                    del got
                for _ in range(30):      # (a fixed count: every rank must issue the same collectives; enough to bring the clocks back up)
                    call()
                barrier()
                tm = {}
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                tw = time.perf_counter()
                e0.record()
                for _ in range(reps):
                    call(tm)
                e1.record()
                barrier()
                wall_c = (time.perf_counter() - tw) / reps
            evs = tm["events"]
            seg = [sum(e[i].elapsed_time(e[i + 1]) for e in evs) / len(evs) for i in range(5)]
            bc = 0.0
            if "rescale_events" in tm:
                bc = sum(e[1].elapsed_time(e[2]) for e in tm["rescale_events"]) / len(tm["rescale_events"])
            tot = e0.elapsed_time(e1) / reps
            vals = torch.tensor([tot - seg[1] - seg[3] - bc, seg[1], seg[3], bc, tot, wall_c * 1e3], device="cuda", dtype=torch.float64)
            if world > 1:
                dist.all_reduce(vals, op=dist.ReduceOp.MAX)
            comp, j1, j2, jb, tot, wl = (float(x) for x in vals.tolist())
            ntt_count = L + dnum * (L + K) - L + 2 * K + 2 * L + (2 + 2 * (L - 1) if kind == "hmult" else 0)
            name = "strong_scaling_config5" if kind == "rotate" else "strong_scaling_config4"
            what = "one rotation" if kind == "rotate" else "one hmult (multiply + relinearize + rescale)"
            out = {
                "workload": f"{what}, N=2^{logn}, L={L}, K={K}, dnum={dnum}, limbs sharded over {world} rank(s) "
                            f"(rank 0 owns {lay['cn']} ciphertext + {lay['sn']} special limbs)",
                "scaling": "strong", "n_gpus": world, "backend": (dist.get_backend() if world > 1 else "none (1 rank: no collective issued)"),
                "us_per_call_device": tot * 1e3, "us_per_call_wall": wl * 1e3,
                "us_compute_phases": comp * 1e3, "us_all_gather_1 (input, coefficient form)": j1 * 1e3,
                "us_all_gather_2 (special limbs of both halves)": j2 * 1e3,
                "bytes_all_gather_1_per_rank": plan.rows1 * n * 8, "bytes_all_gather_2_per_rank": plan.rows2 * n * 8,
                "limb_ntts_per_call": ntt_count, "limb_ntt_per_s": ntt_count / (tot * 1e-3)}
            if kind == "hmult":
                out["us_broadcast (last limbs of both parts)"] = jb * 1e3
                out["bytes_broadcast"] = 2 * n * 8
            if world > 1:
                out["sharded_equals_single"] = equal       # rank 0: gathered rows == fhe_rotate / fhe_hmult on one device, word for word
            out["traffic_floor_us"] = traffic_floor_us(kind, logn, L, K, dnum)
            del plan
            return {name: out}
        # A collective that never completes (a rank lost, a fabric fault) must not cost the headline: past the limit every rank
        # leaves, rank 0 with the JSON line it has (the leg marked as timed out) -- the one line the contract asks for either way.
        import threading
        done = threading.Event()

        def watchdog():
            if done.wait(float(os.environ.get("FHE_BENCH_STRONG_LIMIT_S", "240"))):
                return
            if rank == 0:
                also.setdefault("strong_scaling", {"error": "timed out: a sharded leg did not finish within the limit; the headline above is unaffected"})
                result["also"] = also
                print(json.dumps(result), flush=True)
            os._exit(4)      # the line is out; the status says that a leg hung (a collective that never completed must not read as success)
        def strong_scaling_hoisted():
            # eight HOISTED baby rotations of one sharded ciphertext at the config-5 shape (dist.sharded_rotate_hoisted): the input's
            # all-gather and the digit extension once, per rotation the inner product, ONE all-gather and the mod-down
            from fhe_reliability_gpu_amd.dist import ShardedKeySwitch, all_gather_slots, ks_layout, sharded_rotate_hoisted
            logn, L, K, dnum, n_rot, reps = 16, 44, 11, 4, 8, 12
            n = 1 << logn
            qk = F.create_moduli(n, [args.bits] * (L + K))
            tk = eng.tables(logn, qk)
            lay = ks_layout(L, K, world, rank)
            gg = torch.Generator(device="cuda")
            gg.manual_seed(11)
            mk = lambda *shape: torch.randint(0, qk[0], shape, generator=gg, device="cuda", dtype=torch.int64)
            c0_full, c1_full, gk_full = mk(L, n), mk(L, n), mk(dnum, 2, L + K, n)
            ct_rows = slice(lay["clo"], lay["clo"] + lay["cn"])
            key_rows = list(range(lay["clo"], lay["clo"] + lay["cn"])) + list(range(lay["slo"], lay["slo"] + lay["sn"]))
            c0, c1, gk = c0_full[ct_rows].contiguous(), c1_full[ct_rows].contiguous(), gk_full[:, :, key_rows, :].contiguous()
            elts = [pow(3, b + 1, 2 * n) for b in range(n_rot)]
            plan = ShardedKeySwitch(eng, tk, L, K, dnum)
            with torch.cuda.stream(stream):
                pk = plan.prepare_galois_key(gk, elts[0])          # (one prepared key stands for all eight: timing and self-check only)
                call = lambda k=n_rot: sharded_rotate_hoisted(plan, c0, c1, elts[:k], [pk] * k)
                equal = None
                if world > 1:
                    (h0, h1), = call(1)
                    got = []
                    for part in (h0, h1):
                        buf = torch.zeros((world * lay["cmax"], n), dtype=torch.int64, device="cuda")
                        buf[rank * lay["cmax"]:rank * lay["cmax"] + part.shape[0]] = part
                        all_gather_slots(buf, lay["cmax"])
                        got.append(buf)
                    torch.cuda.synchronize()
                    if rank == 0:
                        ks1 = F.KeySwitch(eng, tk, L, K, dnum)
                        pk1 = torch.empty_like(gk_full)
                        check(lib.fhe_galois_key_prepare(eng._h, ks1._h, P(pk1), P(gk_full), elts[0], sptr))
                        w0, w1 = torch.empty((L, n), dtype=torch.int64, device="cuda"), torch.empty((L, n), dtype=torch.int64, device="cuda")
                        a0, a1, kk, ge = (C.c_void_p * 1)(w0.data_ptr()), (C.c_void_p * 1)(w1.data_ptr()), (C.c_void_p * 1)(pk1.data_ptr()), (C.c_uint32 * 1)(elts[0])
                        check(lib.fhe_rotate_hoisted(eng._h, ks1._h, a0, a1, P(c0_full), P(c1_full), ge, kk, 1, sptr))
                        torch.cuda.synchronize()
                        equal = True
                        for r in range(world):
                            lr = ks_layout(L, K, world, r)
                            for want, have in ((w0, got[0]), (w1, got[1])):
                                if lr["cn"] and not torch.equal(have[r * lay["cmax"]:r * lay["cmax"] + lr["cn"]], want[lr["clo"]:lr["clo"] + lr["cn"]]):
                                    equal = False
                        del ks1, pk1
                for _ in range(4):
                    call()
                barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    call()
                e1.record()
                barrier()
            vals = torch.tensor([e0.elapsed_time(e1) / reps / n_rot], device="cuda", dtype=torch.float64)
            if world > 1:
                dist.all_reduce(vals, op=dist.ReduceOp.MAX)
            out = {"workload": f"{n_rot} hoisted rotations of one ciphertext, N=2^{logn}, L={L}, K={K}, dnum={dnum}, limbs sharded over {world} rank(s): "
                               f"one all-gather of the input for all of them, one all-gather of the special limbs per rotation",
                   "scaling": "strong", "n_gpus": world, "us_per_rotation_device": float(vals.item()) * 1e3,
                   "bytes_all_gather_per_rotation_per_rank": plan.rows2 * n * 8, "bytes_all_gather_shared_per_rank": plan.rows1 * n * 8}
            if world > 1:
                out["sharded_equals_single"] = equal
            del plan
            return {"strong_scaling_config5_hoisted": out}

        if world > 1:
            threading.Thread(target=watchdog, daemon=True).start()
        try:
            ss = strong_scaling_hoisted()
            if rank == 0:
                also.update(ss)
        except Exception as ex:
            if rank == 0:
                also["strong_scaling_config5_hoisted"] = {"error": repr(ex)}
        for kind in ("rotate", "hmult"):
            try:
                ss = strong_scaling(kind)
                if rank == 0:
                    also.update(ss)
            except Exception as ex:       # the headline must survive a failure of this leg; it is reported, not hidden
                if rank == 0:
                    also["strong_scaling_" + kind] = {"error": repr(ex)}
        done.set()
    # ------------------------------------------------------------------ secondary measurements (rank 0, one GPU)
    if rank == 0 and extras and world == 1 and not os.environ.get("FHE_BENCH_SKIP_SINGLE_GPU_LEGS"):
        def rate(limbs, polys, bits, inverse=False, steps=300):
            q2 = F.create_moduli(N, [bits] * limbs)
            t2 = eng.tables(LOGN, q2)
            buf = torch.empty((polys, limbs, N), dtype=torch.int64, device="cuda")
            for l, q in enumerate(q2):
                buf[:, l, :] = torch.randint(0, q, (polys, N), generator=g, device="cuda", dtype=torch.int64)
            torch.cuda.synchronize()
            fn = lib.fhe_ntt_inverse_batch if inverse else lib.fhe_ntt_forward_batch
            ms = timed_loop(lambda: check(fn(eng._h, P(buf), t2._h, polys, limbs, 0, sptr)), steps, max(3, steps // 4))
            del buf
            return {"ntt_per_s": limbs * polys / (ms * 1e-3), "ms_per_step_device": ms,
                    "frac_of_hbm_roofline": 16.0 * N * limbs * polys / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

        # the batch that stays inside the Infinity Cache (round 1's headline shape), same two-stream loop
        small = 256
        sm_args = [(C.c_void_p(data.data_ptr() + (i * small // n_str) * args.limbs * N * 8), (i + 1) * small // n_str - i * small // n_str,
                    C.c_void_p(st.cuda_stream)) for i, st in enumerate(streams)]

        def step_small():
            for ptr, cnt, sp in sm_args:
                check(lib.fhe_ntt_forward_batch(eng._h, ptr, tables._h, cnt, args.limbs, 0, sp))
        if args.polys >= small:
            for _ in range(200):
                step_small()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True)
            ee = [torch.cuda.Event(enable_timing=True) for _ in streams]
            e0.record(stream)
            for _ in range(1000):
                step_small()
            for e, st in zip(ee, streams):
                e.record(st)
            torch.cuda.synchronize()
            ms = max(e0.elapsed_time(e) for e in ee) / 1000
            fr = 16.0 * N * small * args.limbs / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            result["roofline"]["frac_cache_resident"] = fr
            also["cache_resident_256_polys_128MiB (fits the 256 MiB Infinity Cache; round 1's headline shape)"] = {
                "ntt_per_s": small * args.limbs / (ms * 1e-3), "ms_per_step_device": ms, "frac_of_hbm_roofline": fr,
                "note": "the batch is re-read from the Infinity Cache, not from HBM: a fabric-rate figure, not an HBM-rate one"}

        also["inverse_same_batch"] = rate(args.limbs, args.polys, args.bits, inverse=True, steps=100)
        also["L16_distinct_primes_x16_polys (configs[2] shape, 128 MiB)"] = rate(16, 16, args.bits)
        also["L16_distinct_primes_x64_polys (512 MiB, HBM streaming)"] = rate(16, 64, args.bits, steps=100)
        # the RNS-shaped batch of the same size (BASELINE's metric says "L RNS limbs"): sixteen distinct primes, sixteen twiddle tables
        result["roofline"]["frac_L16"] = also["L16_distinct_primes_x64_polys (512 MiB, HBM streaming)"]["frac_of_hbm_roofline"]
        also["61bit_prime_integer_path_same_batch"] = rate(1, args.polys, 61, steps=100)

        def pointwise_rates():
            # coefficient-wise products and base conversion at the configs[2] shape (N = 2^16, L = 16)
            L2, P2 = 16, 16
            q2 = F.create_moduli(N, [args.bits] * L2)
            t2 = eng.tables(LOGN, q2)
            mk = lambda: torch.randint(0, q2[0], (P2, L2, N), generator=g, device="cuda", dtype=torch.int64)
            a, b, c = mk(), mk(), mk()
            units2 = L2 * P2
            ms_mul = timed_loop(lambda: check(lib.fhe_modmul(eng._h, P(c), P(a), P(b), t2._h, P2, L2, 0, sptr)), 200, 50)
            # fhe_polymul uses its factors as scratch: restore them (untimed) before every timed call
            a_keep, b_keep = a.clone(), b.clone()
            pairs = []
            with torch.cuda.stream(stream):
                for i in range(13):
                    a.copy_(a_keep)
                    b.copy_(b_keep)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    check(lib.fhe_polymul(eng._h, P(c), P(a), P(b), t2._h, P2, L2, 0, sptr))
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
            ms_bc = timed_loop(lambda: check(lib.fhe_baseconv_exact(eng._h, P(dst), P(src), bc._h, 64 * N, sptr)), 200, 50)
            out["baseconv_exact_4to4_N=2^22"] = {"ms_per_step_device": ms_bc,
                                                 "frac_of_hbm_roofline": 8.0 * 64 * N * (m_in + k_out) / (ms_bc * 1e-3) / 1e9 / HBM_PEAK_GBS}
            return out
        also.update(pointwise_rates())

        def composite_rates():
            # the composites BASELINE configs[3] and [4] time, on one GPU (parity: tests/test_gpu_configs45.py), plus the two
            # round-1 shapes.  Random NTT-form inputs and keys; algorithmic bytes = every operand the caller hands over or gets
            # back, once: input parts + key + output parts (8 bytes per word).
            out = {}
            shapes = (
                ("keyswitch_N=2^16_L16_K4_dnum4", "keyswitch", 16, 16, 4, 4, 20),
                ("rotate_N=2^14_L4_K1_dnum4 (shape of the reference's SEAL trace, 2543 us on its CPU)", "rotate", 14, 4, 1, 4, 100),
                ("config5_rotate_N=2^16_L44_K11_dnum4 (BASELINE configs[4] on one GPU)", "rotate", 16, 44, 11, 4, 10),
                ("config5_rotate_N=2^16_L44_K4_dnum11 (BASELINE configs[4], dnum sweep of draw_dnum_rot_mul.py:63-65)", "rotate", 16, 44, 4, 11, 10),
                ("config4_hmult_N=2^17_L32_K8_dnum4 (BASELINE configs[3] on one GPU: multiply + relinearize + rescale)", "hmult", 17, 32, 8, 4, 10),
            )
            for name, kind, logn, L, K, dnum, reps in shapes:
                n = 1 << logn
                qk = F.create_moduli(n, [args.bits] * (L + K))
                tk = eng.tables(logn, qk)
                ks = F.KeySwitch(eng, tk, L, K, dnum)
                mk = lambda *shape: torch.randint(0, qk[0], shape, generator=g, device="cuda", dtype=torch.int64)
                c0, c1, b0, b1 = mk(L, n), mk(L, n), mk(L, n), mk(L, n)
                evk = mk(dnum, 2, L + K, n)
                lo = L - 1 if kind == "hmult" else L
                o0, o1 = torch.empty((lo, n), dtype=torch.int64, device="cuda"), torch.empty((lo, n), dtype=torch.int64, device="cuda")
                key_words = dnum * 2 * (L + K)
                if kind == "rotate":
                    call = lambda: check(lib.fhe_rotate(eng._h, ks._h, P(o0), P(o1), P(c0), P(c1), 3, P(evk), sptr))
                    words = 2 * L + key_words + 2 * L
                elif kind == "hmult":
                    call = lambda: check(lib.fhe_hmult(eng._h, ks._h, P(o0), P(o1), P(c0), P(c1), P(b0), P(b1), P(evk), 1, sptr))
                    words = 4 * L + key_words + 2 * (L - 1)
                else:
                    call = lambda: check(lib.fhe_keyswitch_apply(eng._h, ks._h, P(o0), P(o1), P(c0), P(evk), sptr))
                    words = L + key_words + 2 * L
                # warm for 60 ms of wall time (the set-up above leaves the device idle long enough for its clocks to drop), then
                # time at least `reps` calls and at least ~120 ms of them
                tw, done_calls = time.perf_counter(), 0
                while time.perf_counter() - tw < 0.06:
                    for _ in range(5):
                        call()
                    torch.cuda.synchronize()
                    done_calls += 5
                reps = max(reps, int(0.12 / ((time.perf_counter() - tw) / done_calls)))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                tw = time.perf_counter()
                e0.record(stream)
                for _ in range(reps):
                    call()
                e1.record(stream)
                torch.cuda.synchronize()
                wall_c = (time.perf_counter() - tw) / reps
                dev = e0.elapsed_time(e1) / reps
                alg = 8.0 * n * words
                # limb-NTT count of the composite: INTT of the input, the extended limbs of every digit, the special limbs
                # of both halves and the converted limbs of the mod-down (+ the rescale's for hmult)
                ntt_count = L + dnum * (L + K) - L + 2 * K + 2 * L + (2 + 2 * (L - 1) if kind == "hmult" else 0)
                out[name] = {"us_per_call_device": dev * 1e3, "us_per_call_wall": wall_c * 1e6, "limb_ntts_per_call": ntt_count,
                             "limb_ntt_per_s_inside": ntt_count / (dev * 1e-3),
                             "algorithmic_bytes": alg, "frac_of_hbm_roofline (inputs + key + outputs once)": alg / (dev * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic_floor_us": traffic_floor_us(kind, logn, L, K, dnum),
                             "frac_of_traffic_floor": traffic_floor_us(kind, logn, L, K, dnum) / (dev * 1e3)}
                del ks, evk
            return out
        also.update(composite_rates())

        def hoisted_rates():
            # BASELINE configs[4] names a BSGS rotation workload: the baby steps rotate ONE ciphertext by several elements
            # (profile_framewk/src/matmul_ckks.cpp:45-113).  8 baby rotations at N = 2^16, L = 44, K = 11, dnum = 4: hoisted (one
            # decomposition, keys in the un-rotated frame) against 8 calls of fhe_rotate; parity: tests/test_gpu_hoisted.py
            logn, L, K, dnum, n_rot = 16, 44, 11, 4, 8
            n = 1 << logn
            qk = F.create_moduli(n, [args.bits] * (L + K))
            tk = eng.tables(logn, qk)
            ks = F.KeySwitch(eng, tk, L, K, dnum)
            mk = lambda *shape: torch.randint(0, qk[0], shape, generator=g, device="cuda", dtype=torch.int64)
            c0, c1 = mk(L, n), mk(L, n)
            key = mk(dnum, 2, L + K, n)                      # one key buffer stands for all eight (timing only)
            outs0 = [torch.empty((L, n), dtype=torch.int64, device="cuda") for _ in range(n_rot)]
            outs1 = [torch.empty((L, n), dtype=torch.int64, device="cuda") for _ in range(n_rot)]
            elts = [pow(3, b + 1, 2 * n) for b in range(n_rot)]
            vp_t = C.c_void_p * n_rot
            a0, a1 = vp_t(*[x.data_ptr() for x in outs0]), vp_t(*[x.data_ptr() for x in outs1])
            kk, ge = vp_t(*[key.data_ptr()] * n_rot), (C.c_uint32 * n_rot)(*elts)
            hoisted = lambda: check(lib.fhe_rotate_hoisted(eng._h, ks._h, a0, a1, P(c0), P(c1), ge, kk, n_rot, sptr))

            def plain():
                for r in range(n_rot):
                    check(lib.fhe_rotate(eng._h, ks._h, P(outs0[r]), P(outs1[r]), P(c0), P(c1), elts[r], P(key), sptr))
            ms_h, ms_p = timed_loop(hoisted, 30, 10), timed_loop(plain, 30, 10)
            del ks, key
            return {"hoisted_8_baby_rotations_N=2^16_L44_K11_dnum4": {
                "us_per_rotation_hoisted": ms_h * 1e3 / n_rot, "us_per_rotation_plain": ms_p * 1e3 / n_rot,
                "rotations_per_s_hoisted": n_rot / (ms_h * 1e-3), "rotations_per_s_plain": n_rot / (ms_p * 1e-3),
                "speedup": ms_p / ms_h,
                "note": "fhe_rotate_hoisted: INTT + digit extension + column pass of the input once, per element the inner product and the mod-down"}}
        also.update(hoisted_rates())

        def bsgs_rate():
            # the whole baby-step / giant-step product (fhe_bsgs_matvec; profile_framewk/src/matmul_ckks.cpp:45-113) at the config-5 shape:
            # n1 = 8 baby steps (7 hoisted rotations), n2 = 4 giant steps (3 plain rotations), 32 diagonals; parity: tests/test_gpu_hoisted.py
            logn, L, K, dnum, n1, n2 = 16, 44, 11, 4, 8, 4
            n = 1 << logn
            qk = F.create_moduli(n, [args.bits] * (L + K))
            tk = eng.tables(logn, qk)
            ks = F.KeySwitch(eng, tk, L, K, dnum)
            mk = lambda *shape: torch.randint(0, qk[0], shape, generator=g, device="cuda", dtype=torch.int64)
            c0, c1, key = mk(L, n), mk(L, n), mk(dnum, 2, L + K, n)
            diags = mk(n2, n1, L, n)
            y0, y1 = torch.empty((L, n), dtype=torch.int64, device="cuda"), torch.empty((L, n), dtype=torch.int64, device="cuda")
            be = (C.c_uint32 * (n1 - 1))(*[pow(3, b, 2 * n) for b in range(1, n1)])
            ge = (C.c_uint32 * (n2 - 1))(*[pow(3, gg * n1, 2 * n) for gg in range(1, n2)])
            bk = (C.c_void_p * (n1 - 1))(*[key.data_ptr()] * (n1 - 1))
            gk = (C.c_void_p * (n2 - 1))(*[key.data_ptr()] * (n2 - 1))
            call = lambda: check(lib.fhe_bsgs_matvec(eng._h, ks._h, P(y0), P(y1), P(c0), P(c1), P(diags), n1, n2, be, bk, ge, gk, sptr))
            ms = timed_loop(call, 12, 4)
            del ks, key, diags
            return {"bsgs_matvec_N=2^16_L44_K11_dnum4_n1=8_n2=4": {
                "ms_per_product_device": ms, "rotations_per_product": n1 + n2 - 2, "diagonals": n1 * n2,
                "us_per_rotation_equivalent": ms * 1e3 / (n1 + n2 - 2),
                "note": "7 hoisted baby rotations (two streams), 4 inner sums of 8 diagonal products, 3 plain giant rotations accumulated"}}
        also.update(bsgs_rate())

        def fourstep_rate():
            # four_step_ntt (reliability_test/four_step_ntt_prot.py:71-109), MOD = 998244353, as a batch: two launches for all
            # vectors; 2^16 = 256 x 256 and 2^17 = 512 x 256 (the n1 != n2 case of BASELINE configs[3]); 16 N bytes per vector
            out = {}
            for n1, n2, n_vec in ((256, 256, 1), (256, 256, 256), (256, 256, 1024), (512, 256, 512)):
                nn = n1 * n2
                h = C.c_void_p()
                check(lib.fhe_fourstep_create(eng._h, n1, n2, 998244353, 3, C.byref(h)))
                src = torch.randint(0, 998244353, (n_vec, nn), generator=g, device="cuda", dtype=torch.int64)
                ms = timed_loop(lambda: check(lib.fhe_fourstep_ntt_batch(eng._h, P(src), P(src), h, n_vec, sptr)), 100, 10)
                out[f"fourstep_{n1}x{n2}_batch{n_vec}"] = {"us_per_call_device": ms * 1e3, "vectors_per_s": n_vec / (ms * 1e-3),
                                                         "frac_of_hbm_roofline (16N bytes per vector)": 16.0 * nn * n_vec / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                lib.fhe_fourstep_destroy(h)
                del src
            return out
        also.update(fourstep_rate())

        def abft_rate():
            # forward transform with the weighted-checksum detector on the headline batch, against the
            # unchecked transform on the SAME single stream and batch
            ab = F.Abft(eng, tables)
            flags = torch.zeros(args.polys * args.limbs, dtype=torch.int32, device="cuda")
            call = lambda: check(lib.fhe_ntt_forward_checked(eng._h, P(data), tables._h, ab._h, args.polys, args.limbs, 0, P(flags), sptr))
            # interleaved (unchecked, checked) x 3, so that both sides see the same clocks this late in the run
            pairs = [(timed_loop(step, 60, 10), timed_loop(call, 60, 10)) for _ in range(3)]
            base, ms = sum(p[0] for p in pairs) / 3, sum(p[1] for p in pairs) / 3
            return {"abft_checked_forward_same_batch": {"ms_per_step_device": ms, "unchecked_ms_same_stream": base,
                                                        "overhead_vs_unchecked_same_stream": ms / base - 1.0,
                                                        "note": "both sides: one library call per step on one caller stream, library defaults "
                                                                "(96 MiB sub-batches, side stream, ping-pong hand-off), measured "
                                                                "alternately three times in this leg",
                                                        "flags_raised": int(flags.sum().item())}}
        also.update(abft_rate())

        def pcie_inclusive():
            # the boundary can hand over HOST buffers (ntt_test does): pinned host -> device, transform, device -> host
            cnt = min(args.polys, 256)
            host = torch.empty((cnt, args.limbs, N), dtype=torch.int64).pin_memory()
            host.copy_(data[:cnt].cpu())
            dev = data[:cnt]

            def call():
                with torch.cuda.stream(stream):
                    dev.copy_(host, non_blocking=True)
                    check(lib.fhe_ntt_forward_batch(eng._h, P(dev), tables._h, cnt, args.limbs, 0, sptr))
                    host.copy_(dev, non_blocking=True)
            call()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                call()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 5
            return {"pcie_inclusive_256_polys (pinned host in, host out; not the headline)": {
                "ms_per_step_wall": dt * 1e3, "ntt_per_s": cnt * args.limbs / dt, "GBps_each_way": cnt * args.limbs * N * 8 / (dt / 2) / 1e9}}
        also.update(pcie_inclusive())

    if rank == 0 and also:
        result["also"] = also

    # ------------------------------------------------------------------ CPU baselines (rank 0, one GPU)
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import cport as O
        from oracle import pyport as PY
        cores = host_cores()
        rp = O.root_powers(qs[0], LOGN)
        a = first_limb.cpu().numpy().view(np.uint64)
        # (1) the C port on ONE core: the scalar restatement (u128 mulmod), ~5 s
        O.set_threads(1)
        O.nwt_forward(a, qs[0], rp)
        reps, t = 0, time.perf_counter()
        while time.perf_counter() - t < 5.0:
            O.nwt_forward(a, qs[0], rp)
            reps += 1
        c1 = reps / (time.perf_counter() - t)
        # (2) the C port on ALL host cores: independent limbs spread by OpenMP (cpu_cxx_allcores), ~5 s
        used = O.set_threads(cores)
        lim = max(16, 2 * used)
        batch = np.stack([a] * lim)
        qv, rpv = [qs[0]] * lim, np.stack([rp] * lim)
        O.nwt_forward_batch(batch, qv, rpv)
        rb, t = 0, time.perf_counter()
        while time.perf_counter() - t < 5.0:
            O.nwt_forward_batch(batch, qv, rpv)
            rb += 1
        call_c = rb * lim / (time.perf_counter() - t)
        # (3) the reference's own CPU form is pure Python (motivation/ntt.py, rfhe_framewk/src/negaclic_ntt.py): the
        # oracle's Python restatement on the same limb, one core, a few repetitions
        a_list, rp_list = [int(x) for x in a], [int(x) for x in rp]
        py_reps, t = 0, time.perf_counter()
        while py_reps < 2 or (time.perf_counter() - t < 5.0 and py_reps < 16):
            PY.nwt_forward(a_list, qs[0], rp_list)
            py_reps += 1
        py1 = py_reps / (time.perf_counter() - t)
        # (4) the same on all cores: multiprocessing.Pool, one independent limb per task (BASELINE.md cpu_python_allcores)
        py_all = None
        try:
            import multiprocessing as mp
            ctx = mp.get_context("spawn")       # workers never touch the GPU; spawn keeps them clear of this process's HIP state
            workers = min(cores, 64)             # one process per core, bounded (the GPU box caps processes per job)
            with ctx.Pool(workers) as pool:
                pool.map(_py_ntt_task, [(a_list[:256], qs[0], rp_list[:256], 1)] * workers)    # start the workers (untimed)
                t = time.perf_counter()
                pool.map(_py_ntt_task, [(a_list, qs[0], rp_list, 1)] * (2 * workers))
                py_all = 2 * workers / (time.perf_counter() - t)
        except Exception as ex:
            py_all = repr(ex)
        # (5) BASELINE configs[0]: N = 2^12, one 61-bit prime, cyclic NTT with motivation/ntt.py semantics (generator root)
        q61, root = 2305843009211596801, 37
        import random
        rnd = random.Random(12)
        v12 = [rnd.randrange(q61) for _ in range(1 << 12)]
        PY.ntt_cyclic(v12, q61, root)
        r12, t = 0, time.perf_counter()
        while time.perf_counter() - t < 2.0:
            PY.ntt_cyclic(v12, q61, root)
            r12 += 1
        c0_py = r12 / (time.perf_counter() - t)
        result["cpu_baseline"] = {
            "value": call_c, "unit": "NTT/s", "cores": used, "kind": "port",
            "sample": f"{rb} batches of {lim} forward NTTs of an N=2^16 limb, limbs spread over {used} host threads by OpenMP "
                      f"(oracle/fhe_oracle.c orc_nwt_forward_batch, u128 mulmod, gcc -O3), ~5 s",
            "one_core": {"value": c1, "unit": "NTT/s", "cores": 1, "sample": f"{reps} forward NTTs of one N=2^16 limb (orc_nwt_forward), ~5 s"},
            "pure_python_one_core": {"value": py1, "unit": "NTT/s", "cores": 1,
                                     "sample": f"{py_reps} forward NTTs of the same limb with oracle/pyport.py nwt_forward (Python integers, the reference's CPU form)"},
            "pure_python_all_cores": {"value": py_all, "unit": "NTT/s", "cores": min(cores, 64),
                                      "sample": f"{2 * min(cores, 64)} independent limbs over multiprocessing.Pool({min(cores, 64)}), same function"},
            "config0_N=2^12_61bit_cyclic_python": {"value": c0_py, "unit": "NTT/s", "cores": 1,
                                                   "sample": f"{r12} cyclic NTTs (oracle/pyport.py ntt_cyclic = motivation/ntt.py:8-32 semantics), q61={q61}, root={root}"},
            "host_cores": cores, "host_cores_note": "affinity mask capped by the cgroup CPU quota; os.cpu_count() = " + str(os.cpu_count()),
        }
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
