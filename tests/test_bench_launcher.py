"""bench.py as its own launcher: `python bench.py --gpus N` with no WORLD_SIZE starts
torch.distributed.run as a CHILD process, relays the ranks' JSON line and exits with their status; a world size that is not the
one asked for is refused.  Runs without a GPU: FHE_BENCH_DRYRUN makes the ranks report in before anything touches the device."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra, timeout=240):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_launcher_command_is_the_drivers_form():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "7"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [BENCH, "--gpus", "4", "--steps", "7"]


def test_plain_start_with_two_gpus_launches_two_ranks_and_relays_their_line():
    r = run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"FHE_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["dryrun"] is True


def test_a_failing_rank_fails_the_launcher():
    r = run(["--gpus", "2"], {"FHE_BENCH_DRYRUN": "1", "FHE_BENCH_DRYRUN_RC": "5"})
    assert r.returncode != 0


def test_world_size_other_than_asked_for_is_refused():
    r = run(["--gpus", "8"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "FHE_BENCH_DRYRUN": "1"})
    assert r.returncode == 2 and "refusing" in r.stderr


def test_traffic_floor_counts_the_launches_own_bytes():
    sys.path.insert(0, ROOT)
    import bench
    # config 5 key switch: 88 + 110 + 176 + 385 + 44 + 55 + 88 + 154 MiB of sweeps, + 22 MiB of sigma(c1) for a rotation
    us = bench.traffic_floor_us("rotate", 16, 44, 11, 4)
    mib = us * 1e-6 * bench.FABRIC_SUSTAINED_GBS * 1e9 / (1 << 20)
    assert abs(mib - (88 + 110 + 176 + 385 + 44 + 55 + 88 + 154 + 22)) < 1.0
    assert bench.traffic_floor_us("hmult", 17, 32, 8, 4) > bench.traffic_floor_us("keyswitch", 17, 32, 8, 4)
