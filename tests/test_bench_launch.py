"""bench.py --gpus N starts its own N ranks (it used to run one rank and print n_gpus 1).
CPU leg: the rank start-up / rendezvous / MAX-over-ranks plumbing under gloo without any GPU work (the line says so:
value null).  GPU leg (-m gpu): the real bench with two ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one
device), asserting a two-rank line with verified outputs."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_bench_gpus2_starts_two_ranks_plumbing_only():
    p, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"BMI_BENCH_REHEARSE": "plumbing"}, 300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size_seen"] == 2 and rec["value"] is None and rec["max_over_ranks_check"] == 2.0


def test_bench_gpus8_starts_eight_ranks_plumbing_only():
    """the N = 8 launch the driver's scaling run uses (no GPU here: gloo, no PBS): eight ranks rendezvous, the line reports
    eight, the MAX reduction sees the last rank, and the ranks' contiguous parts tile the whole-job batch exactly"""
    p, lines = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "1000"], {"BMI_BENCH_REHEARSE": "plumbing"}, 600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["world_size_seen"] == 8 and rec["value"] is None and rec["max_over_ranks_check"] == 8.0
    ranges = rec["shard_ranges"]
    assert len(ranges) == 8 and ranges[0][0] == 0 and ranges[-1][1] == 8 * 1000
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and all(hi - lo == 1000 for lo, hi in ranges)


def test_bench_refuses_more_gpus_than_visible():
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("8 GPUs visible")
    p, lines = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], {}, 120)
    assert p.returncode != 0 and not lines        # never an n_gpus = 1 line for a --gpus 8 request


def test_bench_refuses_a_mismatched_launcher():
    env = {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_bench_gpus2_on_one_gpu_reports_two_ranks():
    p, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "1024", "--no-inverse",
                     "--no-cpu-baseline", "--no-second-field"],
                    {"BMI_BENCH_SHARE_DEVICE": "1", "BMI_BENCH_BACKEND": "gloo"}, 900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["world_size_seen"] == 2 and rec["value"] > 0
    assert rec["config"]["verified_decrypt"] is True and rec["scaling"] == "weak"
