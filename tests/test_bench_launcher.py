"""bench.py's own rank launcher: `python bench.py --gpus N` with no launcher around it
must start N rank processes itself, pass rank 0's single JSON line through and return the ranks' return
code; the torch.distributed.run shape must keep working.  No GPU here, so the scan is stubbed
(--stub-scan: gloo, fabricated hits); everything around it is the real code of bench.py: environment
bootstrap, shard map (dcp_dist_shard), per-step gather through dcp_dist_merge_hits, max-over-ranks
timing, the one-line output."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                        "DCP_BENCH_SELF_LAUNCHED")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def _expected_hits(world, steps, warmup):
    return sum((i + r) % 5 for i in range(steps + warmup) for r in range(world))


@pytest.mark.timeout(300)
def test_self_launch_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-scan",
                        "--nprof", "999"], env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout  # ONE line, from rank 0 only
    out = lines[0]
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["stub"] is True and out["self_launched"] is True
    (b0, e0), (b1, e1) = out["shards"]
    assert b0 == 0 and e0 == b1 and e1 == 999 and 0 < e0 < 999  # the ranks' shards tile the DB
    assert out["hits_gathered"] == _expected_hits(2, 3, 1)


@pytest.mark.timeout(300)
def test_self_launch_propagates_a_rank_failure():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-scan",
                        "--nprof", "500", "--stub-fail-rank", "1", "--launch-grace", "2"],
                       env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert not [l for l in _json_lines(r.stdout) if not l.get("stub_failed")]  # no result line from a failed job


@pytest.mark.timeout(300)
def test_torchrun_shape_still_works():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                        "--steps", "2", "--warmup", "1", "--stub-scan", "--nprof", "700"],
                       env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    assert lines[0]["n_gpus"] == 2 and lines[0]["self_launched"] is False
    assert lines[0]["hits_gathered"] == _expected_hits(2, 2, 1)


@pytest.mark.timeout(120)
def test_single_rank_does_not_go_through_the_launcher():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--stub-scan",
                        "--nprof", "300"], env=dict(_clean_env(), MASTER_ADDR="127.0.0.1", MASTER_PORT="29577"),
                       capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["self_launched"] is False
