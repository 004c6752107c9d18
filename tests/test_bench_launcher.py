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
                        "DCP_BENCH_SELF_LAUNCHED", "DCP_BENCH_PARENT_PID", "LD_PRELOAD")}
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


@pytest.mark.timeout(600)
def test_self_launch_eight_ranks_diagnostics():
    """The N = 8 shape nobody can rehearse on hardware here, on gloo with the stubbed scan: the
    shard map tiles the 20 000-profile C3 DB with a sum-M imbalance under 1 %, ONE JSON line, and that line carries
    what a first real run needs to be diagnosable: per-rank time / cells / shard bounds / gather time, the transport."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--stub-scan"],
                       env=_clean_env(), capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == 8 and out["self_launched"] is True
    mg = out["multi_gpu"]
    assert mg["ranks"] == 8 and len(mg["per_rank"]) == 8 and mg["shards_tile_the_db"] is True
    assert [p["rank"] for p in mg["per_rank"]] == list(range(8))
    assert mg["per_rank"][0]["shard"][0] == 0 and mg["per_rank"][7]["shard"][1] == 20000
    assert all(mg["per_rank"][i]["shard"] == out["shards"][i] for i in range(8))
    assert 0 <= mg["sum_m_imbalance"] < 0.01
    assert sum(p["shard_sum_m"] for p in mg["per_rank"]) == 3566452  # the C3 DB's sum M (BENCH_r03's workload)
    assert all(p["ms_per_step"] > 0 and p["gather_ms_per_step"] > 0 and p["cells"] > 0 for p in mg["per_rank"])
    assert mg["ms_per_step_min"] <= mg["ms_per_step_max"] and "gloo" in mg["transport"]
    assert out["hits_gathered"] == _expected_hits(8, 2, 1)


@pytest.mark.timeout(300)
def test_eight_rank_failure_propagates():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--stub-scan",
                        "--nprof", "2000", "--stub-fail-rank", "5", "--launch-grace", "2"],
                       env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert not _json_lines(r.stdout)


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    # a reaped-by-init zombie is gone for our purposes
    try:
        return open(f"/proc/{pid}/stat").read().split(") ")[1][0] != "Z"
    except OSError:
        return False


@pytest.mark.timeout(300)
def test_killing_the_launcher_leaves_no_rank_behind(tmp_path):
    """`timeout -k` (SIGTERM) on the self-launching parent used to orphan the ranks, blocked in a collective
    and holding their GPUs.  The parent now terminates exactly the PIDs it started; SIGKILL on the parent is covered by
    PR_SET_PDEATHSIG in the ranks."""
    import signal
    import time

    for sig in (signal.SIGTERM, signal.SIGKILL):
        d = tmp_path / f"pids{int(sig)}"
        d.mkdir()
        p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "100000", "--warmup", "0", "--stub-scan",
                              "--nprof", "300", "--stub-sleep", "0.05"],
                             env=dict(_clean_env(), DCP_BENCH_STUB_PIDDIR=str(d)),
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        t0 = time.time()
        while len(list(d.glob("rank*.pid"))) < 2 and time.time() - t0 < 120:
            time.sleep(0.1)
        pids = [int(f.read_text()) for f in d.glob("rank*.pid")]
        assert len(pids) == 2 and all(_alive(x) for x in pids)
        time.sleep(0.5)  # let them get into their step loop (and the prctl)
        p.send_signal(sig)
        rc = p.wait(60)
        assert rc != 0
        t0 = time.time()
        while any(_alive(x) for x in pids) and time.time() - t0 < 30:
            time.sleep(0.1)
        assert not any(_alive(x) for x in pids), (sig, pids)
        if sig == signal.SIGTERM:
            assert rc == 128 + int(signal.SIGTERM)


@pytest.mark.timeout(60)
def test_self_launch_is_refused_under_a_profiler_preload():
    """rocprofv3's preload initialises the GPU in the parent: starting rank children from there is the forbidden
    exec-after-HIP-init hop, so bench.py says so and exits non-zero instead."""
    env = dict(_clean_env(), ROCPROF_OUTPUT_PATH="/tmp/x")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-scan"], env=env, capture_output=True, text=True,
                       timeout=50)
    assert r.returncode == 2 and "profiler" in r.stderr and not _json_lines(r.stdout)
