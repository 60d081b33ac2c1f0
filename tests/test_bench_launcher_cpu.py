"""bench.py's own N > 1 launcher (`python bench.py --gpus N`, WORLD_SIZE unset) and the torch.distributed.run route, driven on the CPU with
the stub step (MUDPT_BENCH_STUB=1: gloo, a CPU bucket of the real size): process fan-out, rendezvous on 127.0.0.1, max-over-ranks timing
and the one JSON line on rank 0.  The real step needs MI355Xs; the driver's SCALE run is its hardware test."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"MUDPT_BENCH_STUB": "1"}, **(extra_env or {}))
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2, 3])
def test_bare_command_launches_its_own_ranks(n):
    out = run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "3", "--warmup", "1"])
    assert out["n_gpus"] == n and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 256 * n and out["collective"]["world_size"] == n
    assert out["value"] > 0 and out["data"] == "stub"


def test_torch_distributed_run_route_still_works():
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29731", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["collective"]["backend"] == "gloo"


def test_world_size_mismatch_is_refused():
    env = {k: v for k, v in os.environ.items()}
    env.update(WORLD_SIZE="2", RANK="0", MUDPT_BENCH_STUB="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "must agree" in (r.stdout + r.stderr)


def test_a_failing_rank_ends_the_job():
    """A rank that dies must not leave its peers (and the parent) hanging in the rendezvous / a collective: the parent ends the others and
    returns the failing rank's exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MUDPT_BENCH_STUB="1", MUDPT_BENCH_STUB_FAIL_RANK="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)


def test_a_hanging_rank_hits_the_deadline():
    """A rank that is alive but never reaches the rendezvous (or sits in a collective) is not a dead rank: the parent's own deadline
    (MUDPT_BENCH_DEADLINE_S) terminates the children it started, names the ranks still running on stderr and exits 124; the ranks' own
    stderr arrives rank-prefixed.  The other rank's rendezvous timeout is longer than the deadline here, so the deadline is what fires."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MUDPT_BENCH_STUB="1", MUDPT_BENCH_STUB_HANG_RANK="1", MUDPT_BENCH_DEADLINE_S="6", MUDPT_DIST_TIMEOUT_S="300")
    t0 = time.time()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 124, (r.returncode, r.stdout, r.stderr)
    assert time.time() - t0 < 60
    assert "deadline of 6 s passed" in r.stderr and "still running" in r.stderr
    assert "[rank 1] stub rank: sleeping past the deadline" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no JSON line: the job did not finish


def test_a_rendezvous_timeout_ends_a_rank_on_its_own():
    """init_process_group carries a bounded timeout (MUDPT_DIST_TIMEOUT_S, default 120 s; torch's default is 10 min = the driver's whole
    bench limit): the rank that waits for a peer that never arrives fails by itself, and that failure ends the job."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(MUDPT_BENCH_STUB="1", MUDPT_BENCH_STUB_HANG_RANK="1", MUDPT_BENCH_DEADLINE_S="100", MUDPT_DIST_TIMEOUT_S="4")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode not in (0, 124), (r.returncode, r.stderr)
    assert "[rank 0]" in r.stderr and "rank 0 exited with code" in r.stderr


def test_every_tracked_workload_has_its_traffic_profile():
    """bench.py's roofline.traffic comes from the rocprofv3 PMC passes committed under profiles/ for the SAME workload (tools/profile_round.sh);
    every workload it knows a tag for must have its file, with the dominant kernel's class in it."""
    sys.path.insert(0, ROOT)
    import bench
    for key, tag in bench.TRAFFIC_TAGS.items():
        path = bench.traffic_json(*key)
        assert path and os.path.exists(path), (key, tag)
        classes = json.load(open(path))["classes"]
        assert classes["gemm_pp"]["traffic_bytes_per_launch"] > 0, tag
    assert bench.traffic_json("vit_b16", 4, 50, "bf16") is None  # no tracked profile: traffic is reported as null
