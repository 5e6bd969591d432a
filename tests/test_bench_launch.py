"""bench.py's launch contract: `--gpus N` means N RCCL ranks, one per GPU -- never a silent
single rank. CPU part: the refusals. GPU part: one rank directly and under torch.distributed.run; both shard
schemes on two ranks when the box has two GPUs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--steps", "2", "--warmup", "1", "--repeats", "2", "--cpu-baseline-seconds", "0", "--e2e-seconds", "0"]


def run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def gpu_count():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.skipif(gpu_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_more_gpus_than_the_box_has_is_refused():
    r = run(["--gpus", "2"] + SMALL)
    assert r.returncode != 0
    assert "refusing" in r.stderr and "{" not in r.stdout


def test_world_size_must_match_gpus():
    # a launcher that started one rank for a two-GPU request: no line, non-zero exit
    r = run(["--gpus", "2"] + SMALL, {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                                       "MASTER_PORT": "29999"})
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout) and "\"n_gpus\"" not in r.stdout


def test_chunk_shard_needs_one_cloud():
    r = run(["--shard", "chunks", "--clouds", "2"] + SMALL)
    assert r.returncode != 0 and "--clouds 1" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_single_rank_line():
    r = run(["--gpus", "1", "--clouds", "2", "--points", "100000"] + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    assert d["repeats"]["blocks"] == 2 and d["value"] > 0
    assert d["roofline"]["frac"] > 0 and d["decode"]["value"] > 0


@pytest.mark.gpu
def test_single_rank_under_the_launcher_takes_the_rccl_path():
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", BENCH, "--gpus", "1", "--clouds", "2", "--points", "100000"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json_line(r.stdout)["n_gpus"] == 1


@pytest.mark.gpu
def test_single_rank_chunk_shard_runs_the_exchange_protocol():
    r = run(["--gpus", "1", "--workload", "c2", "--shard", "chunks", "--points", "200000"] + SMALL)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json_line(r.stdout)
    assert d["scaling"] == "strong" and d["config"]["shard"] == "chunks" and d["n_gpus"] == 1


@pytest.mark.gpu
@pytest.mark.skipif(gpu_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("shard", ["clouds", "chunks"])
def test_two_ranks(shard):
    extra = ["--clouds", "4", "--points", "200000"] if shard == "clouds" else ["--workload", "c2", "--shard", "chunks",
                                                                               "--points", "400000"]
    r = run(["--gpus", "2"] + extra + SMALL, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json_line(r.stdout)
    assert d["n_gpus"] == 2
    one = json_line(run(["--gpus", "1"] + extra + SMALL).stdout)
    if shard == "clouds":  # weak scaling: twice the clouds, twice the bytes
        assert d["job_stage1_bytes"] == 2 * one["job_stage1_bytes"]
    else:                  # strong scaling: the same cloud, the same stream size whatever the cut
        assert d["job_stage1_bytes"] == one["job_stage1_bytes"]
