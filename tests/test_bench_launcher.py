"""CPU: `python bench.py --gpus N` must work exactly as the driver starts it -- bare, with no WORLD_SIZE in the environment.
The parent spawns the N ranks before any GPU call; `--dry-run` lets the whole protocol (rank formation over gloo, the ONE
calibration all-reduce of 2 x 768 statistics, barrier + max-over-ranks timing, one JSON line from rank 0) run without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["OMP_NUM_THREADS"] = "1"
    return env


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n", [2, 3])
def test_bench_spawns_its_own_ranks(n):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
                        "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                                      # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["dry_run"] is True and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["calibration"]["allreduce_elements"] == 2 * 768            # [-min | max] of one per-channel input quantizer
    assert rec["scaling"] == "weak" and rec["value"] is None


@pytest.mark.timeout(120)
def test_bench_single_rank_dry_run_and_world_mismatch():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "0"],
                       env=_clean_env(), capture_output=True, text=True, timeout=100)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["calibration"]["allreduce_elements"] == 0
    env = _clean_env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=100)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


@pytest.mark.timeout(300)
def test_bench_under_torch_distributed_run():
    """The other way the driver starts N > 1: torch.distributed.run provides the environment; no second level of spawning."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--dry-run"], env=_clean_env(), capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["calibration"]["allreduce_elements"] == 2 * 768
