"""Two ranks over RCCL on real GPUs (skipped on a one-GPU box): `bench.py --gpus 2` started bare spawns its ranks, the calibration
statistics of the two shards are merged by ONE all-reduce(MAX) over 2 x 768 floats (backend nccl = RCCL), and -- with
--calib-comm capi -- by libspq's own RCCL binding (spq_comm_init / spq_allreduce_minmax).  The scales that come out must be those
of one process seeing both shards: checked through the forward outputs agreeing across ranks being implied by identical scales
(bench.py asserts finiteness; the bit-identity of the merge itself is covered on gloo by tests/test_dist_gloo.py)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("comm", ["torch", "capi"])
@pytest.mark.timeout(600)
def test_two_ranks_over_rccl(comm):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--calib-comm", comm], env=env, capture_output=True, text=True, timeout=580)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["calibration"]["allreduce_elements"] == 2 * 768 and rec["calibration"]["comm"] == comm
    assert rec["value"] > 0 and rec["scaling"] == "weak"
