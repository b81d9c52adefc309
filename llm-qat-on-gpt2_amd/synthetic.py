"""Seeded synthetic tensors of the benchmark and test workloads: GPT-2-style weights, activations with
outlier entries, LoRA factors.  Pure data generation on the CPU (torch generators), shared by ``bench.py``, ``tools/`` and --
so that the checker sees the very same inputs -- by ``oracle/``; no arithmetic of the path lives here."""
import math
from typing import Optional

import torch


def kaiming_uniform_a5(rows: int, cols: int, gen: torch.Generator) -> torch.Tensor:
    """nn.init.kaiming_uniform_(t[rows, cols], a=sqrt(5)) (lora.py:37): U(-b, b), b = 1/sqrt(fan_in),
    fan_in = cols for a 2-D tensor."""
    bound = math.sqrt(6.0 / ((1 + 5.0) * cols))
    return (torch.rand(rows, cols, generator=gen) * 2 - 1) * bound


def make_workload(M: int, K: int, N: int, r: int, seed: int = 0, batch: Optional[int] = None):
    """W~N(0,.02^2), bias~N(0,.02^2), x~N(0,1) with 0.1% entries x20, A kaiming-uniform, B~N(0,.01^2)."""
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(N, K, generator=g) * 0.02
    bias = torch.randn(N, generator=g) * 0.02
    A = kaiming_uniform_a5(K, r, g)
    B = torch.randn(r, N, generator=g) * 0.01

    def act(s):
        gg = torch.Generator().manual_seed(1000 + s)
        x = torch.randn(M, K, generator=gg)
        x = torch.where(torch.rand(M, K, generator=gg) < 1e-3, x * 20, x)
        if batch:
            x = x.view(batch, M // batch, K)
        return x
    return W, bias, A, B, act(0), act(1)


def make_cpt_workload(M, K, N, r, seed=0, batch=1):
    """The same tensors for part2's CPTLinear: ``lora_B`` is [N, r] there (cpt_model.py:24); non-zero, because the
    reference's zero init would make the LoRA branch trivially zero."""
    W, bias, A, B_rn, x0, x1 = make_workload(M, K, N, r, seed=seed, batch=batch)
    return W, bias, A, B_rn.t().contiguous(), x0, x1
