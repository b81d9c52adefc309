"""CPU, world_size 2, gloo: the one exchange step of the data-parallel path -- the all-reduce(MAX) of [-min | max] of
every collecting input quantizer -- gives every rank the statistics one process would have got from the union of the
batches (bit-identical), with ONE collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_cpu as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    import llm_qat_on_gpt2_amd as pkg
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []
        orig = dist.all_reduce

        def counting(t, *a, **k):
            calls.append(t.numel())
            return orig(t, *a, **k)
        dist.all_reduce = counting

        K = 48
        # two "layers" (per-channel minmax, per-tensor log) as a model of several quantizers in flight
        q1 = pkg.LearnableFakeQuantize(4, channel_dim=-1, quantizer_type="minmax", is_input=True)
        q2 = pkg.LearnableFakeQuantize(6, channel_dim=-1, quantizer_type="log", per_channel=False, is_input=True)
        q3 = pkg.LearnableFakeQuantize(8, channel_dim=0)                      # not collecting: must be left alone
        # part2's quantizer (per-width scale dictionaries) takes part in the same single collective
        q4 = pkg.cpt.LearnableFakeQuantize(6, channel_dim=-1, quantizer_type="log", is_input=True)
        model = torch.nn.ModuleList([q1, q2, q3, q4])
        batches = [O.make_workload(64, K, 8, 4, seed=10 + 2 * r + i, batch=2)[4] for r in range(world) for i in range(2)]
        mine = batches[2 * rank:2 * rank + 2]
        for q, log in ((q1, False), (q2, True), (q4, True)):
            q.start_calibration()
            # the statistics kernel is HIP-only; on CPU feed this rank's statistics from the oracle (host logic under test)
            oq = O.QuantState(q.num_bits, q.quantizer_type, -1, q.per_channel)
            oq.start()
            for b in mine:
                oq.observe(b)
            q.temp_min, q.temp_max, q.num_batches_collected = oq.tmin.clone(), oq.tmax.clone(), len(mine)
        n = pkg.allreduce_calibration_stats(model)
        # [agreement check: 2 floats per collecting quantizer] + [the ONE data collective]
        assert calls == [2 * 3, n] and n == 2 * (K + 1 + K), (calls, n)
        # expected: one process over the union of all ranks' batches
        for q in (q1, q2, q4):
            oq = O.QuantState(q.num_bits, q.quantizer_type, -1, q.per_channel)
            oq.start()
            for b in batches:
                oq.observe(b)
            assert torch.equal(q.temp_min, oq.tmin) and torch.equal(q.temp_max, oq.tmax), f"rank {rank}: merged stats differ"
            assert q.temp_min.shape == oq.tmin.shape
        assert q3.temp_min is None
        # nothing collecting any more -> no collective
        q1.collecting_stats = q2.collecting_stats = q4.collecting_stats = False
        assert pkg.allreduce_calibration_stats(model) == 0 and len(calls) == 2
        # a rank whose loader ran dry (no statistics for a collecting quantizer): every rank raises, nobody hangs
        q5 = pkg.LearnableFakeQuantize(4, channel_dim=-1, quantizer_type="minmax", is_input=True)
        q5.start_calibration()
        if rank == 0:
            q5.temp_min, q5.temp_max, q5.num_batches_collected = torch.zeros(1, 1, K), torch.ones(1, 1, K), 1
        try:
            pkg.allreduce_calibration_stats([q5])
            raise AssertionError("missing statistics on one rank were not detected")
        except RuntimeError as e:
            assert "differ in size across ranks" in str(e), str(e)
        # nobody saw a batch: no data collective, the quantizer stays as it is
        q6 = pkg.LearnableFakeQuantize(4, channel_dim=-1, quantizer_type="minmax", is_input=True)
        q6.start_calibration()
        assert pkg.allreduce_calibration_stats([q6]) == 0 and q6.temp_min is None
        ret[rank] = "ok"
    except Exception as e:  # pragma: no cover
        ret[rank] = f"{type(e).__name__}: {e}"
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_allreduce_calibration_stats_world2_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_single_process_is_a_no_op():
    import llm_qat_on_gpt2_amd as pkg
    q = pkg.LearnableFakeQuantize(4, channel_dim=-1, is_input=True)
    q.start_calibration()
    q.temp_min, q.temp_max = torch.zeros(1, 1, 4), torch.ones(1, 1, 4)
    assert pkg.allreduce_calibration_stats([q]) == 0
    assert torch.equal(q.temp_max, torch.ones(1, 1, 4))
