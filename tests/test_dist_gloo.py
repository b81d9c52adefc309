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


# ---- calibrate_model end to end at world size 2 --------------------------------------------------------------
# The statistics and scale kernels are HIP-only, so on the CPU the two quantizer methods that call them are replaced by the
# oracle's torch arithmetic (test doubles); everything else is the product's: the SPLinearWithLoRA protocol that calibrate_model
# drives (weight quantizers, start / finish, calibration_mode, forwards through the model, LoRA quantizers), the ONE data
# collective and its agreement check.
def _cpu_doubles(pkg):
    LFQ = pkg.LearnableFakeQuantize

    def collect(self, x):
        oq = O.QuantState(self.num_bits, self.quantizer_type, self.channel_dim, self.per_channel)
        oq.start()
        if self.temp_min is not None:
            oq.tmin, oq.tmax, oq.nbatches = self.temp_min, self.temp_max, self.num_batches_collected
        oq.observe(x.detach())
        self.temp_min, self.temp_max = oq.tmin, oq.tmax
        self.num_batches_collected += 1

    def finish(self, debug=False):
        if self.num_batches_collected > 0 and self.temp_min is not None:
            oq = O.QuantState(self.num_bits, self.quantizer_type, self.channel_dim, self.per_channel)
            oq.start()
            oq.tmin, oq.tmax, oq.nbatches = self.temp_min, self.temp_max, self.num_batches_collected
            oq.finish()
            self.running_min, self.running_max = oq.running_min.clone(), oq.running_max.clone()
            self.scale, self.zero_point = oq.scale.clone(), oq.zero_point.clone()
            self.calibrated = True
            self._epoch += 1
        self.collecting_stats = False
        self.temp_min = self.temp_max = None
    LFQ._collect_statistics_batch = collect
    LFQ.finish_calibration = finish


class SPLinearWithLoRA(torch.nn.Module):                    # (calibrate_model finds its layers by this class name, as models_sp.py does)
    def __init__(self, pkg, K, N, bits, qtype):
        super().__init__()
        key = f"{bits}bit"
        self.linear = torch.nn.Linear(K, N)
        self.quantizers_weight = torch.nn.ModuleDict({key: pkg.LearnableFakeQuantize(bits, channel_dim=0, quantizer_type=qtype)})
        self.quantizers_input = torch.nn.ModuleDict({key: pkg.LearnableFakeQuantize(bits, channel_dim=-1, quantizer_type=qtype, is_input=True)})
        self.lora_adapters = torch.nn.ModuleDict()
        self.calibration_mode = False
        self.key = key
        self.saw_calibration_mode = []

    def forward(self, x):
        self.saw_calibration_mode.append(self.calibration_mode)
        self.quantizers_input[self.key](x)                   # collecting: records statistics, returns x
        return torch.nn.functional.linear(x, self.linear.weight, self.linear.bias)


def _calibrate_worker(rank, world, port, ret):
    import llm_qat_on_gpt2_amd as pkg
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _cpu_doubles(pkg)
        calls = []
        orig = dist.all_reduce

        def counting(t, *a, **k):
            calls.append(t.numel())
            return orig(t, *a, **k)
        dist.all_reduce = counting
        K, H, N, bits = 24, 40, 16, 6

        def make_model():
            torch.manual_seed(0)                              # replicated weights
            return torch.nn.Sequential(SPLinearWithLoRA(pkg, K, H, bits, "minmax"), torch.nn.GELU(), SPLinearWithLoRA(pkg, H, N, bits, "log"))
        batches = [O.make_workload(32, K, 8, 4, seed=40 + i, batch=2)[4] for i in range(2 * world)]
        model = make_model()
        n = pkg.calibrate_model(model, bits, batches[2 * rank:2 * rank + 2], lora=False)
        # [agreement check: 2 floats per input quantizer] + [the ONE data collective over both layers' statistics]
        assert calls == [2 * 2, n] and n == 2 * (K + H), (calls, n)
        from llm_qat_on_gpt2_amd import calibration
        assert calibration.LAST_EXCHANGE["elements"] == n and calibration.LAST_EXCHANGE["allreduce_ms"] > 0
        # the reference's protocol: every calibration forward ran with calibration_mode set, and it is cleared afterwards
        for layer in (model[0], model[2]):
            assert layer.saw_calibration_mode == [True, True] and layer.calibration_mode is False
        # one process over the union of the batches (no collective: world size 1 semantics through a fresh, undistributed call)
        dist.all_reduce = orig
        ref = make_model()
        for layer in (ref[0], ref[2]):
            layer.quantizers_weight[layer.key].start_calibration()
            layer.quantizers_weight[layer.key](layer.linear.weight.data)
            layer.quantizers_weight[layer.key].finish_calibration()
            layer.quantizers_input[layer.key].start_calibration()
        with torch.no_grad():
            for b in batches:
                ref(b)
        for layer in (ref[0], ref[2]):
            layer.quantizers_input[layer.key].finish_calibration()
        for got, want in ((model[0], ref[0]), (model[2], ref[2])):
            for qs in ("quantizers_input", "quantizers_weight"):
                g, w_ = getattr(got, qs)[got.key], getattr(want, qs)[want.key]
                assert g.calibrated and not g.collecting_stats
                for name in ("scale", "zero_point", "running_min", "running_max"):
                    assert torch.equal(getattr(g, name), getattr(w_, name)), f"rank {rank} {qs}.{name}"
        ret[rank] = "ok"
    except Exception as e:  # pragma: no cover
        import traceback
        ret[rank] = f"{type(e).__name__}: {e}\n{traceback.format_exc()}"
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_calibrate_model_world2_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_calibrate_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_single_process_is_a_no_op():
    import llm_qat_on_gpt2_amd as pkg
    q = pkg.LearnableFakeQuantize(4, channel_dim=-1, is_input=True)
    q.start_calibration()
    q.temp_min, q.temp_max = torch.zeros(1, 1, 4), torch.ones(1, 1, 4)
    assert pkg.allreduce_calibration_stats([q]) == 0
    assert torch.equal(q.temp_max, torch.ones(1, 1, 4))
