"""Calibration protocol and its data-parallel merge.

``calibrate_model`` follows ``CalibrationManager._calibrate_precision`` + ``calibrate_lora_only``
(part1_switchable_precision/train_sp.py:47-123, 125-163) step for step.  The forward is replicated per GPU and
sharded over the batch, so ranks see different calibration batches: the one exchange step of the whole path is
``allreduce_calibration_stats`` -- a single all-reduce(MAX) over the flat buffer ``[-min_0 .. -min_n | max_0 ..
max_n]`` of every *input* quantizer still collecting, placed between the last calibration forward and
``finish_calibration``.  min/max are exact and associative, so every rank derives scales that are
bit-identical to one process having seen the union of the batches.  Weight and LoRA quantizers observe replicated
tensors and need no collective.  Backend: ``nccl`` (= RCCL over xGMI) for GPU tensors, ``gloo`` in the CPU tests.
"""
from typing import Iterable, List, Optional

import time

import torch
import torch.distributed as dist

from . import _lib
from .fake_quantize import LearnableFakeQuantize


def _collecting_quantizers(module_or_list) -> List[LearnableFakeQuantize]:
    """Every quantizer in statistics-collecting mode, whether or not it has seen a batch yet: module structure and the
    ``collecting_stats`` flags are replicated, so this list is the same on every rank."""
    if isinstance(module_or_list, torch.nn.Module):
        mods = module_or_list.modules()
    else:
        mods = module_or_list
    return [m for m in mods if isinstance(m, LearnableFakeQuantize) and m.collecting_stats]


class SpqComm:
    """RCCL communicator held through the C ABI (spq_comm_init / spq_allreduce_minmax, include/spq.h): the binding a
    host without torch.distributed would use.  ``bootstrap`` ships rank 0's 128-byte id to the other ranks; the default
    uses the already initialised torch.distributed group (any backend) as that side channel."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        import ctypes
        lib = _lib.load()
        buf = ctypes.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        handle = ctypes.c_void_p()
        _lib.check(lib.spq_comm_init(rank, world, ctypes.cast(buf, ctypes.c_void_p), ctypes.byref(handle)), 'spq_comm_init')
        self._h, self.rank, self.world = handle, rank, world

    @staticmethod
    def unique_id() -> bytes:
        import ctypes
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(_lib.load().spq_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), 'spq_comm_unique_id')
        return buf.raw

    @classmethod
    def from_process_group(cls, group: Optional[dist.ProcessGroup] = None) -> "SpqComm":
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(rank, world, box[0])

    def allreduce_max_(self, flat: torch.Tensor) -> torch.Tensor:
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        _lib.check(_lib.load().spq_allreduce_minmax(self._h, flat.data_ptr(), flat.numel(), _lib.stream_ptr(flat.device)),
                   'spq_allreduce_minmax')
        return flat

    def destroy(self):
        if self._h is not None and self._h.value:
            _lib.check(_lib.load().spq_comm_destroy(self._h), 'spq_comm_destroy')
        self._h = None


# what the last data collective of allreduce_calibration_stats moved and how long it took on this rank (None: no collective ran)
LAST_EXCHANGE = {"allreduce_ms": None, "elements": 0}


def allreduce_calibration_stats(module_or_quantizers, group: Optional[dist.ProcessGroup] = None,
                                comm: Optional[SpqComm] = None) -> int:
    """Merge the running min/max of every collecting quantizer across ranks with ONE data collective (preceded by a
    2-floats-per-quantizer agreement check, so that ranks with missing statistics fail together instead of hanging).

    Returns the number of fp32 elements exchanged (0 when not distributed).  Every rank must hold the same
    quantizers in the same order with the same statistic shapes (true for data-parallel replicas).

    Log-domain quirk carried from the reference (quantization.py:194-197): a rank whose first batch had no
    ``|x| > eps`` seeds its statistics with ``log2(eps)``, the smallest value a log-domain statistic can take; under
    MIN/MAX it can only pull the global minimum to what the clamp would have produced anyway.
    """
    qs = _collecting_quantizers(module_or_quantizers)
    if comm is None and (not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1):
        return 0
    if not qs:
        return 0

    def reduce_max_(t):
        if comm is not None:
            comm.allreduce_max_(t)                                  # C ABI -> ncclAllReduce(ncclMax) of RCCL
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)   # backend "nccl" is RCCL on ROCm
        return t

    # Agreement check (2 floats per quantizer): a rank whose loader ran dry, or that never reached some layer, holds no -- or
    # differently sized -- statistics; entering the data collective with a different element count would hang RCCL or merge
    # misaligned statistics.  max(n) and max(-n) over ranks must describe the same n everywhere; every rank sees the same
    # merged vector, so every rank raises (nobody is left waiting).
    dev = next((q.temp_min.device for q in qs if q.temp_min is not None), None)
    if dev is None:
        on_gpu = comm is not None or dist.get_backend(group) == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    sizes = [float(q.temp_min.numel()) if q.temp_min is not None else 0.0 for q in qs]
    chk = reduce_max_(torch.tensor(sizes + [-n for n in sizes], dtype=torch.float32, device=dev)).cpu()
    hi, lo = chk[:len(qs)], -chk[len(qs):]
    if not torch.equal(hi, lo):
        bad = [i for i in range(len(qs)) if hi[i] != lo[i]]
        raise RuntimeError(
            f"calibration statistics differ in size across ranks for {len(bad)} of {len(qs)} collecting quantizers (first: #{bad[0]}, "
            f"{int(lo[bad[0]])}..{int(hi[bad[0]])} elements, this rank {int(sizes[bad[0]])}): every rank must run the same "
            "number (>= 1) of calibration batches through the same layers")
    qs = [q for q, n in zip(qs, sizes) if n > 0]                    # nobody saw a batch for the others: they stay uncalibrated
    if not qs:
        return 0
    mins = [q.temp_min.reshape(-1) for q in qs]
    maxs = [q.temp_max.reshape(-1) for q in qs]
    flat = torch.cat([-torch.cat(mins), torch.cat(maxs)])                    # max(-min) == -min(min)
    # the collective itself, timed (SURVEY.md 8e asks for its latency): device events on the launch stream for GPU tensors
    if flat.is_cuda:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reduce_max_(flat)
        e1.record(); e1.synchronize()
        LAST_EXCHANGE.update(allreduce_ms=e0.elapsed_time(e1), elements=flat.numel())
    else:
        t0 = time.perf_counter()
        reduce_max_(flat)
        LAST_EXCHANGE.update(allreduce_ms=(time.perf_counter() - t0) * 1e3, elements=flat.numel())
    half = flat.numel() // 2
    off = 0
    for q in qs:
        n = q.temp_min.numel()
        q.temp_min = (-flat[off:off + n]).reshape(q.temp_min.shape).clone()
        q.temp_max = flat[half + off:half + off + n].reshape(q.temp_max.shape).clone()
        off += n
    return flat.numel()


def calibrate_weight_quantizers(model: torch.nn.Module, bits: int):
    """train_sp.py:58-83: one min/max pass over each frozen weight."""
    key = f'{bits}bit'
    for module in model.modules():
        if not hasattr(module, 'quantizers_weight') or key not in module.quantizers_weight:
            continue
        q = module.quantizers_weight[key]
        q.start_calibration()
        with torch.no_grad():
            q(module.linear.weight.data)
        q.finish_calibration(debug=False)


def calibrate_lora_only(model: torch.nn.Module, bits: int):
    """train_sp.py:125-163: LoRA factor quantizers are calibrated on the parameters themselves."""
    if bits >= 32:
        return
    key = f'{bits}bit'
    for module in model.modules():
        if not hasattr(module, 'lora_adapters') or key not in module.lora_adapters:
            continue
        lora = module.lora_adapters[key]
        if not lora.enabled:
            continue
        for q, t in ((lora.quantize_A, lora.lora_A), (lora.quantize_B, lora.lora_B)):
            q.start_calibration()
            with torch.no_grad():
                q(t)
            q.finish_calibration(debug=False)


def _set_calibration_mode(model, flag: bool):
    for module in model.modules():                    # models_sp.py:236-246 matches on the class name
        if module.__class__.__name__ == 'SPLinearWithLoRA':
            module.calibration_mode = flag


def calibrate_model(model: torch.nn.Module, bits: int, batches: Iterable, forward=None,
                    group: Optional[dist.ProcessGroup] = None, lora: bool = True, comm: Optional[SpqComm] = None) -> int:
    """Calibrate every quantizer of bit-width ``bits`` under ``model`` (any module tree containing
    SPLinearWithLoRA layers; a single layer works too).  ``batches`` yields this rank's calibration inputs;
    ``forward(model, batch)`` defaults to ``model(batch)``.  Returns the element count of the all-reduce."""
    if bits >= 32:
        return 0
    key = f'{bits}bit'
    if hasattr(model, 'set_precision'):
        model.set_precision(bits)
    calibrate_weight_quantizers(model, bits)
    started = []
    for module in model.modules():
        if hasattr(module, 'quantizers_input') and key in module.quantizers_input:
            module.quantizers_input[key].start_calibration()
            started.append(module.quantizers_input[key])
    _set_calibration_mode(model, True)
    try:
        with torch.no_grad():
            for batch in batches:
                forward(model, batch) if forward is not None else model(batch)
    finally:
        _set_calibration_mode(model, False)
    exchanged = allreduce_calibration_stats(started, group, comm)
    for q in started:
        q.finish_calibration(debug=False)
    if lora:
        calibrate_lora_only(model, bits)
    return exchanged


def calibrate_layer(layer, bits: int, batches: Iterable, group=None, comm: Optional[SpqComm] = None) -> int:
    """Convenience alias: a lone SPLinearWithLoRA is a model of one layer."""
    return calibrate_model(layer, bits, batches, group=group, comm=comm)
