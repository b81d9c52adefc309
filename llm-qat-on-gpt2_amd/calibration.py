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

import torch
import torch.distributed as dist

from .fake_quantize import LearnableFakeQuantize


def _collecting_quantizers(module_or_list) -> List[LearnableFakeQuantize]:
    if isinstance(module_or_list, torch.nn.Module):
        mods = module_or_list.modules()
    else:
        mods = module_or_list
    out = []
    for m in mods:
        if isinstance(m, LearnableFakeQuantize) and m.collecting_stats and m.temp_min is not None:
            out.append(m)
    return out


def allreduce_calibration_stats(module_or_quantizers, group: Optional[dist.ProcessGroup] = None) -> int:
    """Merge the running min/max of every collecting quantizer across ranks with ONE collective.

    Returns the number of fp32 elements exchanged (0 when not distributed).  Every rank must hold the same
    quantizers in the same order with the same statistic shapes (true for data-parallel replicas).

    Log-domain quirk carried from the reference (quantization.py:194-197): a rank whose first batch had no
    ``|x| > eps`` seeds its statistics with ``log2(eps)``, the smallest value a log-domain statistic can take; under
    MIN/MAX it can only pull the global minimum to what the clamp would have produced anyway.
    """
    qs = _collecting_quantizers(module_or_quantizers)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1 or not qs:
        return 0
    mins = [q.temp_min.reshape(-1) for q in qs]
    maxs = [q.temp_max.reshape(-1) for q in qs]
    flat = torch.cat([-torch.cat(mins), torch.cat(maxs)])           # max(-min) == -min(min)
    dist.all_reduce(flat, op=dist.ReduceOp.MAX, group=group)
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
                    group: Optional[dist.ProcessGroup] = None, lora: bool = True) -> int:
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
    exchanged = allreduce_calibration_stats(started, group)
    for q in started:
        q.finish_calibration(debug=False)
    if lora:
        calibrate_lora_only(model, bits)
    return exchanged


def calibrate_layer(layer, bits: int, batches: Iterable, group=None) -> int:
    """Convenience alias: a lone SPLinearWithLoRA is a model of one layer."""
    return calibrate_model(layer, bits, batches, group=group)
