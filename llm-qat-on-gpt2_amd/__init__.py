"""MI355X-native fake-quantized linear layer for switchable-precision GPT-2.

Drop-in replacements for the reference's ``SPLinearWithLoRA`` / ``LoRALayer`` / ``LearnableFakeQuantize``
(Laurence-Wu/LLM-QAT-on-gpt2, part1_switchable_precision/{lora,quantization,quantization_methods}.py) whose
arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI of ``include/spq.h``.
"""
from . import _lib
from .fake_quantize import (LearnableFakeQuantize, LogQuantizationFunction, MinMaxQuantizationFunction,
                            apply_log_quantization, apply_minmax_quantization, fake_quantize)
from .sp_linear import LoRALayer, SPLinearWithLoRA
from .calibration import SpqComm, allreduce_calibration_stats, calibrate_layer, calibrate_model
from . import cpt, deploy, synthetic
from .blocks import SPMLP, SPAttention, SPBlock, SPLMHeadModel, SPModel, SwitchableLayerNorm
from .cpt import CPTLinear, LoRAAdapter, GradientQuantizer, calibrate_cpt_layer, calibrate_cpt_model, cpt_mlp_forward
from .graphs import GraphedForward

__all__ = ["SPLinearWithLoRA", "LoRALayer", "LearnableFakeQuantize", "MinMaxQuantizationFunction",
           "LogQuantizationFunction", "apply_minmax_quantization", "apply_log_quantization", "fake_quantize",
           "allreduce_calibration_stats", "calibrate_layer", "calibrate_model", "SpqComm",
           "SwitchableLayerNorm", "SPMLP", "SPAttention", "SPBlock", "SPModel", "SPLMHeadModel", "deploy", "cpt", "CPTLinear", "LoRAAdapter", "GradientQuantizer", "calibrate_cpt_layer", "calibrate_cpt_model", "cpt_mlp_forward", "GraphedForward"]
