"""ctypes binding of libspq.so (the C ABI declared in include/spq.h).

There is NO CPU fallback: if the library is missing, or a tensor is not on a gfx950 device, the calls below
raise.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C llm-qat-on-gpt2_amd/csrc``).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPQ_LIB") or os.path.join(_HERE, "libspq.so")   # SPQ_LIB: an alternative build (kernel tuning)

MINMAX, LOG = 0, 1
STAGE_ALL, STAGE_ACTIVATIONS, STAGE_CONTRACTION = 0, 1, 2
EPILOGUE_NONE, EPILOGUE_GELU = 0, 1
COMM_ID_BYTES = 128
LIMB_SCALE_WORKSPACE_BYTES = 16384
PATH_AUTO, PATH_F32, PATH_F16X2, PATH_F16X3, PATH_I8 = 0, 1, 2, 4, 5    # (3: the former byte-level path, tools/variants/gemm_u8x2.h)
PATH_NAMES = {1: "f32", 2: "f16x2", 4: "f16x3", 5: "i8"}
LOG_DIRECT = 2
QTYPE_CODE = {"minmax": MINMAX, "log": LOG}            # part1 quantizers
QTYPE_CODE_CPT = {"minmax": MINMAX, "log": LOG_DIRECT}  # part2 quantizers (log without the level round trip)
QTYPE_ANY = {"minmax": MINMAX, "log": LOG, "log_direct": LOG_DIRECT}

_lib = None
_p = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_f = C.c_float
_sz = C.c_size_t


class PrepareArgs(C.Structure):
    """struct spq_prepare_args (include/spq.h): the arguments of spq_prepare_f16x2."""
    _fields_ = [("W", _p), ("N", _i64), ("K", _i64), ("sw", _p), ("zw", _p),
                ("w_per_channel", _int), ("w_bits", _int), ("w_qtype", _int), ("w_symmetric", _int),
                ("B", _p), ("r", _i64), ("sb", _p), ("zb", _p),
                ("b_per_channel", _int), ("b_bits", _int), ("b_qtype", _int), ("b_symmetric", _int), ("scaling", _f),
                ("A", _p), ("sa", _p), ("za", _p),
                ("a_per_channel", _int), ("a_bits", _int), ("a_qtype", _int), ("a_symmetric", _int),
                ("sx", _p), ("x_per_channel", _int),
                ("w_prep", _p), ("w_prep_bytes", _sz), ("w_rowscale", _p), ("a_prep", _p), ("path", _int)]


class FwdArgs(C.Structure):
    """struct spq_fwd_args (include/spq.h)."""
    _fields_ = [("M", _i64), ("K", _i64), ("N", _i64), ("r", _i64),
                ("bits", _int), ("qtype", _int), ("symmetric", _int), ("quantize_input", _int),
                ("x_per_channel", _int), ("path", _int),
                ("x", _p), ("sx", _p), ("zx", _p), ("x_limb_scale", _p),
                ("w_prep", _p), ("w_rowscale", _p), ("bias", _p), ("a_prep", _p), ("b_prep", _p),
                ("lora_scaling", _f),
                ("y", _p), ("workspace", _p), ("workspace_bytes", _sz),
                ("ev_gemm_begin", _p), ("ev_gemm_end", _p), ("t_out", _p), ("lora_on_fq_input", _int), ("stage", _int), ("epilogue", _int), ("a_limb_scale", _p),
                ("prepare", C.POINTER(PrepareArgs)), ("ln_weight", _p), ("ln_bias", _p), ("ln_eps", _f),
                ("out_levels", _p), ("out_levels_ld", _i64), ("out_scale", _p), ("out_scale_per_channel", _int), ("out_bits", _int),
                ("out_levels_lo", _p), ("out_zero", _p), ("out_qtype", _int), ("out_symmetric", _int), ("out_limb_scale", _p)]


# name -> (restype, argtypes); must list every symbol include/spq.h declares (tests/test_cabi.py checks).
SIGNATURES = {
    "spq_comm_unique_id": (_int, [_p]),
    "spq_comm_init": (_int, [_int, _int, _p, C.POINTER(C.c_void_p)]),
    "spq_comm_destroy": (_int, [_p]),
    "spq_allreduce_minmax": (_int, [_p, _p, _sz, _p]),
    "spq_dynamic_limb_scale": (_int, [_p, _i64, _p, _p, _sz, _p]),
    "spq_gemm_f32_tn_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "spq_gemm_f32_tn": (_int, [_p, _i64, _p, _i64, _i64, _i64, _i64, _f, _p, _p, _sz, _p]),
    "spq_prepare_cpt": (_int, [_p, _i64, _i64, _p, _p, _int, _int, _int, _int, _p, _p, _i64, _p, _p, _int, _int, _int, _int, _f,
                                _p, _int, _int, _p, _sz, _p, _p, _p, _p, _p, _p]),
    "spq_layernorm": (_int, [_p, _i64, _i64, _p, _p, _f, _p, _p]),
    "spq_debug_reload_switches": (_int, []),
    "spq_debug_split_k": (_int, [_i64, _i64, _i64, _i64, _int]),
    "spq_version": (_int, []),
    "spq_last_error": (C.c_char_p, []),
    "spq_device_arch": (_int, [C.c_char_p, _int]),
    "spq_stats_workspace_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "spq_minmax_stats": (_int, [_p, _i64, _i64, _i64, _int, _int, _f, _f, _int, _p, _p, _p, _sz, _p]),
    "spq_finish_scale": (_int, [_p, _p, _i64, _int, _int, _int, _f, _p, _p, _p]),
    "spq_fakequant": (_int, [_p, _i64, _i64, _i64, _p, _p, _int, _int, _int, _int, _p, _p, _int, _p]),
    "spq_fakequant_transposed": (_int, [_p, _i64, _i64, _p, _p, _int, _int, _int, _int, _f, _p, _p]),
    "spq_gemm_f32_nt": (_int, [_p, _i64, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _f, _p, _p, _i64, _i64, _i64, _p]),
    "spq_fwd_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _int]),
    "spq_linear_lora_fwd": (_int, [C.POINTER(FwdArgs), _p]),
    "spq_prep_f16x2_bytes": (_sz, [_i64, _i64, _i64]),
    "spq_prep_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "spq_prepare_f16x2_args": (_int, [C.POINTER(PrepareArgs), _p]),
    "spq_prepare_f16x2": (_int, [_p, _i64, _i64, _p, _p, _int, _int, _int, _int,          # W
                                 _p, _i64, _p, _p, _int, _int, _int, _int, _f,            # B
                                 _p, _p, _p, _int, _int, _int, _int,                      # A
                                 _p, _int, _p, _sz, _p, _p, _p]),                         # sx, outputs, stream
}


class SpqError(RuntimeError):
    pass


def load():
    """Load libspq.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"libspq.so not found at {LIB_PATH}: the HIP extension is not built and there is no CPU fallback. "
            "Run `python -c \"import __graft_entry__ as g; g.build()\"` at the repo root.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if name.startswith("spq_debug_") and not hasattr(lib, name):
            continue                                      # (an older build loaded through SPQ_LIB for an A/B)
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def set_switch(name: str, value):
    """Set (or, with None, unset) one of the library's SPQ_* tuning switches and make the library re-read them: they are read from
    the environment once, at load time, not per call (spq_debug_reload_switches).  For tests and tools."""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    if hasattr(load(), "spq_debug_reload_switches"):
        check(load().spq_debug_reload_switches(), "spq_debug_reload_switches")


def check(rc, what):
    if rc != 0:
        msg = load().spq_last_error().decode("utf-8", "replace")
        raise SpqError(f"{what} failed with code {rc}: {msg}")


def require_gpu(t: torch.Tensor, what: str = "tensor"):
    if not t.is_cuda:
        raise RuntimeError(
            f"llm_qat_on_gpt2_amd: {what} is on '{t.device}'. The fake-quant path runs only as HIP kernels on a "
            "gfx950 (MI355X) device; there is no CPU fallback.")
    if t.dtype != torch.float32:
        raise TypeError(f"llm_qat_on_gpt2_amd: {what} must be float32, got {t.dtype}")


def stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


_arch_checked = set()


def check_device(device):
    """Once per device: the library is gfx950-only."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _arch_checked:
        return
    with torch.cuda.device(idx):
        buf = C.create_string_buffer(128)
        check(load().spq_device_arch(buf, 128), "spq_device_arch")
    _arch_checked.add(idx)


_workspaces = {}


class _NoContext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CONTEXT = _NoContext()


def on_device(device):
    """``torch.cuda.device(device)`` only when it is not the current device already (the switch costs ~5 us per call)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_CONTEXT
    return torch.cuda.device(idx)


def workspace(device, nbytes: int, stream: int = None, slot: int = 0) -> torch.Tensor:
    """One grow-only scratch buffer per (device, stream, slot); calls on one stream serialise, so layers share it.  Slot 1 is the
    level matrix a producer layer hands to its consumer (cpt_mlp_forward): the consumer of layer i has read it before the
    producer of layer i+1 -- later on the same stream -- writes it again."""
    key = (device.index, stream_ptr(device) if stream is None else stream, slot)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf
