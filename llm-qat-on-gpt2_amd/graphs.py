"""HIP-graph replay of an eval forward.

The fused layer is two or three launches (activation pass, contraction, in a re-quantising forward the weight preparation) issued
from Python through ctypes; at small token counts -- perplexity evaluation at batch 1, a 128-token probe -- those launches and
the Python between them, not the kernels, set the pace (a 768 -> 768 layer at 1024 tokens: ~12 us of kernels behind ~60 us of
host work).  Everything the library launches goes to the caller's stream with no host synchronisation and no hidden allocation,
so a whole forward can be captured once and replayed as ONE graph launch.

The reference has no counterpart (it is eager PyTorch); this is plumbing around the drop-in classes, not part of their surface.
"""
import torch


class GraphedForward:
    """``g = GraphedForward(fn, *example_inputs)`` captures ``fn(*example_inputs)`` (no grad) into a HIP graph after ``warmup`` eager
    calls; ``g(*inputs)`` copies the inputs into the captured buffers, replays, and returns the captured output tensor(s) -- the SAME
    tensors on every call: clone what must outlive the next call.

    What a capture freezes: shapes and dtypes of the inputs, every device pointer the forward used (parameters, calibration
    buffers, cached weight operands) and every host-side decision (operand path, whether prepared operands were reused).  Values
    behind those pointers stay live -- an optimizer step on the LoRA matrices is seen by a replay IF the capture re-quantises
    (``module.cache_operands = False`` or ``SPQ_CACHE_OPERANDS=0`` while capturing); a capture that reused cached operands keeps
    using them.  Re-capture (``g.capture()``) after ``set_precision``, a calibration, loading a checkpoint, or re-assigning a
    parameter's ``.data``.
    """

    def __init__(self, fn, *example_inputs, warmup: int = 3):
        if not example_inputs or not all(isinstance(t, torch.Tensor) and t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedForward needs device tensors as example inputs (there is no CPU path)")
        self.fn = fn
        self.warmup = int(warmup)
        self.static_inputs = [t.detach().clone() for t in example_inputs]
        self.graph = None
        self.static_outputs = None
        self.capture()

    def capture(self):
        dev = self.static_inputs[0].device
        with torch.no_grad(), torch.cuda.device(dev):
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                    # workspaces grow, operands get prepared, lazy buffers appear
                for _ in range(max(self.warmup, 1)):
                    self.fn(*self.static_inputs)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.fn(*self.static_inputs)
        self.graph, self.static_outputs = graph, out
        return self

    def __call__(self, *inputs):
        if len(inputs) != len(self.static_inputs):
            raise RuntimeError(f"captured with {len(self.static_inputs)} inputs, called with {len(inputs)}")
        for dst, src in zip(self.static_inputs, inputs):
            if src.shape != dst.shape or src.dtype != dst.dtype or src.device != dst.device:
                raise RuntimeError(f"captured for {tuple(dst.shape)} {dst.dtype} on {dst.device}, called with {tuple(src.shape)} {src.dtype} on "
                                   f"{src.device}: capture another GraphedForward for that shape")
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_outputs
