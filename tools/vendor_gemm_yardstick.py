"""Yardstick only (not used by the library): what does the vendor's fp16 GEMM (torch.mm -> hipBLASLt / rocBLAS) take for the MFMA work of the headline
contraction written as ONE plain GEMM -- the two weight limbs concatenated along K: [q | q] . [Whi | Wlo]^T (K' = 2 x 768) plus the LoRA products
[thi | thi | tlo] . [Bhi | Blo | Bhi]^T (K' = 3 x 64)?  Output dtype fp16 here (torch.mm), so this is a timing of the matrix work, not a usable result."""
import torch
dev = 'cuda:0'
M, N = 8192, 3072
for Kp, what in ((2 * 768 + 3 * 64, 'base + LoRA limbs, K\' = 1728'), (2 * 768, 'base limbs only, K\' = 1536'), (768, 'one limb, K = 768')):
    a = (torch.randint(-7, 8, (M, Kp), device=dev)).half()
    b = (torch.randn(N, Kp, device=dev) * 100).half()
    for dt_out in (None,):
        for _ in range(10): c = torch.mm(a, b.t())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): c = torch.mm(a, b.t())
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 10
        print(f"torch.mm fp16 [{M} x {Kp}] . [{Kp} x {N}] ({what}): {us:.1f} us = {2 * M * N * Kp / us / 1e6:.0f} TFLOP/s of f16 MFMA work")
try:
    a = (torch.randint(-7, 8, (M, 1728), device=dev)).half(); b = (torch.randn(N, 1728, device=dev) * 100).half()
    c = torch.mm(a, b.t(), out_dtype=torch.float32)
    for _ in range(10): c = torch.mm(a, b.t(), out_dtype=torch.float32)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): c = torch.mm(a, b.t(), out_dtype=torch.float32)
    e1.record(); e1.synchronize()
    print(f"torch.mm fp16 -> fp32 output, K' = 1728: {e0.elapsed_time(e1) * 10:.1f} us")
except Exception as ex:
    print("fp32-output mm not available:", type(ex).__name__, str(ex)[:100])
