"""Where the 12-14 us of the weight-preparation launch go (kernel tuning): times spq_prepare_f16x2 at the c_fc shape with
parts of the job switched off (HIP events, warm)."""
import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd import synthetic as S
L = pkg._lib; lib = L.load(); dev = "cuda:0"
N, K, r = 3072, 768, 64
W, bias, A, B, x0, x1 = S.make_workload(64, K, N, r, seed=0)
W, A, B = W.to(dev), A.to(dev), B.to(dev)
sw = torch.rand(N, device=dev) * 0.01 + 0.005; zw = torch.zeros(N, device=dev)
sb = torch.rand(N, device=dev) * 0.001 + 0.001; zb = torch.zeros(N, device=dev)
sa = torch.rand(r, device=dev) * 0.01 + 0.01; za = torch.zeros(r, device=dev)
sx = torch.rand(K, device=dev) + 0.5
wprep = torch.empty(lib.spq_prep_bytes(N, K, r, L.PATH_F16X2), dtype=torch.uint8, device=dev)
rows = torch.empty(3072, device=dev); aT = torch.zeros(64, K, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(with_b, with_a, bits=4, x_pc=1, n=200):
    pa = L.PrepareArgs(W=W.data_ptr(), N=N, K=K, sw=sw.data_ptr(), zw=zw.data_ptr(), w_per_channel=1, w_bits=bits, w_qtype=0, w_symmetric=1,
                       B=B.data_ptr() if with_b else None, r=r if with_b else 0, sb=sb.data_ptr(), zb=zb.data_ptr(), b_per_channel=1, b_bits=bits, b_qtype=0,
                       b_symmetric=1, scaling=1.0, A=A.data_ptr() if (with_a and with_b) else None, sa=sa.data_ptr(), za=za.data_ptr(), a_per_channel=1,
                       a_bits=bits, a_qtype=0, a_symmetric=1, sx=sx.data_ptr(), x_per_channel=x_pc, w_prep=wprep.data_ptr(),
                       w_prep_bytes=wprep.numel(), w_rowscale=rows.data_ptr(), a_prep=aT.data_ptr() if (with_a and with_b) else None, path=L.PATH_F16X2)
    for _ in range(20):
        L.check(lib.spq_prepare_f16x2_args(ctypes.byref(pa), st), "prep")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        lib.spq_prepare_f16x2_args(ctypes.byref(pa), st)
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f"prepare (back-to-back launches, us): full {run(True, True):.1f} | no FQ(A)^T tiles {run(True, False):.1f} | W rows only {run(False, False):.1f} | "
      f"W rows only, 32-bit identity quantizer {run(False, False, bits=32):.1f}")
x = torch.randn(8192, 768, device=dev)
y = torch.empty_like(x)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5): y.copy_(x)
a.record()
for _ in range(100): y.copy_(x)
b.record(); b.synchronize()
print(f"for scale: torch copy of 25 MB {a.elapsed_time(b) / 100 * 1e3:.1f} us")
