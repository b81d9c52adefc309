"""True error (against an fp64 product on the GPU) of the two-limb x two-limb contraction (SPQ_PATH_F16X3 with an identity
quantizer, as the backward uses it) and of the fp32-MFMA contraction, over seeds and input distributions."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg
from llm_qat_on_gpt2_amd.sp_linear import _LimbGemm, _gemm_nt
dev = 'cuda:0'
lg = _LimbGemm()
for dist in ('normal', 'outliers x20 (0.1%)', 'heavy tail (student-t 3)', 'one huge channel x100'):
    worst_l = worst_f = 0.0
    for seed in range(12):
        g = torch.Generator(device='cpu').manual_seed(seed)
        M, C, R = 1024, 3072, 768
        a = torch.randn(M, C, generator=g)
        if dist.startswith('outliers'):
            a = torch.where(torch.rand(M, C, generator=g) < 1e-3, a * 20, a)
        elif dist.startswith('heavy'):
            a = a / torch.sqrt(torch.distributions.Chi2(3.0).sample((M, C)) / 3.0)
        elif dist.startswith('one huge'):
            a[:, 7] *= 100
        b = torch.randn(R, C, generator=g) * 0.02
        a, b = a.to(dev), b.to(dev)
        ref = a.double() @ b.double().t()
        rms = ref.pow(2).mean().sqrt()
        el = float(((lg(a, b).double() - ref).abs() / (1e-5 * ref.abs() + 1e-5 * rms)).max())
        ef = float(((_gemm_nt(a, b).double() - ref).abs() / (1e-5 * ref.abs() + 1e-5 * rms)).max())
        worst_l, worst_f = max(worst_l, el), max(worst_f, ef)
    print(f'{dist:28s}: max err / (1e-5|y| + 1e-5 rms) over 12 seeds: limb GEMM {worst_l:.3f}, fp32-MFMA GEMM {worst_f:.3f}', flush=True)
