"""Times the activation pass alone (stage = SPQ_STAGE_ACTIVATIONS of spq_linear_lora_fwd) at the headline shape, one child
process per library variant (SPQ_LIB), so that diagnostic builds of the kernel (tools/build_variants.sh, -DSPQ_XP_DIAG=...) can be
compared on one box.      python tools/xpass_probe.py [variant.so ...]      (kernel tuning only)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import torch
    sys.path.insert(0, ROOT)
    import llm_qat_on_gpt2_amd as pkg
    from llm_qat_on_gpt2_amd import synthetic as S
    dev = "cuda:0"
    M, K, N, r, bits = int(os.environ.get("XP_M", 8192)), int(os.environ.get("XP_K", 768)), int(os.environ.get("XP_N", 3072)), 64, int(os.environ.get("XP_BITS", 4))
    qtype = os.environ.get("XP_QTYPE", "minmax")
    W, bias, A, B, _, _ = S.make_workload(8, K, N, r, seed=0)
    layer = pkg.SPLinearWithLoRA(K, N, [bits, 32], {bits: r, 32: 0}, {bits: 64, 32: 0}, {bits: qtype, 32: None}, per_channel=True)
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters[f"{bits}bit"].lora_A.copy_(A); layer.lora_adapters[f"{bits}bit"].lora_B.copy_(B)
    layer = layer.to(dev).eval(); layer.set_precision(bits); layer.cache_operands = True
    g = torch.Generator().manual_seed(5)
    mk = lambda: torch.randn(M, K, generator=g).to(dev)
    pkg.calibrate_layer(layer, bits, [mk(), mk()])
    x = mk()
    lib = pkg._lib.load()
    real = lib.spq_linear_lora_fwd
    only = [False]

    last = []

    def fwd(argref, st):
        if only[0]: argref._obj.stage = pkg._lib.STAGE_ACTIVATIONS
        last[:] = [argref, st]
        return real(argref, st)
    lib.spq_linear_lora_fwd = fwd

    def direct(stage, n=400):                               # the C entry point back to back, without the Python layer around it
        argref, st = last
        argref._obj.stage = stage
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(50): real(argref, st)
        a.record()
        for _ in range(n): real(argref, st)
        b.record(); b.synchronize()
        return a.elapsed_time(b) / n * 1e3

    def timeit(n=300):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(50): layer(x)
        a.record()
        for _ in range(n): layer(x)
        b.record(); b.synchronize()
        return a.elapsed_time(b) / n * 1e3
    with torch.no_grad():
        full = timeit()
        act = direct(pkg._lib.STAGE_ACTIVATIONS); con = direct(pkg._lib.STAGE_CONTRACTION); both = direct(0)
    if hasattr(lib, "spq_debug_xp_stamps"):                  # -DSPQ_XP_DIAG=128 builds: per-wave segment sums of the streaming kernel
        import ctypes, numpy as np
        nwg = min(2048, (M + 31) // 32)
        buf = (ctypes.c_ulonglong * (nwg * 8 * 4))()
        torch.cuda.synchronize()
        rc = lib.spq_debug_xp_stamps(buf, nwg * 8 * 4)
        st = np.frombuffer(buf, dtype=np.uint64).reshape(nwg, 8, 4).astype(np.float64)
        nck = K // 64
        print(f"stamps rc={rc} (s_memtime ticks; 100 MHz clock: x10 ns), mean over {nwg} workgroups, per chunk:  wait-for-copies {st[:, :, 0].mean() / nck:7.1f}  barrier "
              f"{st[:, :, 1].mean() / nck:7.1f}  body {st[:, :, 2].mean() / nck:7.1f}  | loop total {st[:, :, 3].mean():8.1f}")
        for w in range(8):
            print(f"   wave {w}: wait {st[:, w, 0].mean() / nck:7.1f} barrier {st[:, w, 1].mean() / nck:7.1f} body {st[:, w, 2].mean() / nck:7.1f}")
    print(f"{os.path.basename(os.environ.get('SPQ_LIB', 'libspq.so')):20s} layer (cached weight operands) {full:6.1f} us | C entry point back to back: "
          f"activation pass {act:6.1f} us, contraction {con:6.1f} us, both {both:6.1f} us", flush=True)


if __name__ == "__main__":
    if os.environ.get("XP_CHILD"):
        child()
    else:
        for v in [""] + sys.argv[1:]:
            env = dict(os.environ, XP_CHILD="1")
            if v: env["SPQ_LIB"] = os.path.abspath(v)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
