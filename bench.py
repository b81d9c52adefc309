#!/usr/bin/env python3
"""Benchmark of the hot path: SPLinearWithLoRA.forward on GPT-2-small c_fc (768 -> 3072), 4-bit minmax
per-channel, LoRA rank 64, batch 8 x seq 1024 tokens per GPU (BASELINE.json metric; SURVEY.md §8d headline).

    python bench.py --gpus N --steps K --warmup W

N > 1 either way: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the ranks read
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or started bare -- then this process, BEFORE it touches the
GPU, starts the N ranks itself as fresh child processes (one per device, rendezvous on 127.0.0.1) and exits with the worst
child's code.

A step = one forward of the layer over one batch of synthetic activations already resident in HBM, including the
per-call weight-side fake-quant that the reference also performs on every call (unless --hoist-weights).  Ranks are
data-parallel replicas over the batch (weak scaling); the only collective is the all-reduce of the calibration
min/max statistics, done once before the timed region.  Rank 0 prints ONE JSON line.

--dry-run: rehearsal of the launcher and the exchange step without a GPU (gloo): the ranks form, each derives the statistics of
its own activation shard with plain torch ops, the product's single all-reduce merges them, the timing protocol runs over
empty steps and rank 0 prints the line with "dry_run": true and no throughput.  tests/test_bench_launcher.py uses it.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M_TOKENS, K_IN, N_OUT, RANK, BITS, BATCH = 8192, 768, 3072, 64, 4, 8
FLOP_PER_STEP = 2 * M_TOKENS * K_IN * N_OUT + 2 * M_TOKENS * K_IN * RANK + 2 * M_TOKENS * RANK * N_OUT   # 42 681 237 504
BYTES_PER_STEP = 4 * (M_TOKENS * K_IN + N_OUT * K_IN + K_IN * RANK + RANK * N_OUT + N_OUT + M_TOKENS * N_OUT) \
    + 4 * (K_IN + N_OUT + RANK + N_OUT)                                                                        # ~136.29 MB
PEAK = {"f32": 157.3, "f16": 2500.0, "i8": 5000.0, "hbm_gbs": 8000.0}   # MI355X_MICROARCH.md: dense MFMA TFLOP/s, HBM GB/s
SETUP_STEPS = 2000        # untimed, before the W warm-ups: grows the workspaces and lets the clocks settle (~0.2 s; not part of W or K).
                          # Measured (tools/setup_steps_ab.sh, one box, --steps 20 --warmup 5): 200 -> 0.0994 / 0.1016 / 0.1013 ms per step in the
                          # first timed region, 2000 -> 0.0963 / 0.0965: after 20 ms of work the chip has not reached the clock it then holds.
STAT_REPEATS = 5          # extra repetitions of the K-step region for the median / min figure


# ---------------------------------------------------------------------------------------------------------------------
# launcher: N fresh rank processes, started before this process has made any GPU call
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, script=None):
    """Start `n` copies of this script as children (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set), wait, return the worst exit code.
    Nothing in this process has initialised HIP at this point (importing torch does not), and the children are fresh
    interpreters, so no process that holds a GPU context ever execs or forks."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for rank in range(n):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, SPQ_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env))
    worst = 0
    try:
        live = list(procs)
        while live:                                  # poll ALL ranks: a rank that dies while an earlier one sits in a collective
            for p in list(live):                     # must be noticed at once, not after that collective's time-out
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and worst == 0:
                    worst = rc
                    for q in live:                   # one rank died: the others would wait in a collective forever
                        q.terminate()
            if live:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return worst


class HipEvents:
    """Raw hipEvent pairs (the C ABI records them around the dominant kernel on the launch stream)."""

    def __init__(self, n):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.pairs = []
        for _ in range(n):
            b, e = ctypes.c_void_p(), ctypes.c_void_p()
            # hipEventDisableSystemFence: the record is a timestamp marker on the stream without the system-scope cache release a
            # default event carries (nothing on the host reads data the kernel wrote; only the two timestamps are compared)
            flags = 0x20000000
            assert self.hip.hipEventCreateWithFlags(ctypes.byref(b), flags) == 0 and self.hip.hipEventCreateWithFlags(ctypes.byref(e), flags) == 0
            self.pairs.append((b, e))

    def elapsed_ms(self):
        out = []
        for b, e in self.pairs:
            ms = ctypes.c_float()
            if self.hip.hipEventElapsedTime(ctypes.byref(ms), b, e) == 0:
                out.append(ms.value)
        return out

    def destroy(self):
        for b, e in self.pairs:
            self.hip.hipEventDestroy(b); self.hip.hipEventDestroy(e)


def _time_forwards(fn, budget_s, min_n, max_n):
    times, t_end = [], time.perf_counter() + budget_s
    while len(times) < min_n or (time.perf_counter() < t_end and len(times) < max_n):
        t0 = time.perf_counter(); fn(); times.append(time.perf_counter() - t0)
    times.sort()
    return times


def _cpu_share():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands each job a share
    of its 256 logical CPUs; torch's default of one thread per physical core then oversubscribes it several times over)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except Exception:
            continue
    return n


def cpu_baseline_and_parity(weights, calib, x, y_gpu, x_levels_gpu, w_levels_gpu, budget_s=10.0):
    """The CPU leg: the oracle -- torch CPU ops in the reference's op order, bit-identical to the imported
    reference on the golden fixtures -- (i) timed on this host's cores at all threads and at one thread on the very tensors the
    GPU was timed on, and (ii) used as the checker of the GPU result of the timed region (the parity gate of the same run)."""
    from oracle import ref_cpu as O
    W, bias, A, B = weights
    layer = O.build_calibrated_layer(W, bias, A, B, calib, BITS, "minmax", True, 64, RANK)
    model = ""
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    with torch.no_grad():
        y_ref = layer.forward(x)
        layer.forward(x)
        n_default = torch.get_num_threads()
        n_all = max(1, min(n_default, _cpu_share()))
        torch.set_num_threads(n_all)
        layer.forward(x)
        t_all = _time_forwards(lambda: layer.forward(x), budget_s, 5, 60)
        torch.set_num_threads(1)
        layer.forward(x)
        t_one = _time_forwards(lambda: layer.forward(x), budget_s, 2, 5)
        torch.set_num_threads(n_default)
        # parity gate: SURVEY.md §8d -- levels exact, |dy| <= 1e-5 |y_ref| + 1e-5 rms(y_ref)
        yd, yr = y_gpu.double(), y_ref.double().reshape(y_gpu.shape)
        rms = float(yr.pow(2).mean().sqrt())
        ratio = (yd - yr).abs() / (1e-5 * yr.abs() + 1e-5 * rms)
        rel = (yd - yr).abs() / (yr.abs() + rms)
        x_lv = layer.qx.levels(x).to(torch.int32).reshape(x_levels_gpu.shape)
        w_lv = layer.qw.levels(W).to(torch.int32)
        parity = {"checked_against": "oracle/ref_cpu.py (CPU restatement, bit-identical to the reference on tests/golden)",
                  "tolerance": "|dy| <= 1e-5*|y_ref| + 1e-5*rms(y_ref)", "max_err_over_bound": round(float(ratio.max()), 4),
                  "pass": bool(ratio.max() <= 1.0), "max_rel_err": float(f"{float(rel.max()):.3e}"),
                  "mean_rel_err": float(f"{float(rel.mean()):.3e}"),
                  "level_mismatches": {"activation": int((x_lv != x_levels_gpu).sum()), "weight": int((w_lv != w_levels_gpu).sum()),
                                       "of": [x_lv.numel(), w_lv.numel()]}}
    med = t_all[len(t_all) // 2]
    med1 = t_one[len(t_one) // 2]
    base = {"value": round(FLOP_PER_STEP / med / 1e9, 2), "unit": "GFLOP/s", "cores": n_all, "kind": "port",
            "sample": f"{len(t_all)} full forwards of the same workload (M={M_TOKENS}), median {med * 1e3:.1f} ms, "
                      f"min {t_all[0] * 1e3:.1f} ms; host cpu_count={os.cpu_count()}, usable share {_cpu_share()} [{model}]",
            "min_ms": round(t_all[0] * 1e3, 1), "median_ms": round(med * 1e3, 1),
            "one_thread": {"value": round(FLOP_PER_STEP / med1 / 1e9, 2), "unit": "GFLOP/s", "cores": 1,
                           "sample": f"{len(t_one)} full forwards, median {med1 * 1e3:.0f} ms, min {t_one[0] * 1e3:.0f} ms"}}
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--path", choices=["auto", "f32", "f16x2", "f16x3", "i8"], default="auto")
    ap.add_argument("--hoist-weights", action="store_true",
                    help="reuse prepared weight operands across steps (eval-mode behaviour of the module)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--setup-steps", type=int, default=SETUP_STEPS,
                    help="untimed forwards BEFORE the --warmup steps (workspace growth, code-object load, clock ramp); 0 = the "
                         "timed region starts after exactly --warmup warm-ups")
    ap.add_argument("--calib-comm", choices=["torch", "capi"], default="torch",
                    help="calibration all-reduce through torch.distributed (backend nccl = RCCL) or through libspq's own "
                         "RCCL binding (spq_comm_init / spq_allreduce_minmax)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: rehearse launcher, rank formation (gloo), the one calibration all-reduce and the timing "
                         "protocol; prints the line with dry_run=true and no throughput")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))            # no GPU call has happened in this process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (either start it bare, or with torch.distributed.run "
                 f"--nproc-per-node {args.gpus})")
    if args.dry_run:
        return dry_run(args, world, rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback; --dry-run rehearses the launch)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import llm_qat_on_gpt2_amd as pkg
    from llm_qat_on_gpt2_amd import calibration
    from llm_qat_on_gpt2_amd import synthetic as O   # seeded input generator (the CPU leg's oracle draws the same tensors)

    W, bias, A, B, _, _ = O.make_workload(8, K_IN, N_OUT, RANK, seed=0)          # replicated weights
    gen = torch.Generator().manual_seed(1000 + 17 * rank)                        # this rank's batch shard

    def act():
        x = torch.randn(M_TOKENS, K_IN, generator=gen)
        x = torch.where(torch.rand(M_TOKENS, K_IN, generator=gen) < 1e-3, x * 20, x)
        return x.view(BATCH, M_TOKENS // BATCH, K_IN)

    layer = pkg.SPLinearWithLoRA(K_IN, N_OUT, [BITS, 32], {BITS: RANK, 32: 0}, {BITS: 64, 32: 0},
                                 {BITS: "minmax", 32: None}, per_channel=True)
    key = f"{BITS}bit"
    with torch.no_grad():
        layer.linear.weight.copy_(W); layer.linear.bias.copy_(bias)
        layer.lora_adapters[key].lora_A.copy_(A); layer.lora_adapters[key].lora_B.copy_(B)
    layer = layer.to(dev).eval()
    layer.set_precision(BITS)
    layer.operand_path = {"auto": pkg._lib.PATH_AUTO, "f32": pkg._lib.PATH_F32, "f16x2": pkg._lib.PATH_F16X2,
                          "f16x3": pkg._lib.PATH_F16X3, "i8": pkg._lib.PATH_I8}[args.path]
    layer.cache_operands = bool(args.hoist_weights)

    # calibration: 2 local batches per rank, then ONE all-reduce(MAX) of [-min | max] (RCCL) -> identical scales
    calib_host = [act(), act()]
    calib_dev = [c.to(dev) for c in calib_host]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    comm = None
    if args.calib_comm == "capi":
        comm = pkg.SpqComm.from_process_group() if world > 1 else pkg.SpqComm(0, 1, pkg.SpqComm.unique_id())
    exchanged = pkg.calibrate_layer(layer, BITS, calib_dev, comm=comm)
    torch.cuda.synchronize()
    calib_ms = (time.perf_counter() - t0) * 1e3
    del calib_dev

    x_host = act()
    x = x_host.to(dev)

    def timed_region(n):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            out = layer(x)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out

    with torch.no_grad():
        for _ in range(max(0, args.setup_steps)):          # setup, not warm-up: workspace growth, code-object load, clock ramp
            y = layer(x)
        # the dominant kernel is timed live inside the timed region, on every EVERY-th step (an event pair costs two marker
        # packets on the launch stream; on every step that alone took 8 % off the throughput it was meant to explain); EVERY
        # is chosen so that even a 20-step run averages over >= 8 launches
        EVERY = int(os.environ.get('SPQ_BENCH_EVENT_EVERY', '8'))
        if args.steps < 64 and 'SPQ_BENCH_EVENT_EVERY' not in os.environ:
            # a short region (the driver's 20 steps): ONE event pair inside it -- each pair is two marker packets on the launch
            # stream, ~20 us of a 2 ms region apiece -- and the other >= 7 samples on the first repeat region below
            EVERY = max(EVERY, args.steps)
        n_main = (args.steps + EVERY - 1) // EVERY
        n_extra = max(0, 8 - n_main)      # a short run: further samples ride on the first repeat region below, >= 8 launches in all
        ev = HipEvents(n_main + n_extra)
        import gc
        gc.collect(); gc.disable()        # a collector pause of the host inside a 2-ms region is a 10-20 % outlier (seen: 0.119 against 0.097 ms/step)
        # All host-side preparation (events, the collector) happens BEFORE the W warm-up steps, so that the timed region follows GPU work,
        # not an idle gap: the chip drops its clock within milliseconds of idling and the first ~1 ms after that runs slow -- with the
        # event creation between warm-up and region the FIRST region read 0.110-0.119 ms/step where the repeats right after it read 0.100.
        for _ in range(args.warmup):
            y = layer(x)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            # (a sample sits on the LAST step of its group of EVERY: in a 20-step region the one pair then lands at the region's end, where its two
            # marker packets delay nothing behind them -- at step 0 they sat in the launch ramp of the whole region)
            layer._gemm_events = ev.pairs[i // EVERY] if (i % EVERY == EVERY - 1 or i == args.steps - 1) and i // EVERY < n_main else None
            y = layer(x)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        gc.enable()
        layer._gemm_events = None
        # median / min over repeats of the same K-step region (no event markers), SURVEY.md §8d
        repeats = []
        for rep in range(STAT_REPEATS):
            if rep == 0 and n_extra:       # same K-step protocol, event pairs on its first launches
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    layer._gemm_events = ev.pairs[n_main + i] if i < n_extra else None
                    y = layer(x)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize()
                repeats.append(time.perf_counter() - t0)
                layer._gemm_events = None
            else:
                repeats.append(timed_region(args.steps)[0])
        repeats.sort()
    # secondary figure (not `value`): the module's eval-mode behaviour, weight-side operands reused while W/A/B and the
    # scales are unchanged; same protocol, same K
    elapsed_cached = None
    if not args.hoist_weights:
        layer.cache_operands = True
        with torch.no_grad():
            for _ in range(max(3, args.warmup // 4)):
                y2 = layer(x)
            elapsed_cached, y2 = timed_region(args.steps)
        assert torch.equal(y2, y), "cached-operand forward differs from the re-quantising forward"
        layer.cache_operands = False
    # secondary figure: L2 / Infinity-Cache cold (a 512 MB write between forwards evicts activations, weights and operands)
    cold_ms = None
    if world == 1:
        flush = torch.empty(512 * 1024 * 1024 // 4, device=dev)
        samples = []
        with torch.no_grad():
            for _ in range(12):
                flush.fill_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); layer(x); e1.record(); e1.synchronize()
                samples.append(e0.elapsed_time(e1))
        samples.sort()
        cold_ms = samples[len(samples) // 2]
        del flush
    t = torch.tensor([elapsed, elapsed_cached or 0.0] + repeats, dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0].item())
    if elapsed_cached is not None:
        elapsed_cached = float(t[1].item())
    repeats = [float(v) for v in t[2:].tolist()]
    gemm_ms = ev.elapsed_ms()
    ev.destroy()
    path_used = layer._last_path
    assert bool(torch.isfinite(y).all())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * FLOP_PER_STEP * args.steps / elapsed / 1e9
        gemm_avg_ms = sum(gemm_ms) / max(1, len(gemm_ms))
        is_f16 = path_used in (pkg._lib.PATH_F16X2, pkg._lib.PATH_F16X3)
        is_i8 = path_used == pkg._lib.PATH_I8
        achieved = FLOP_PER_STEP / (gemm_avg_ms * 1e-3) / 1e12 if gemm_avg_ms > 0 else 0.0
        peak = PEAK["i8"] if is_i8 else (PEAK["f16"] if is_f16 else PEAK["f32"])
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                rec = json.load(open(tpath)).get("i8" if is_i8 else ("f16x2" if is_f16 else "f32"), {})
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_source = (f"profiles/pmc_traffic.json ({rec.get('kernel', '?')}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                  f"passes of {rec.get('measured', 'an earlier run')}, not this run)")
            except Exception:
                traffic = None
        kernel_name = ("gemm_i8_k128_kernel (dense contraction on v_mfma_i32_32x32x32_i8 + LoRA-up on f16 limbs + bias)" if is_i8 else
                       "gemm_f16x2_t128_kernel (dense contraction + LoRA-up + bias, v_mfma_f32_16x16x32_f16)" if is_f16
                       else "gemm_f32_nt (dense contraction + LoRA-up + bias)")
        dtype_s = ("i8 (int8 levels x int8 weight levels, i32 accumulate: exact)" if is_i8 else
                   "f16x2-limb operands, f32 accumulate (fp32-accurate)" if is_f16 else "f32")
        peak_s = ("i8 dense MFMA (2x the f16 rate; one product per algorithmic product)" if is_i8 else
                  "f16 dense MFMA (2 limb products per algorithmic product)" if is_f16 else "f32-input MFMA")
        # a yardstick, not part of the product: the SAME limb products as one plain fp16 GEMM of the vendor library (torch.mm -> hipBLASLt) --
        # [q | q] . [Whi | Wlo]^T and [thi | thi | tlo] . [Bhi | Blo | Bhi]^T concatenated along K -- timed live on this box (rank 0, one GPU)
        yard = None
        if world == 1 and is_f16 and not is_i8:
            try:
                kq = 2 * K_IN + 3 * ((RANK + 63) // 64 * 64)
                ya = torch.randint(-7, 8, (M_TOKENS, kq), device=dev).half()
                yb = (torch.randn(N_OUT, kq, device=dev) * 100).half()
                for _ in range(10):
                    torch.mm(ya, yb.t(), out_dtype=torch.float32)
                torch.cuda.synchronize()
                y0, y1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                y0.record()
                for _ in range(50):
                    torch.mm(ya, yb.t(), out_dtype=torch.float32)
                y1.record(); y1.synchronize()
                yard = {"ms": round(y0.elapsed_time(y1) / 50, 4),
                        "what": f"torch.mm (hipBLASLt) fp16 [{M_TOKENS} x {kq}] . [{kq} x {N_OUT}] -> fp32: the contraction kernel's limb products (2 per "
                                "weight, 3 per LoRA-B element) as ONE plain GEMM, without its rescale / row scale / bias; a reference, not the product"}
                del ya, yb
            except Exception as ex:                       # (an older torch without out_dtype: skip the yardstick)
                yard = {"ms": None, "what": f"not measured: {type(ex).__name__}"}
        out = {
            "metric": "fused quant-GEMM-LoRA fwd GFLOP/s per GPU, GPT-2 c_fc 768→3072 @ 4-bit",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "setup_steps": args.setup_steps, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_s,
            "data": "synthetic",
            "config": {"workload": "SPLinearWithLoRA c_fc 768->3072, 4-bit minmax per-channel + LoRA r=64, "
                                   "batch 8 x seq 1024 = 8192 tokens per GPU (SURVEY.md 8d headline)",
                       "tokens_per_gpu": M_TOKENS, "parallelism": f"dp{world} (replicas over the batch)",
                       "operand_path": pkg._lib.PATH_NAMES.get(path_used, str(path_used)),
                       "weights_requantized_every_step": not args.hoist_weights,
                       "flop_per_step_per_gpu": FLOP_PER_STEP, "algorithmic_bytes_per_step_per_gpu": BYTES_PER_STEP},
            "value_per_gpu": round(value / world, 1),
            "ms_per_step_stats": {"median": round(repeats[len(repeats) // 2] / args.steps * 1e3, 4),
                                  "min": round(repeats[0] / args.steps * 1e3, 4), "max": round(repeats[-1] / args.steps * 1e3, 4),
                                  "repeats": len(repeats),
                                  "note": f"{len(repeats)} further repeats of the {args.steps}-step region (max over ranks each), "
                                          "after the region `value` is taken from"},
            "algorithmic_GBps_per_gpu": round(BYTES_PER_STEP * args.steps / elapsed / 1e9, 1),
            "frac_of_hbm_peak": round(BYTES_PER_STEP * args.steps / elapsed / 1e9 / PEAK["hbm_gbs"], 4),
            "calibration": {"ms": round(calib_ms, 2), "allreduce_elements": exchanged, "comm": args.calib_comm,
                            "allreduce_ms": (None if not exchanged else round(calibration.LAST_EXCHANGE["allreduce_ms"], 4)),
                            "note": "allreduce_ms: HIP events around the ONE data collective on rank 0 (null at one GPU: no collective)"},
            "with_cached_weight_operands": None if elapsed_cached is None else {
                "ms_per_step": round(elapsed_cached / args.steps * 1e3, 4),
                "value": round(world * FLOP_PER_STEP * args.steps / elapsed_cached / 1e9, 1),
                "note": "eval-mode module: FQ(W), FQ(A), FQ(B) operands reused while unchanged; bit-identical output"},
            "cold_caches": None if cold_ms is None else {
                "ms_per_step": round(cold_ms, 4), "value": round(FLOP_PER_STEP / cold_ms / 1e6, 1),
                "note": "median of 12 single forwards, each after a 512 MB write that evicts L2 and Infinity Cache; event-timed"},
            "roofline": {"bound": "mfma", "kernel": kernel_name,
                         "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_launches_timed": len(gemm_ms), "peak_dtype": peak_s, "kernel_ms_avg": round(gemm_avg_ms, 4),
                         "frac_vs_f16_mfma_peak": round(achieved / PEAK["f16"], 4),
                         "kernel_ms_min": round(min(gemm_ms), 4) if gemm_ms else None,
                         "vendor_gemm_same_limb_products": yard,
                         "frac_vs_f32_mfma_peak": round(achieved / PEAK["f32"], 4),
                         # the contraction on its OWN flops (2MKN + 2MrN: the LoRA-down product 2MKr runs in the activation pass)
                         "achieved_own_flops": round((FLOP_PER_STEP - 2 * M_TOKENS * K_IN * RANK) / (gemm_avg_ms * 1e-3) / 1e12, 2) if gemm_avg_ms > 0 else None,
                         # the rest of a step: activation pass (+ weight rows) and FQ(A)^T -- HBM-bound helpers.  bytes = what they must
                         # move (x read 4MK, levels written 2MK, t limbs 4Mr + row scales, W read 4NK + limbs written 4NK, A/B small)
                         "helpers": (lambda hb, hms: {"ms": round(hms, 4), "bytes": hb, "GBps": round(hb / (hms * 1e-3) / 1e9, 1) if hms > 0 else None,
                                                      "frac_of_hbm_peak": round(hb / (hms * 1e-3) / 1e9 / PEAK["hbm_gbs"], 4) if hms > 0 else None,
                                                      "note": "ms = ms_per_step - kernel_ms_avg (launch gaps included)"})(
                             6 * M_TOKENS * K_IN + 4 * M_TOKENS * RANK + 4 * M_TOKENS + (0 if args.hoist_weights else 8 * N_OUT * K_IN + 12 * RANK * (K_IN + N_OUT)),
                             ms_per_step - gemm_avg_ms)},
        }
        if world == 1 and not args.no_cpu_baseline:
            with torch.no_grad():
                x_lv = layer.quantizers_input[key].quantize_levels(x).cpu()
                w_lv = layer.quantizers_weight[key].quantize_levels(layer.linear.weight.detach()).cpu()
            out["cpu_baseline"], out["parity"] = cpu_baseline_and_parity((W, bias, A, B), calib_host, x_host, y.cpu(), x_lv, w_lv)
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.destroy()
    if world > 1:
        dist.destroy_process_group()


def dry_run(args, world, rank):
    """Launcher / collective rehearsal on the CPU (gloo).  The statistics kernel is HIP-only, so each rank derives the min/max
    of its shard with plain torch reductions and hands them to the product's merge (`allreduce_calibration_stats`): what is
    exercised is rank formation, the ONE all-reduce(MAX) of [-min | max], the barrier + max-over-ranks timing and the single
    JSON line -- not the kernels."""
    import llm_qat_on_gpt2_amd as pkg
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    gen = torch.Generator().manual_seed(1000 + 17 * rank)
    q = pkg.LearnableFakeQuantize(BITS, channel_dim=-1, quantizer_type="minmax", is_input=True)
    q.start_calibration()
    for _ in range(2):
        xb = torch.randn(BATCH, 64, K_IN, generator=gen)
        lo, hi = xb.amin(dim=(0, 1), keepdim=True), xb.amax(dim=(0, 1), keepdim=True)
        q.temp_min = lo if q.temp_min is None else torch.minimum(q.temp_min, lo)
        q.temp_max = hi if q.temp_max is None else torch.maximum(q.temp_max, hi)
        q.num_batches_collected += 1
    mine = (q.temp_min.clone(), q.temp_max.clone())
    exchanged = pkg.allreduce_calibration_stats([q])
    assert bool((q.temp_min <= mine[0]).all()) and bool((q.temp_max >= mine[1]).all())
    for _ in range(args.warmup):
        pass
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "fused quant-GEMM-LoRA fwd GFLOP/s per GPU, GPT-2 c_fc 768→3072 @ 4-bit", "value": None,
                          "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "calibration": {"allreduce_elements": exchanged, "comm": "torch(gloo)"},
                          "config": {"workload": "launcher + calibration-exchange rehearsal, no kernels",
                                     "parallelism": f"dp{world} (replicas over the batch)"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
