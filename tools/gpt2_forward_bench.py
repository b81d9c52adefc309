"""BASELINE configs 4 / 5: a GPT-2 stack (embeddings + n_layer x SPBlock + final LayerNorm; lm_head skipped -- its logits for
262 144 tokens are 52.7 GB and it is not on the quantized path) at set_precision(bits): calibration over
`--calib` micro-batches per rank with ONE all-reduce of the input statistics, then the timed forward of `--batch` sequences
per GPU in micro-batches of `--micro`.  Data-parallel: run under torch.distributed.run with one rank per GPU
(python -m torch.distributed.run --nproc-per-node N tools/gpt2_forward_bench.py ...); prints one JSON line on rank 0.
    config 4: --layers 12 --embd 768 --heads 12 --bits 4 --qtype minmax          (GPT-2-small)
    config 5: --layers 24 --embd 1024 --heads 16 --bits 6 --qtype log            (GPT-2-medium dims)"""
import argparse, json, os, sys, time, types, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llm_qat_on_gpt2_amd as pkg

ap = argparse.ArgumentParser()
ap.add_argument('--layers', type=int, default=12); ap.add_argument('--embd', type=int, default=768); ap.add_argument('--heads', type=int, default=12)
ap.add_argument('--bits', type=int, default=4); ap.add_argument('--qtype', default='minmax'); ap.add_argument('--seq', type=int, default=1024)
ap.add_argument('--batch', type=int, default=256, help='sequences in the whole job (split evenly over the ranks)')
ap.add_argument('--micro', type=int, default=32); ap.add_argument('--calib', type=int, default=10); ap.add_argument('--vocab', type=int, default=50257)
ap.add_argument('--gpus', type=int, default=0, help='started bare with --gpus N > 1: spawn the N ranks (before any GPU call)')
args = ap.parse_args()
if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    import bench                                             # the launcher of bench.py: fresh child processes, no GPU call here
    sys.exit(bench.launch_ranks(args.gpus, sys.argv[1:], script=os.path.abspath(__file__)))
world, rank, local = int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0))
torch.cuda.set_device(local)
dev = torch.device('cuda', local)
if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('nccl', device_id=dev)
E, bits, r = args.embd, args.bits, 64
cfg = types.SimpleNamespace(n_embd=E, n_head=args.heads, n_positions=args.seq, layer_norm_epsilon=1e-5, bit_widths=[bits, 32],
                            lora_rank_per_bit={bits: r, 32: 0}, lora_alpha_per_bit={bits: 64, 32: 0},
                            quantizer_per_bit={bits: args.qtype, 32: None}, per_channel_quantization=True)


class Stack(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.wte = torch.nn.Embedding(args.vocab, E); self.wpe = torch.nn.Embedding(args.seq, E)
        self.h = torch.nn.ModuleList([pkg.SPBlock(cfg, bit_widths=[bits, 32]) for _ in range(args.layers)])
        self.ln_f = pkg.SwitchableLayerNorm(E, precision_levels=[bits, 32], eps=1e-5)

    def set_precision(self, b):
        for blk in self.h: blk.set_precision(b)
        self.ln_f.set_precision(b)

    def forward(self, ids):
        x = self.wte(ids) + self.wpe(torch.arange(ids.shape[1], device=ids.device).unsqueeze(0))
        for blk in self.h: x = blk(x)
        return self.ln_f(x)


torch.manual_seed(0)                                     # replicated weights: N(0, 0.02^2), LoRA-B N(0, 0.01^2)
model = Stack()
with torch.no_grad():
    for n, p in model.named_parameters():
        if 'lora_B' in n: p.normal_(0, 0.01)
        elif p.dim() > 1 and 'lora_A' not in n: p.normal_(0, 0.02)
model = model.to(dev).eval()
gen = torch.Generator().manual_seed(1000 + rank)         # this rank's shard of the token stream
mk = lambda n: torch.randint(0, args.vocab, (n, args.seq), generator=gen).to(dev)
per_rank = args.batch // world
micro = min(args.micro, per_rank)
t0 = time.perf_counter()
exchanged = pkg.calibrate_model(model, bits, [mk(micro) for _ in range(args.calib)])
torch.cuda.synchronize(); calib_s = time.perf_counter() - t0
batches = [mk(micro) for _ in range(per_rank // micro)]
with torch.no_grad():
    model(batches[0])
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in batches: y = model(b)
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
tt = torch.tensor([dt], dtype=torch.float64, device=dev)
if world > 1: dist.all_reduce(tt, op=dist.ReduceOp.MAX)
dt = float(tt.item())
tokens = world * len(batches) * micro * args.seq
lin_flop = tokens * args.layers * sum(2 * (K * N + K * r + r * N) for K, N in ((E, 3 * E), (E, E), (E, 4 * E), (4 * E, E)))
if rank == 0:
    print(json.dumps({'workload': f'{args.layers}-layer GPT-2 stack E={E}, {bits}-bit {args.qtype} + LoRA r=64, {tokens} tokens, dp{world}',
                      'n_gpus': world, 'forward_s': round(dt, 4), 'tokens_per_s': round(tokens / dt, 1),
                      'linear_only_TFLOPs': round(lin_flop / dt / 1e12, 1), 'micro_batch': micro,
                      'calibration_s': round(calib_s, 3), 'calibration_allreduce_elements': exchanged, 'finite': bool(torch.isfinite(y).all())}))
if world > 1: dist.destroy_process_group()
