/*
 * spq.h -- C ABI of libspq.so: the MI355X (gfx950) implementation of the fake-quantized linear hot path
 * of Laurence-Wu/LLM-QAT-on-gpt2 (SPLinearWithLoRA.forward and everything it calls).
 *
 * The reference is pure Python and has no FFI layer; this header is the boundary a
 * maintainer binds with ctypes.  Each entry point names the reference code it
 * replaces (paths relative to part1_switchable_precision/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory (torch: tensor.data_ptr()); no ownership
 *     transfer, no hidden allocation, no host synchronisation; work is enqueued on `stream` (hipStream_t
 *     passed as void*; NULL = the null stream).
 *   - tensors are fp32, row-major, contiguous unless a leading dimension is given.
 *   - return value: 0 on success, a negative SPQ_ERR_* otherwise; spq_last_error() gives the message of the
 *     calling thread's last failure.
 *   - functions keep no state; they are safe to call from several threads on different streams.
 */
#ifndef SPQ_H
#define SPQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPQ_VERSION 101 /* 0.1.1 */

enum spq_status {
  SPQ_OK = 0,
  SPQ_ERR_INVALID = -1,     /* bad argument (null pointer, non-positive size, unsupported bit-width ...) */
  SPQ_ERR_LAUNCH = -2,      /* HIP reported an error at launch */
  SPQ_ERR_UNSUPPORTED = -3, /* valid request this build has no kernel for */
  SPQ_ERR_DEVICE = -4,      /* no gfx950 device / wrong architecture */
  SPQ_ERR_WORKSPACE = -5    /* workspace too small or misaligned */
};

enum spq_qtype {
  SPQ_MINMAX = 0,
  SPQ_LOG = 1,        /* part1_switchable_precision/quantization_methods.py:33-79 */
  SPQ_LOG_DIRECT = 2  /* part2_cyclic_precision_training/quantization_methods.py:23-47: same levels, the normalised level is
                         dequantised without part1's * (2^b-1) / (2^b-1) round trip */
};

/* GEMM operand paths of spq_linear_lora_fwd (see DESIGN.md "Kernels") */
enum spq_path {
  SPQ_PATH_AUTO = 0,
  SPQ_PATH_F32 = 1,   /* fp32-input MFMA on dequantised fp32 operands: always valid */
  SPQ_PATH_F16X2 = 2, /* exact integer levels (fp16) x 2-limb fp16 weights on f16 MFMA: minmax, symmetric, bits<=12 */
  SPQ_PATH_U8X2 = 3,  /* RETIRED (round 3): the byte-level ring kernel, measured slower than F16X2; spq_linear_lora_fwd returns
                         SPQ_ERR_UNSUPPORTED for it.  The value stays reserved. */
  SPQ_PATH_F16X3 = 4, /* any input quantizer (log, asymmetric, >12 bit) or none at all (quantize_input = 0: plain fp32
                         activations, e.g. a gradient): FQ(x)*2^G as two fp16 limbs x 2-limb weights, three f16 MFMA
                         products per algorithmic product; operands prepared with sx = 1 (no scale folding) */
  SPQ_PATH_I8 = 5     /* int8 matrix cores (2x the f16 rate), ONE product per algorithmic product: symmetric minmax input
                         quantizer of <= 8 bits with a PER-TENSOR scale (it then leaves the sum) and symmetric minmax weights
                         of <= 8 bits: y = (sw[n] * sx) * sum_k q[m,k] wq[n,k], the integer sum exact in i32; LoRA-up stays on
                         fp16 limbs.  (The reference's evaluation loader forces per-tensor scales: deploy.py:210,238.) */
};

enum spq_epilogue { SPQ_EPILOGUE_NONE = 0, SPQ_EPILOGUE_GELU = 1 };
enum spq_stage { SPQ_STAGE_ALL = 0, SPQ_STAGE_ACTIVATIONS = 1, SPQ_STAGE_CONTRACTION = 2 };

typedef void* spq_stream_t;

int spq_version(void);
/* Re-read the SPQ_* tuning switches from the environment (they are read once, at the first call; tests flip them in-process). */
int spq_debug_reload_switches(void);
/* Host logic only (no launch): the number of workgroups that would share one 128 x 128 tile's k range in the contraction of this
 * shape (1 = no split), under the current SPQ_SPLIT_K switch; DESIGN.md 3.3a. */
int spq_debug_split_k(int64_t M, int64_t K, int64_t N, int64_t r, int path);
const char* spq_last_error(void);
/* Writes the gcnArchName of the current device ("gfx950:sramecc+:xnack-") into buf. */
int spq_device_arch(char* buf, int buflen);

/* ---------------------------------------------------------------------------------------------------
 * Calibration statistics.  Replaces LearnableFakeQuantize._collect_statistics_batch +
 * _get_reduction_dims + _reduce_min_max (quantization.py:141-209).
 *
 * x is viewed as [outer, chan, inner]; the statistic keeps `chan` (per_channel=1) or reduces everything
 * (per_channel=0, then min_io/max_io hold 1 value).  log_domain=0: running min/max of x.
 * log_domain=1: running min/max of log2(clamp(|x|, eps)), skipped when no element has |x| > eps, except
 * that a first batch without such an element fills both with log_eps_fill (= fp32 log2(eps), :194-197).
 * first_batch!=0 assigns, otherwise merges (minimum/maximum) into min_io/max_io, in place.
 * workspace: spq_stats_workspace_bytes() bytes, 16-byte aligned.
 * ------------------------------------------------------------------------------------------------- */
size_t spq_stats_workspace_bytes(int64_t outer, int64_t chan, int64_t inner, int per_channel);
int spq_minmax_stats(const float* x, int64_t outer, int64_t chan, int64_t inner, int per_channel,
                     int log_domain, float eps, float log_eps_fill, int first_batch, float* min_io,
                     float* max_io, void* workspace, size_t workspace_bytes, spq_stream_t stream);

/* running min/max -> scale / zero_point.  Replaces LearnableFakeQuantize.finish_calibration
 * (quantization.py:104-139).  minmax symmetric: scale = clamp(max(|min|,|max|), eps)/(2^(b-1)-1), zp = 0;
 * minmax asymmetric: scale = clamp(max-min, eps)/(2^b-1), zp = round(-min/scale);
 * log: scale = max-min (log range), zp = min (log min). */
int spq_finish_scale(const float* rmin, const float* rmax, int64_t len, int bits, int qtype, int symmetric,
                     float eps, float* scale_out, float* zp_out, spq_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Standalone quantize-dequantize.  Replaces MinMaxQuantizationFunction.forward
 * (quantization_methods.py:8-22) and LogQuantizationFunction.forward (:33-79; for qtype=SPQ_LOG
 * `scale` is the log range and `zp` the log min, the argument order of quantization.py:237-239).
 *
 * x viewed as [outer, chan, inner]; scale/zp have `chan` entries (per_channel=1) or 1 entry.
 * out_f32 (nullable): dequantised values, same shape as x.  out_levels (nullable): the integer levels
 * before dequantisation, as int8/int16/int32 (levels_bytes = 1, 2 or 4).
 * transpose_out!=0 (requires outer==1... i.e. a 2-D [chan, inner] or [outer, chan] view): out_f32 is
 * written transposed, used to lay LoRA factors out K-contiguous for the GEMM.
 * ------------------------------------------------------------------------------------------------- */
int spq_fakequant(const float* x, int64_t outer, int64_t chan, int64_t inner, const float* scale,
                  const float* zp, int per_channel, int bits, int qtype, int symmetric, float* out_f32,
                  void* out_levels, int levels_bytes, spq_stream_t stream);

/* out[c, r] = fakequant(x)[r, c] for a 2-D x[rows, cols] whose scale is per column (chan = cols,
 * per_channel=1) or per tensor; the layout the GEMM wants for LoRA factors (lora.py:39-40,49-50). */
int spq_fakequant_transposed(const float* x, int64_t rows, int64_t cols, const float* scale,
                             const float* zp, int per_channel, int bits, int qtype, int symmetric,
                             float out_scaling, float* out_f32, spq_stream_t stream);

/* scale_out2 = {2^G, 2^-G} (device) with max|x| * 2^G in [2^13, 2^14): the x_limb_scale of SPQ_PATH_F16X3 for a tensor
 * that has no calibrated range -- the incoming gradient g of the backward contraction g . FQ(W)
 * (quantization_methods.py:25-28: the straight-through backward passes g unchanged).  One HBM pass over x, no host
 * synchronisation.  workspace: SPQ_LIMB_SCALE_WORKSPACE_BYTES bytes, 16-B aligned. */
#define SPQ_LIMB_SCALE_WORKSPACE_BYTES 16384
int spq_dynamic_limb_scale(const float* x, int64_t n, float* scale_out2, void* workspace, size_t workspace_bytes,
                           spq_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Token contraction of the backward pass, fp32-input MFMA:  out[i, j] = alpha * sum_m P[m, i] * Q[m, j]
 * (out [I, J] contiguous).  d/dA = s * x^T . (g . FQ(B)^T) and d/dB = s * (x . FQ(A))^T . g of LoRALayer.forward
 * (lora.py:51-52 under autograd): one side is rank-thin, M is long, so the sum is split over M and reduced in a
 * fixed order (deterministic).  workspace: spq_gemm_f32_tn_workspace_bytes() bytes, 16-B aligned.
 * ------------------------------------------------------------------------------------------------- */
size_t spq_gemm_f32_tn_workspace_bytes(int64_t M, int64_t I, int64_t J);
int spq_gemm_f32_tn(const float* P, int64_t ldp, const float* Q, int64_t ldq, int64_t M, int64_t I, int64_t J,
                    float alpha, float* out, void* workspace, size_t workspace_bytes, spq_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * SwitchableLayerNorm.forward (switchable_batchnorm.py:102-109), the producer of c_attn's and c_fc's input:
 *   mean = x.mean(-1); var = x.var(-1, unbiased=False); out = weight * ((x - mean) / sqrt(var + eps)) + bias
 * as one pass (the reference runs it as eight elementwise / reduction kernels).  x, out [rows, cols]; weight, bias [cols] of
 * the active precision.  One wavefront per row, two-pass statistics on the row held in registers (cols <= 8192).
 * ------------------------------------------------------------------------------------------------- */
int spq_layernorm(const float* x, int64_t rows, int64_t cols, const float* weight, const float* bias, float eps, float* out,
                  spq_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * The path's only exchange step: before finish_calibration every rank merges the running
 * statistics of its input quantizers (quantization.py:202-207 across ranks) with ONE in-place
 * all-reduce(MAX) over the flat fp32 buffer [-min_0 .. | max_0 ..] -- ncclAllReduce(ncclFloat, ncclMax) of
 * RCCL over xGMI.  min/max are exact and associative, so the result is bit-identical to one process seeing
 * every batch.  RCCL is bound at run time (dlopen librccl.so.1); SPQ_ERR_UNSUPPORTED if it is absent.
 *   spq_comm_unique_id : rank 0 fills a SPQ_COMM_ID_BYTES host buffer; the caller ships it to the other
 *                        ranks (any side channel: the torch.distributed store, MPI, a file).
 *   spq_comm_init      : collective; binds the calling thread's current HIP device.
 * ------------------------------------------------------------------------------------------------- */
#define SPQ_COMM_ID_BYTES 128
typedef void* spq_comm_t;
int spq_comm_unique_id(void* id_out /* host, SPQ_COMM_ID_BYTES */);
int spq_comm_init(int rank, int nranks, const void* unique_id /* host */, spq_comm_t* comm_out);
int spq_comm_destroy(spq_comm_t comm);
int spq_allreduce_minmax(spq_comm_t comm, float* neg_min_and_max /* device, in place */, size_t len,
                         spq_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Dense contraction on fp32-input MFMA (v_mfma_f32_32x32x2_f32), "NT" layout:
 *   C[m, n] = sum_k A[m,k] * B[n,k]  + bias[n]  + alpha2 * sum_j A2[m,j] * B2[n,j]
 * bias, A2/B2 nullable (K2 = 0).  Replaces F.linear (lora.py:144) and the two torch.matmul of
 * LoRALayer.forward (lora.py:51-52).  Leading dimensions in elements.
 * ------------------------------------------------------------------------------------------------- */
int spq_gemm_f32_nt(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K, const float* A2,
                    int64_t lda2, const float* B2, int64_t ldb2, int64_t K2, float alpha2,
                    const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, spq_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * The fused forward.  Replaces SPLinearWithLoRA.forward for current_bits < 32 (lora.py:133-150):
 *   y = FQ_x(x) . FQ_w(W)^T + bias + scaling * (x . FQ_A(A)) . FQ_B(B)        (LoRA on the RAW x)
 * from operands prepared once per (weights, scales) by spq_prepare_* below.
 * ------------------------------------------------------------------------------------------------- */
/* The arguments of spq_prepare_f16x2 (below) as a struct: spq_prepare_f16x2_args(&p, stream) is the same call, and a pointer
 * to one in spq_fwd_args.prepare makes the forward (re)build the weight-side operands itself -- what the reference does on
 * every call (lora.py:142 FQ(W), :49-50 FQ(A), FQ(B)) -- without a launch of their own where the shapes allow it. */
typedef struct spq_prepare_args {
  const float* W; int64_t N, K; const float* sw; const float* zw; int w_per_channel, w_bits, w_qtype, w_symmetric;
  const float* B; int64_t r; const float* sb; const float* zb; int b_per_channel, b_bits, b_qtype, b_symmetric;
  float scaling;
  const float* A; const float* sa; const float* za; int a_per_channel, a_bits, a_qtype, a_symmetric;
  const float* sx; int x_per_channel;
  void* w_prep; size_t w_prep_bytes; float* w_rowscale; float* a_prep;
  int path;  /* 0 / SPQ_PATH_F16X2 / U8X2 / F16X3: two fp16 limb planes (w_prep of spq_prep_f16x2_bytes());
                SPQ_PATH_I8: the int8 plane of weight levels (w_prep of spq_prep_bytes(N, K, r, path)) */
} spq_prepare_args;

typedef struct spq_fwd_args {
  /* problem */
  int64_t M, K, N, r;          /* r = 0: no LoRA branch (calibration_mode or disabled adapter) */
  int bits, qtype, symmetric;  /* of the INPUT quantizer */
  int quantize_input;          /* 0: x is used as is (input quantizer collecting statistics) */
  int x_per_channel;           /* sx/zx have K entries (1) or one entry (0) */
  int path;                    /* enum spq_path actually prepared for */
  /* activations */
  const float* x;              /* [M, K] */
  const float* sx;             /* input scale (log: range) */
  const float* zx;             /* input zero point (log: min) */
  const float* x_limb_scale;   /* F16X3 only: device {2^G, 2^-G}, 2^G * (bound of |FQ(x)|) in [2^13, 2^14) */
  /* prepared operands (spq_prepare_*) */
  const void* w_prep;          /* F32: fp32 FQ(W) [N,K];  F16X2: hi/lo limb planes */
  const float* w_rowscale;     /* F16X2: per-row power-of-two descale [N]; else NULL */
  const float* bias;           /* [N] or NULL */
  const float* a_prep;         /* fp32 FQ(A)^T [ceil(r/64)*64, K]; rows >= r are zero (spq_fakequant_transposed into a zeroed buffer) */
  const void* b_prep;          /* F32: fp32 FQ(B)^T [N, r]; F16X2: limb planes */
  float lora_scaling;          /* alpha / rank: multiplies the low-rank partial sum (lora.py:53) */
  /* output + scratch */
  float* y;                    /* [M, N] */
  void* workspace;
  size_t workspace_bytes;
  /* profiling hooks (nullable hipEvent_t): recorded on `stream` right before / after the dominant
   * (dense-contraction) kernel, so a benchmark can time that kernel inside its own timed region */
  void* ev_gemm_begin;
  void* ev_gemm_end;
  /* optional fp32 [M, r] copy of the LoRA-down product x . FQ(A) (lora.py:51): a training forward keeps it for d/dB, and
   * the backward takes g . FQ(B)^T out of the activation pass of its own contraction.  With r > 0, b_prep may then be
   * NULL: LoRA-down only, the contraction skips the LoRA-up stages. */
  float* t_out;
  /* 0: the LoRA branch consumes the raw x (part1 SPLinearWithLoRA, lora.py:149); 1: it consumes FQ(x) (part2 CPTLinear,
   * cpt_model.py:112) */
  int lora_on_fq_input;
  /* enum spq_stage.  The F16 paths run two launches -- the activation pass (needs x, the input scale and a_prep) and the
   * contraction (needs w_prep / b_prep).  Issuing them as two calls with the same arguments lets the caller prepare the
   * weight operands on another stream while the activation pass runs, and make `stream` wait for them in between. */
  int stage;
  /* enum spq_epilogue: SPQ_EPILOGUE_GELU stores gelu(y) (exact erf form, nn.GELU() of models_sp.py:107) instead of y --
   * the activation between mlp.c_fc and mlp.c_proj fused into c_fc's store (F16X2 / F16X3 paths). */
  int epilogue;
  /* RETIRED (round 3): must be NULL.  (It selected the LoRA-down product on the f16 matrix pipe, tools/variants/xpass_panel16.h.) */
  const float* a_limb_scale;
  /* F16 operand paths, optional: (re)make w_prep / w_rowscale / a_prep / b_prep from the fp32 weights as part of this call
   * (its buffers must be the ones named above).  In the activation stage only.  Where the streaming activation kernel runs
   * (K % 64 == 0, K <= 1024, rank <= 64) the row work rides in ITS launch as extra workgroups; otherwise the ordinary preparation
   * launch is issued first. */
  const struct spq_prepare_args* prepare;
  /* F16 / I8 operand paths, optional LayerNorm prologue: with ln_weight / ln_bias [K] set, `x` is the INPUT of
   * the SwitchableLayerNorm that precedes the layer (switchable_batchnorm.py:102-109; models_sp.py:160-171: ln_1 -> c_attn,
   * ln_2 -> c_fc) and the activation pass normalises each row on the fly -- the same arithmetic, bit for bit, as
   * spq_layernorm -- so the normalised activation is never written to or read from memory.  Needs K % 64 == 0, K <= 1024,
   * rank <= 64; SPQ_ERR_UNSUPPORTED otherwise (run spq_layernorm first then). */
  const float* ln_weight;
  const float* ln_bias;
  float ln_eps;
  /* F16X2 / F16X3 paths, optional levels-out store (second half: "GELU -> input quantizer of the next linear",
   * models_sp.py:124-128 / cpt_model.py:196-198): with out_levels set the contraction's store ALSO writes the NEXT quantized
   * layer's activation operand -- fp16 integer levels clamp(round(v / out_scale[n]), +-(2^(out_bits-1) - 1)) of the value v
   * it stores (after `epilogue`), exactly what quantization_methods.py:14-15 computes from y -- to
   * out_levels[m * out_levels_ld + n].  The consumer then runs with stage = SPQ_STAGE_CONTRACTION on a workspace that starts
   * with that level matrix (row pitch ceil(N/64)*64 = its padded K), and with y = NULL here the fp32 activation is never
   * written or re-read.  Exact wherever the consumer's LoRA branch does not read the RAW activation: r = 0, or part2's
   * CPTLinear whose branch consumes FQ(x) and is folded into the weight (cpt_model.py:112); part1's branch reads the raw
   * x (lora.py:149) and still needs y.  Needs the 128x128 contraction kernel (the default), N % 64 == 0, a symmetric
   * min-max consumer of 2..12 bits; out_scale has N entries (out_scale_per_channel) or one.  SPQ_ERR_UNSUPPORTED otherwise. */
  void* out_levels;
  int64_t out_levels_ld;
  const float* out_scale;
  int out_scale_per_channel;
  int out_bits;
  /* ... for ANY consumer quantizer (log, asymmetric, 13..24 bit: the consumer's SPQ_PATH_F16X3 operand): with out_levels_lo set
   * the store writes the two fp16 limbs of FQ(v) * 2^G -- hi to out_levels, lo to out_levels_lo (the consumer workspace's second
   * plane, ceil(M/256)*256 * ceil(N/64)*64 halves behind the first) -- with FQ the consumer's quantize-dequantize
   * (out_qtype / out_symmetric / out_bits 1..24, out_scale and out_zero both with N entries or one; log: range and min) and
   * out_limb_scale the consumer's device {2^G, 2^-G} (its x_limb_scale). */
  void* out_levels_lo;
  const float* out_zero;
  int out_qtype;
  int out_symmetric;
  const float* out_limb_scale;
} spq_fwd_args;

size_t spq_fwd_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t r, int path);
int spq_linear_lora_fwd(const spq_fwd_args* args, spq_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Weight-side operands for SPQ_PATH_F16X2 (valid when the INPUT quantizer is symmetric minmax, <= 12 bit;
 * the weight and LoRA quantizers may be of any type).  Replaces the per-call FQ(W) of lora.py:142 and FQ(B) of
 * lora.py:50, and folds the input scale sx[k] (quantization.py scale of quantizers_input) into the weight:
 *   W'[n,k] = FQ(W)[n,k] * sx[k],  B'[n,j] = scaling * FQ(B)[j,n];  both * 2^e[n], split into two fp16 limbs.
 * W [N,K]; B [r,N], A [K,r] (nullable when r = 0); B's quantizer params have N entries (per_channel) or 1, A's r or 1.
 * a_prep (out): FQ(A)^T in rows 0..r-1 of a caller-zeroed [ceil(r/64)*64, K] fp32 buffer (lora.py:49), same launch;
 * A may be NULL (with r > 0) when FQ(A)^T is produced separately (spq_fakequant_transposed).
 * w_prep: spq_prep_f16x2_bytes() bytes, 16-B aligned.  w_rowscale: ceil(N/128)*128 floats (2^-e[n]).
 * ------------------------------------------------------------------------------------------------- */
size_t spq_prep_f16x2_bytes(int64_t N, int64_t K, int64_t r);
size_t spq_prep_bytes(int64_t N, int64_t K, int64_t r, int path);   /* per operand path (an int8 plane for SPQ_PATH_I8) */
int spq_prepare_f16x2_args(const spq_prepare_args* args, spq_stream_t stream);
int spq_prepare_f16x2(const float* W, int64_t N, int64_t K, const float* sw, const float* zw,
                      int w_per_channel, int w_bits, int w_qtype, int w_symmetric, const float* B, int64_t r,
                      const float* sb, const float* zb, int b_per_channel, int b_bits, int b_qtype,
                      int b_symmetric, float scaling, const float* A, const float* sa, const float* za,
                      int a_per_channel, int a_bits, int a_qtype, int a_symmetric, const float* sx,
                      int x_per_channel, void* w_prep, size_t w_prep_bytes, float* w_rowscale, float* a_prep,
                      spq_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Weight-side operand of part2's CPTLinear (part2_cyclic_precision_training/cpt_model.py:96-113).  Its LoRA branch
 * consumes FQ(x) like the base term, so the whole layer is ONE contraction  y = FQ(x) . W_eff^T + bias  with
 *   W_eff[n,k] = FQ_w(W)[n,k] + scaling * sum_j FQ_l(B)[n,j] * FQ_l(A)[k,j]       (W [N,K], A [K,r], B [N,r], r <= 64)
 * FQ_l: the ONE quantizer part2 shares between A and B (channel_dim = 1: r or 1 parameters).  A quantizer that passes
 * its input through (32 bit, uncalibrated width) is given as bits = 32.  Outputs: w_eff [N,K] fp32 (the operand of
 * SPQ_PATH_F32 and of the backward), aq [K,r], bq [N,r], optionally aq_t = FQ(A)^T [ceil(r/64)*64, K] (the a_prep of a
 * training forward that keeps FQ(x).FQ(A) via t_out + lora_on_fq_input), and for path F16X2 / F16X3 the limb planes
 * w_prep + w_rowscale exactly as spq_prepare_f16x2 writes them (sx folded for F16X2; pass the input scale, or a
 * one-element tensor holding 1.0 with x_per_channel = 0 for F16X3).
 * ------------------------------------------------------------------------------------------------- */
int spq_prepare_cpt(const float* W, int64_t N, int64_t K, const float* sw, const float* zw, int w_per_channel, int w_bits,
                    int w_qtype, int w_symmetric, const float* A, const float* B, int64_t r, const float* sl,
                    const float* zl, int l_per_channel, int l_bits, int l_qtype, int l_symmetric, float scaling,
                    const float* sx, int x_per_channel, int path, void* w_prep, size_t w_prep_bytes, float* w_rowscale,
                    float* w_eff, float* aq, float* bq, float* aq_t, spq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPQ_H */
