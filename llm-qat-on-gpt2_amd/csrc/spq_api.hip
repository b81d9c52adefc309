// libspq C ABI: error plumbing, device query and the fused-forward orchestrator.
#include <stdarg.h>
#include <string.h>

#include "spq_common.h"

namespace spq {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SPQ_ERR_LAUNCH;
  }
  return SPQ_OK;
}

int launch_gemm_f32_nt(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K, const float* A2,
                       int64_t lda2, const float* B2, int64_t ldb2, int64_t K2, float alpha2,
                       const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, hipStream_t st);
int fwd_f16x2(const spq_fwd_args* a, hipStream_t st);
size_t fwd_f16x2_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t r, int path);

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace spq

using namespace spq;

extern "C" int spq_version(void) { return SPQ_VERSION; }
extern "C" const char* spq_last_error(void) { return g_err; }

extern "C" int spq_device_arch(char* buf, int buflen) {
  SPQ_REQUIRE(buf && buflen > 0, "spq_device_arch: bad buffer");
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    (void)hipGetLastError();
    set_error("spq_device_arch: no HIP device");
    buf[0] = 0;
    return SPQ_ERR_DEVICE;
  }
  strncpy(buf, prop.gcnArchName, (size_t)buflen - 1);
  buf[buflen - 1] = 0;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("spq_device_arch: device is %s, this library is built for gfx950 only", prop.gcnArchName);
    return SPQ_ERR_DEVICE;
  }
  return SPQ_OK;
}

extern "C" size_t spq_fwd_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t r, int path) {
  if (M <= 0 || K <= 0 || N <= 0 || r < 0) return 0;
  if (path == SPQ_PATH_F16X2 || path == SPQ_PATH_U8X2 || path == SPQ_PATH_F16X3 || path == SPQ_PATH_I8)
    return fwd_f16x2_workspace_bytes(M, K, N, r, path);
  // F32 path: fake-quantised activations [M,K] + low-rank intermediate [M,r]
  return align_up((size_t)M * K * sizeof(float), 256) + align_up((size_t)M * (size_t)r * sizeof(float), 256) + 256;
}

extern "C" int spq_linear_lora_fwd(const spq_fwd_args* a, spq_stream_t stream) {
  SPQ_REQUIRE(a, "spq_linear_lora_fwd: null args");
  SPQ_REQUIRE(a->M > 0 && a->K > 0 && a->N > 0 && a->r >= 0, "spq_linear_lora_fwd: bad shape M=%lld K=%lld N=%lld r=%lld",
              (long long)a->M, (long long)a->K, (long long)a->N, (long long)a->r);
  SPQ_REQUIRE((a->x || a->stage == SPQ_STAGE_CONTRACTION) && a->w_prep && (a->y || a->out_levels), "spq_linear_lora_fwd: null operand");
  SPQ_REQUIRE(!a->out_levels || (a->path == SPQ_PATH_F16X2 || a->path == SPQ_PATH_F16X3),
              "spq_linear_lora_fwd: the levels-out store exists on the F16X2 / F16X3 operand paths only");
  SPQ_REQUIRE(!a->quantize_input || (a->sx && a->zx), "spq_linear_lora_fwd: input scale missing");
  SPQ_REQUIRE(!a->quantize_input || (a->bits >= 1 && a->bits <= 24), "spq_linear_lora_fwd: bits %d outside [1,24]", a->bits);
  SPQ_REQUIRE(a->r == 0 || (a->a_prep && (a->b_prep || a->t_out)), "spq_linear_lora_fwd: LoRA operands missing");
  SPQ_REQUIRE(a->workspace && aligned16(a->workspace), "spq_linear_lora_fwd: workspace missing or misaligned");
  hipStream_t st = (hipStream_t)stream;
  if (a->workspace_bytes < spq_fwd_workspace_bytes(a->M, a->K, a->N, a->r, a->path)) {
    set_error("spq_linear_lora_fwd: workspace %zu B < required %zu B", a->workspace_bytes,
              spq_fwd_workspace_bytes(a->M, a->K, a->N, a->r, a->path));
    return SPQ_ERR_WORKSPACE;
  }
  if (a->path == SPQ_PATH_F16X2 || a->path == SPQ_PATH_U8X2 || a->path == SPQ_PATH_F16X3 || a->path == SPQ_PATH_I8)
    return fwd_f16x2(a, st);
  if (a->path != SPQ_PATH_F32) {
    set_error("spq_linear_lora_fwd: unknown operand path %d", a->path);
    return SPQ_ERR_UNSUPPORTED;
  }
  SPQ_REQUIRE(a->stage == SPQ_STAGE_ALL, "spq_linear_lora_fwd: stages are split for the F16 operand paths only");
  SPQ_REQUIRE(a->epilogue == SPQ_EPILOGUE_NONE, "spq_linear_lora_fwd: the fused epilogue exists on the F16 operand paths only");
  // ---- F32 path: [x -> FQ(x)] , [t = x . FQ(A)] , [y = FQ(x) . FQ(W)^T + bias + s * t . FQ(B)]
  char* ws = (char*)a->workspace;
  float* xq = (float*)ws;
  float* t = a->t_out ? a->t_out : (float*)(ws + align_up((size_t)a->M * a->K * sizeof(float), 256));
  const bool lora_up = a->r > 0 && a->b_prep != nullptr;
  int rc;
  const float* act = a->x;
  if (a->quantize_input) {  // lora.py:141
    rc = spq_fakequant(a->x, a->M, a->K, 1, a->sx, a->zx, a->x_per_channel, a->bits, a->qtype, a->symmetric, xq,
                       nullptr, 0, stream);
    if (rc) return rc;
    act = xq;
  }
  if (a->r > 0) {  // lora.py:51 on the RAW x; cpt_model.py:112 on FQ(x)
    rc = launch_gemm_f32_nt(a->lora_on_fq_input ? act : a->x, a->K, a->a_prep, a->K, a->K, nullptr, 0, nullptr, 0, 0, 1.f,
                            nullptr, t, a->r, a->M, a->r, st);
    if (rc) return rc;
  }
  if (a->ev_gemm_begin) (void)hipEventRecord((hipEvent_t)a->ev_gemm_begin, st);
  rc = launch_gemm_f32_nt(act, a->K, (const float*)a->w_prep, a->K, a->K, lora_up ? t : nullptr, a->r,
                          (const float*)a->b_prep, a->r, lora_up ? a->r : 0, a->lora_scaling, a->bias, a->y, a->N, a->M,
                          a->N, st);
  if (a->ev_gemm_end) (void)hipEventRecord((hipEvent_t)a->ev_gemm_end, st);
  return rc;
}
