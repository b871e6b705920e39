// Version / error-string entry points of libtmdiff_hip.so.
#include "common.h"

namespace tmdiff {
char* last_error_buf() {
  static thread_local char buf[512] = "";
  return buf;
}
}  // namespace tmdiff

extern "C" int tmdiff_version(void) { return TMDIFF_ABI_VERSION; }
extern "C" const char* tmdiff_last_error_string(void) { return tmdiff::last_error_buf(); }

// ---- per-operator names (include/tmdiff_hip.h, "minimum exports") --------------------------------------------------
namespace {
int conv_with_ksize(const tmdiff_conv3d_desc* d, int ks, const char* who, tmdiff_stream_t stream) {
  if (!d) return tmdiff::fail(TMDIFF_E_INVALID, "%s: null descriptor", who);
  if (d->ksize != ks) return tmdiff::fail(TMDIFF_E_INVALID, "%s: descriptor has ksize %d", who, d->ksize);
  return tmdiff_conv3d_fwd(d, stream);
}
int wgrad_with_ksize(const tmdiff_conv3d_desc* d, int ks, const char* who, const float* g, float* dw, void* ws,
                     tmdiff_stream_t stream) {
  if (!d) return tmdiff::fail(TMDIFF_E_INVALID, "%s: null descriptor", who);
  if (d->ksize != ks) return tmdiff::fail(TMDIFF_E_INVALID, "%s: descriptor has ksize %d", who, d->ksize);
  return tmdiff_conv3d_wgrad(d, g, dw, ws, stream);
}
}  // namespace

extern "C" {
int tmdiff_conv3d_k3_fwd(const tmdiff_conv3d_desc* d, tmdiff_stream_t s) { return conv_with_ksize(d, 3, "conv3d_k3_fwd", s); }
int tmdiff_conv3d_k3_dgrad(const tmdiff_conv3d_desc* d, tmdiff_stream_t s) { return conv_with_ksize(d, 3, "conv3d_k3_dgrad", s); }
int tmdiff_conv3d_k1_fwd(const tmdiff_conv3d_desc* d, tmdiff_stream_t s) { return conv_with_ksize(d, 1, "conv3d_k1_fwd", s); }
int tmdiff_conv3d_k1_dgrad(const tmdiff_conv3d_desc* d, tmdiff_stream_t s) { return conv_with_ksize(d, 1, "conv3d_k1_dgrad", s); }
int tmdiff_conv3d_k3_wgrad(const tmdiff_conv3d_desc* d, const float* g, float* dw, void* ws, tmdiff_stream_t s) {
  return wgrad_with_ksize(d, 3, "conv3d_k3_wgrad", g, dw, ws, s);
}
int tmdiff_conv3d_k1_wgrad(const tmdiff_conv3d_desc* d, const float* g, float* dw, void* ws, tmdiff_stream_t s) {
  return wgrad_with_ksize(d, 1, "conv3d_k1_wgrad", g, dw, ws, s);
}

int tmdiff_haar_dwt2d_fwd(const float* x, float* ll, float* lh, float* hl, float* hh, int64_t planes, int32_t H,
                          int32_t W, float ll_scale, float hi_scale, tmdiff_stream_t s) {
  return tmdiff_haar_dwt2d(x, ll, lh, hl, hh, planes, H, W, ll_scale, hi_scale, s);
}
int tmdiff_haar_dwt2d_bwd(const float* g_ll, const float* g_lh, const float* g_hl, const float* g_hh, float* dx,
                          int64_t planes, int32_t H, int32_t W, float ll_scale, float hi_scale, tmdiff_stream_t s) {
  const int n_hi = (g_lh != nullptr) + (g_hl != nullptr) + (g_hh != nullptr);
  if (n_hi != 0 && n_hi != 3) return tmdiff::fail(TMDIFF_E_INVALID, "haar_dwt2d_bwd: give all three high-band gradients or none");
  if (n_hi == 3 && hi_scale != 1.f)
    return tmdiff::fail(TMDIFF_E_UNSUPPORTED, "haar_dwt2d_bwd: hi_scale must be 1 (only the LL band is scaled on the hot path)");
  const float* lls[2] = {g_ll, nullptr};
  float* outs[2] = {dx, nullptr};
  return tmdiff_haar_idwt2d(lls, 1, g_lh, g_hl, g_hh, 0, 0, outs, planes, H / 2, W / 2, ll_scale, s);
}
int tmdiff_haar_idwt2d_fwd(const float* ll, const float* lh, const float* hl, const float* hh, float* out,
                           int64_t planes, int32_t h, int32_t w, float in_scale, tmdiff_stream_t s) {
  const float* lls[2] = {ll, nullptr};
  float* outs[2] = {out, nullptr};
  return tmdiff_haar_idwt2d(lls, 1, lh, hl, hh, 0, 0, outs, planes, h, w, in_scale, s);
}
int tmdiff_haar_idwt2d_bwd(const float* g_out, float* g_ll, float* g_lh, float* g_hl, float* g_hh, int64_t planes,
                           int32_t h, int32_t w, float in_scale, tmdiff_stream_t s) {
  return tmdiff_haar_dwt2d(g_out, g_ll, g_lh, g_hl, g_hh, planes, 2 * h, 2 * w, in_scale, 1.f, s);
}

int tmdiff_dpm_axpby2(const float* x0, float c0, const float* x1, float c1, float* out, int64_t n, tmdiff_stream_t s) {
  const float* in[4] = {x0, x1, nullptr, nullptr};
  const float coef[4] = {c0, c1, 0.f, 0.f};
  return tmdiff_axpby(in, coef, 2, out, n, s);
}
int tmdiff_dpm_axpby3(const float* x0, float c0, const float* x1, float c1, const float* x2, float c2, float* out,
                      int64_t n, tmdiff_stream_t s) {
  const float* in[4] = {x0, x1, x2, nullptr};
  const float coef[4] = {c0, c1, c2, 0.f};
  return tmdiff_axpby(in, coef, 3, out, n, s);
}
int tmdiff_dpm_axpby4(const float* x0, float c0, const float* x1, float c1, const float* x2, float c2,
                      const float* x3, float c3, float* out, int64_t n, tmdiff_stream_t s) {
  const float* in[4] = {x0, x1, x2, x3};
  const float coef[4] = {c0, c1, c2, c3};
  return tmdiff_axpby(in, coef, 4, out, n, s);
}
}  // extern "C"
