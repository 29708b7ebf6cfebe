// Tacotron2 ConvBNBlock stacks for gfx950: k=5 convolutions as implicit GEMMs on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32), eval-mode BatchNorm folded into the weights, tanh / ReLU fused into
// the epilogue, residual add fused into the last postnet layer.
//
// Follows ConvBNBlock (Tacotron2/TTS/models.py:380-400) as used by Postnet.forward with the
// residual of Tacotron2.inference (:402-417, 809-810) and by Encoder.inference (:449-451).
//
// The kernel and its layouts are in conv_mfma.h; activations are frame-major [B][F][C], which is
// what Tacotron2.inference returns and what the embedding lookup produces.
#include <algorithm>

#include "conv_mfma.h"

using convmfma::Layer;

constexpr int MAX_LAYERS = 8;
struct ConvStack {
  int n;
  Layer layer[MAX_LAYERS];
  bool residual_last;
  float* buf[2];
  size_t buf_cap;
};

namespace {

int stack_create(ConvStack* p, int n, const int* chans, const int* acts, bool residual_last,
                 const float* const* conv_w, const float* const* conv_b, const float* const* bn_w,
                 const float* const* bn_b, const float* const* bn_m, const float* const* bn_v, hipStream_t st) {
  memset(p, 0, sizeof(*p));
  p->n = n, p->residual_last = residual_last;
  for (int i = 0; i < n; ++i) {
    TTSX_ARG(conv_w[i] && conv_b[i] && bn_w[i] && bn_b[i] && bn_m[i] && bn_v[i], "conv stack: null weight (layer %d)", i);
    TTSX_ARG(chans[i + 1] % 4 == 0, "conv stack: Cout=%d unsupported", chans[i + 1]);
    int rc = convmfma::layer_create(&p->layer[i], conv_w[i], conv_b[i], bn_w[i], bn_b[i], bn_m[i], bn_v[i],
                                    chans[i + 1], chans[i], 5, 2, acts[i], 1e-5f, st);
    if (rc) return rc;
  }
  TTSX_HIP(hipStreamSynchronize(st));
  return TTSX_OK;
}

void stack_destroy(ConvStack* p) {
  for (int i = 0; i < p->n; ++i) convmfma::layer_destroy(&p->layer[i]);
  for (int i = 0; i < 2; ++i)
    if (p->buf[i]) (void)hipFree(p->buf[i]);
}

int stack_run(ConvStack* p, const float* in, const int32_t* n_frames, int B, int F, float* out, hipStream_t st) {
  int cmax = 0;
  for (int i = 0; i < p->n; ++i) cmax = std::max(cmax, p->layer[i].cout);
  const size_t need = (size_t)B * F * cmax;
  if (need > p->buf_cap) {
    for (int i = 0; i < 2; ++i) {
      if (p->buf[i]) (void)hipFree(p->buf[i]);
      p->buf[i] = nullptr;
      p->buf_cap = 0;
      TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&p->buf[i]), need * sizeof(float)));
    }
    p->buf_cap = need;
  }
  const float* x = in;
  for (int i = 0; i < p->n; ++i) {
    const bool last = i == p->n - 1;
    const Layer& L = p->layer[i];
    float* y = last ? out : p->buf[i & 1];
    const float* res = (last && p->residual_last) ? in : nullptr;
    int rc = convmfma::layer_run(L, x, L.cin, y, L.cout, 0, res, L.cout, n_frames, B, F, st);
    if (rc) return rc;
    x = y;
  }
  return TTSX_OK;
}

}  // namespace

struct ttsx_postnet {
  ConvStack st;
};
struct ttsx_encoder_convs {
  ConvStack st;
};

extern "C" int ttsx_postnet_create(const ttsx_postnet_weights* w, void* stream, ttsx_postnet** out) {
  TTSX_ARG(w && out, "ttsx_postnet_create: null argument");
  ttsx_postnet* p = new ttsx_postnet();
  const int chans[6] = {80, 512, 512, 512, 512, 80};
  const int acts[5] = {1, 1, 1, 1, 0};
  int rc = stack_create(&p->st, 5, chans, acts, true, w->conv_w, w->conv_b, w->bn_weight, w->bn_bias, w->bn_mean,
                        w->bn_var, static_cast<hipStream_t>(stream));
  if (rc) {
    stack_destroy(&p->st);
    delete p;
    return rc;
  }
  *out = p;
  return TTSX_OK;
}

extern "C" int ttsx_postnet_destroy(ttsx_postnet* p) {
  if (!p) return TTSX_OK;
  stack_destroy(&p->st);
  delete p;
  return TTSX_OK;
}

extern "C" int ttsx_postnet_run(ttsx_postnet* p, const float* mel, const int32_t* n_frames, int B, int F, float* out,
                                void* stream) {
  TTSX_ARG(p && mel && out, "ttsx_postnet_run: null argument");
  TTSX_ARG(B >= 1 && F >= 1, "ttsx_postnet_run: bad shape B=%d F=%d", B, F);
  return stack_run(&p->st, mel, n_frames, B, F, out, static_cast<hipStream_t>(stream));
}

extern "C" int ttsx_encoder_convs_create(const ttsx_encoder_conv_weights* w, void* stream, ttsx_encoder_convs** out) {
  TTSX_ARG(w && out, "ttsx_encoder_convs_create: null argument");
  ttsx_encoder_convs* p = new ttsx_encoder_convs();
  const int chans[4] = {512, 512, 512, 512};
  const int acts[3] = {2, 2, 2};
  int rc = stack_create(&p->st, 3, chans, acts, false, w->conv_w, w->conv_b, w->bn_weight, w->bn_bias, w->bn_mean,
                        w->bn_var, static_cast<hipStream_t>(stream));
  if (rc) {
    stack_destroy(&p->st);
    delete p;
    return rc;
  }
  *out = p;
  return TTSX_OK;
}

extern "C" int ttsx_encoder_convs_destroy(ttsx_encoder_convs* p) {
  if (!p) return TTSX_OK;
  stack_destroy(&p->st);
  delete p;
  return TTSX_OK;
}

extern "C" int ttsx_encoder_convs_run(ttsx_encoder_convs* p, const float* x, const int32_t* lengths, int B, int T,
                                      float* out, void* stream) {
  TTSX_ARG(p && x && out, "ttsx_encoder_convs_run: null argument");
  TTSX_ARG(B >= 1 && T >= 1, "ttsx_encoder_convs_run: bad shape B=%d T=%d", B, T);
  return stack_run(&p->st, x, lengths, B, T, out, static_cast<hipStream_t>(stream));
}
