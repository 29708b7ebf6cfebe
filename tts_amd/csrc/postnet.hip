// Tacotron2 ConvBNBlock stacks for gfx950: k=5 convolutions as implicit GEMMs on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32), eval-mode BatchNorm folded into the weights, tanh / ReLU fused into
// the epilogue, residual add fused into the last postnet layer.
//
// Follows ConvBNBlock (Tacotron2/TTS/models.py:380-400) as used by Postnet.forward with the
// residual of Tacotron2.inference (:402-417, 809-810) and by Encoder.inference (:449-451).
//
// Layout.  Activations are frame-major [B][F][C] (what Tacotron2.inference returns), so the
// K axis of the implicit GEMM, ordered tap-major (k = tap*Cin + ci), is contiguous in memory
// for a fixed tap: the B operand of a 32x32x2 MFMA over four consecutive issues is one
// 16-byte load per lane, no im2col and no LDS staging.  Weights are packed per 32-row block
// and per 8-wide K chunk as 64 lanes x 4 floats in A-operand order (lane l: row l&31,
// k = 4*(l>>5) + s for issue s), so a wave streams them with coalesced 1 KiB loads.
#include <algorithm>

#include "common.h"

namespace {

constexpr int TAPS = 5, HALO = 2;

// wp[((cb*NC + c)*64 + lane)*4 + s] = w[co][ci][tap] * scale[co], co = cb*32 + (lane&31),
// k = c*8 + 4*(lane>>5) + s = tap*Cin + ci
__global__ void pack_conv_kernel(float* __restrict__ wp, float* __restrict__ bias_p, const float* __restrict__ w,
                                 const float* __restrict__ b, const float* __restrict__ g,
                                 const float* __restrict__ beta, const float* __restrict__ mean,
                                 const float* __restrict__ var, int Cout, int Cin) {
  const int NC = TAPS * Cin / 8;
  const size_t total = (size_t)ceil_div(Cout, 32) * NC * 256;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)round_up(Cout, 32)) {
    float v = 0.f;
    if (i < (size_t)Cout) {
      const float sc = g[i] / sqrtf(var[i] + 1e-5f);
      v = (b[i] - mean[i]) * sc + beta[i];
    }
    bias_p[i] = v;
  }
  if (i >= total) return;
  const int s = i & 3, lane = (i >> 2) & 63;
  const size_t grp = i >> 8;
  const int c = grp % NC, cb = grp / NC;
  const int co = cb * 32 + (lane & 31);
  const int k = c * 8 + 4 * (lane >> 5) + s;
  const int tap = k / Cin, ci = k % Cin;
  float v = 0.f;
  if (co < Cout) v = w[((size_t)co * Cin + ci) * TAPS + tap] * (g[co] / sqrtf(var[co] + 1e-5f));
  wp[i] = v;
}

struct ConvP {
  const float* wp;
  const float* bias;
  const float* x;        // [B][F][Cin]
  float* y;              // [B][F][Cout]
  const float* residual; // [B][F][Cout] or null
  const int32_t* n_frames;
  int B, F, Cin, Cout, NC, act;  // act: 0 none, 1 tanh, 2 relu
};

// Block = 4 waves as 2 (Cout) x 2 (frames); each wave owns a 64 x 64 output tile held as
// 2 x 2 accumulators of 32 x 32.  Block tile: 128 channels x 128 frames.
__global__ __launch_bounds__(256) void conv5_mfma_kernel(ConvP p) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int cb0 = blockIdx.y * 4 + wm * 2;             // first 32-row block of this wave
  const int ncb = ceil_div(p.Cout, 32);
  const long n0 = (long)blockIdx.x * 128 + wn * 64;    // first frame column of this wave

  // per-lane frame coordinates of the two column tiles
  int bb[2], ff[2], nf[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long n = n0 + j * 32 + l31;
    bb[j] = (int)(n / p.F);
    ff[j] = (int)(n % p.F);
    nf[j] = bb[j] < p.B ? (p.n_frames ? p.n_frames[bb[j]] : p.F) : 0;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const bool has_cb[2] = {cb0 < ncb, cb0 + 1 < ncb};
  const float4* W0 = reinterpret_cast<const float4*>(p.wp) + (size_t)(has_cb[0] ? cb0 : 0) * p.NC * 64 + lane;
  const float4* W1 = reinterpret_cast<const float4*>(p.wp) + (size_t)(has_cb[1] ? cb0 + 1 : 0) * p.NC * 64 + lane;
  const int chunks_per_tap = p.Cin / 8;

  for (int tap = 0; tap < TAPS; ++tap) {
    // source frame of each column tile for this tap
    const float* xrow[2];
    bool ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int fs = ff[j] + tap - HALO;
      ok[j] = fs >= 0 && fs < nf[j];
      xrow[j] = p.x + ((size_t)bb[j] * p.F + (ok[j] ? fs : 0)) * p.Cin + 4 * half;
    }
    const int cbase = tap * chunks_per_tap;
#pragma unroll 2
    for (int cc = 0; cc < chunks_per_tap; ++cc) {
      const float4 a0 = W0[(size_t)(cbase + cc) * 64];
      const float4 a1 = W1[(size_t)(cbase + cc) * 64];
      float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
      if (ok[0]) x0 = *reinterpret_cast<const float4*>(xrow[0] + cc * 8);
      if (ok[1]) x1 = *reinterpret_cast<const float4*>(xrow[1] + cc * 8);
#define TTSX_MFMA4(A, X, ACC)                                                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, X.x, ACC, 0, 0, 0);           \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, X.y, ACC, 0, 0, 0);           \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, X.z, ACC, 0, 0, 0);           \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, X.w, ACC, 0, 0, 0);
      TTSX_MFMA4(a0, x0, acc[0][0]);
      TTSX_MFMA4(a0, x1, acc[0][1]);
      TTSX_MFMA4(a1, x0, acc[1][0]);
      TTSX_MFMA4(a1, x1, acc[1][1]);
#undef TTSX_MFMA4
    }
  }

  // epilogue: lane holds column l31 (a frame) and rows 8g + 4*half + {0..3} of each 32-row block
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!has_cb[i]) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (bb[j] >= p.B) continue;
      const size_t obase = ((size_t)bb[j] * p.F + ff[j]) * p.Cout;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = (cb0 + i) * 32 + 8 * g + 4 * half;
        if (co >= p.Cout) continue;
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + co);
        float4 v = make_float4(acc[i][j][4 * g] + bv.x, acc[i][j][4 * g + 1] + bv.y, acc[i][j][4 * g + 2] + bv.z,
                               acc[i][j][4 * g + 3] + bv.w);
        if (p.act == 1) v = make_float4(tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w));
        else if (p.act == 2) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if (p.residual) {
          const float4 rv = *reinterpret_cast<const float4*>(p.residual + obase + co);
          v.x += rv.x, v.y += rv.y, v.z += rv.z, v.w += rv.w;
        }
        *reinterpret_cast<float4*>(p.y + obase + co) = v;
      }
    }
  }
}

}  // namespace

constexpr int MAX_LAYERS = 8;
struct ConvStack {
  int n;
  float* wp[MAX_LAYERS];
  float* bias[MAX_LAYERS];
  int cin[MAX_LAYERS], cout[MAX_LAYERS], act[MAX_LAYERS];
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
    const int cin = chans[i], cout = chans[i + 1];
    TTSX_ARG(cin % 8 == 0 && cout % 4 == 0, "conv stack: channel counts %d -> %d unsupported", cin, cout);
    TTSX_ARG(conv_w[i] && conv_b[i] && bn_w[i] && bn_b[i] && bn_m[i] && bn_v[i], "conv stack: null weight (layer %d)", i);
    p->cin[i] = cin, p->cout[i] = cout, p->act[i] = acts[i];
    const int NC = TAPS * cin / 8;
    const size_t nw = (size_t)ceil_div(cout, 32) * NC * 256;
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&p->wp[i]), nw * sizeof(float)));
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&p->bias[i]), round_up(cout, 32) * sizeof(float)));
    pack_conv_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>(p->wp[i], p->bias[i], conv_w[i], conv_b[i], bn_w[i],
                                                                  bn_b[i], bn_m[i], bn_v[i], cout, cin);
    TTSX_LAUNCH_CHECK();
  }
  TTSX_HIP(hipStreamSynchronize(st));
  return TTSX_OK;
}

void stack_destroy(ConvStack* p) {
  for (int i = 0; i < p->n; ++i) {
    if (p->wp[i]) (void)hipFree(p->wp[i]);
    if (p->bias[i]) (void)hipFree(p->bias[i]);
  }
  for (int i = 0; i < 2; ++i)
    if (p->buf[i]) (void)hipFree(p->buf[i]);
}

int stack_run(ConvStack* p, const float* in, const int32_t* n_frames, int B, int F, float* out, hipStream_t st) {
  int cmax = 0;
  for (int i = 0; i < p->n; ++i) cmax = std::max(cmax, p->cout[i]);
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
    ConvP c{};
    c.wp = p->wp[i], c.bias = p->bias[i], c.x = x, c.n_frames = n_frames;
    c.y = last ? out : p->buf[i & 1];
    c.residual = (last && p->residual_last) ? in : nullptr;
    c.B = B, c.F = F, c.Cin = p->cin[i], c.Cout = p->cout[i], c.NC = TAPS * p->cin[i] / 8, c.act = p->act[i];
    const long ncols = (long)B * F;
    dim3 grid((unsigned)((ncols + 127) / 128), ceil_div(c.Cout, 128));
    conv5_mfma_kernel<<<grid, 256, 0, st>>>(c);
    TTSX_LAUNCH_CHECK();
    x = c.y;
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
