// 1-D convolution (any tap count, including 1 = a linear layer over frames) as an implicit
// GEMM on the fp32 MFMA (v_mfma_f32_32x32x2_f32), shared by the Tacotron2 ConvBNBlock stacks
// and the Tacotron CBHG.
//
// Layout.  Activations are frame-major [B][F][C] with a channel pitch (so a layer can write a
// channel slice of a wider concat buffer).  The K axis of the implicit GEMM is tap-major
// (k = tap*Cin + ci), i.e. contiguous in memory for a fixed tap: the B operand of a 32x32x2
// MFMA over four consecutive issues is one 16-byte load per lane, no im2col, no LDS staging.
// Weights are packed per 32-row block and per 8-wide K chunk as 64 lanes x 4 floats in
// A-operand order (lane l: row l&31, k = 4*(l>>5) + s for issue s), so a wave streams them with
// coalesced 1 KiB loads.  Eval-mode BatchNorm and the conv bias are folded into the packed
// weights and a per-channel bias.
#pragma once
#include "common.h"

namespace convmfma {

// wp[((cb*NC + c)*64 + lane)*4 + s] = w[co][ci][tap] * scale[co], co = cb*32 + (lane&31),
// k = c*8 + 4*(lane>>5) + s = tap*Cin + ci;  scale = g/sqrt(var+eps) (1 without BatchNorm),
// bias_p[co] = (b[co] - mean[co]) * scale + beta[co]   (b, BatchNorm terms optional).
static __global__ void pack_conv_kernel(float* __restrict__ wp, float* __restrict__ bias_p,
                                        const float* __restrict__ w, const float* __restrict__ b,
                                        const float* __restrict__ g, const float* __restrict__ beta,
                                        const float* __restrict__ mean, const float* __restrict__ var, int Cout,
                                        int Cin, int taps, float eps) {
  const int NC = taps * Cin / 8;
  const size_t total = (size_t)ceil_div(Cout, 32) * NC * 256;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)round_up(Cout, 32)) {
    float v = 0.f;
    if (i < (size_t)Cout) {
      const float sc = g ? g[i] / sqrtf(var[i] + eps) : 1.f;
      v = ((b ? b[i] : 0.f) - (g ? mean[i] : 0.f)) * sc + (g ? beta[i] : 0.f);
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
  if (co < Cout) v = w[((size_t)co * Cin + ci) * taps + tap] * (g ? g[co] / sqrtf(var[co] + eps) : 1.f);
  wp[i] = v;
}

enum { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2, ACT_SIGMOID = 3 };

struct ConvP {
  const float* wp;
  const float* bias;
  const float* x;         // [B][F][x_pitch], channels 0..Cin-1 used
  float* y;               // [B][F][y_pitch], channels y_off .. y_off+Cout-1 written
  const float* residual;  // [B][F][r_pitch] or null, added after the activation
  const int32_t* n_frames;
  int B, F, Cin, Cout, NC, act, taps, pad_left;
  int x_pitch, y_pitch, y_off, r_pitch;
};

// Bias, activation, residual and store of one 32 x 32 tile (32-row block cb, the lane's frame).
static __device__ __forceinline__ void store_tile(const ConvP& p, int cb, const f32x16& acc, int bb, int ff, int half) {
  if (bb >= p.B) return;
  const size_t frow = (size_t)bb * p.F + ff;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int co = cb * 32 + 8 * g + 4 * half;
    if (co >= p.Cout) continue;
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + co);
    float v[4] = {acc[4 * g] + bv.x, acc[4 * g + 1] + bv.y, acc[4 * g + 2] + bv.z, acc[4 * g + 3] + bv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (p.act == ACT_TANH) v[e] = tanhf(v[e]);
      else if (p.act == ACT_RELU) v[e] = fmaxf(v[e], 0.f);
      else if (p.act == ACT_SIGMOID) v[e] = sigmoidf_(v[e]);
    }
    if (p.residual) {
      const float* r = p.residual + frow * p.r_pitch + co;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    float* dst = p.y + frow * p.y_pitch + p.y_off + co;
    // a channel count that is not a multiple of 4 (1025 linear bins) ends in a partial quad
    if (co + 3 < p.Cout && ((p.y_pitch | p.y_off) & 3) == 0) {
      *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (co + e < p.Cout) dst[e] = v[e];
    }
  }
}

// Block = 4 waves as 2 (Cout) x 2 (frames); each wave owns a 64 x 32 output tile held as two accumulators of
// 32 x 32.  Block tile: 128 channels x 64 frames.  (128-frame block tiles - 64 x 64 per wave, 216 registers, two
// waves per SIMD - measured slower everywhere: on small problems they leave a SIMD with a single wave, which cannot
// cover the latency of its own operand loads (encoder convs at 64 x 128: 286 against 200 us per layer), and on
// large ones the last of 3.1 rounds of tiles runs on an eighth of the chip (postnet at 64 x 800: 4.8 against
// 4.5 ms); here four waves fit a SIMD and the tail is a quarter as long.)
constexpr int NJ = 1;
static __global__ __launch_bounds__(256) void conv_mfma_kernel(ConvP p) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int cb0 = blockIdx.y * 4 + wm * 2;           // first 32-row block of this wave
  const int ncb = ceil_div(p.Cout, 32);
  const long n0 = (long)blockIdx.x * (64 * NJ) + wn * (32 * NJ);  // first frame column of this wave

  // per-lane frame coordinates of the two column tiles
  int bb[NJ], ff[NJ], nf[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const long n = n0 + j * 32 + l31;
    bb[j] = (int)(n / p.F);
    ff[j] = (int)(n % p.F);
    nf[j] = bb[j] < p.B ? (p.n_frames ? min(max(p.n_frames[bb[j]], 0), p.F) : p.F) : 0;   // (lengths clamped to 0..F)
  }

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const bool has_cb[2] = {cb0 < ncb, cb0 + 1 < ncb};
  const float4* W0 = reinterpret_cast<const float4*>(p.wp) + (size_t)(has_cb[0] ? cb0 : 0) * p.NC * 64 + lane;
  const float4* W1 = reinterpret_cast<const float4*>(p.wp) + (size_t)(has_cb[1] ? cb0 + 1 : 0) * p.NC * 64 + lane;
  const int chunks_per_tap = p.Cin / 8;
  // (columns past the last utterance read utterance B-1 and are masked: nf = 0)
  const float* xbase[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) xbase[j] = p.x + (size_t)min(bb[j], p.B - 1) * p.F * p.x_pitch + 4 * half;

  // operands of K chunk c (8 wide): two weight fragments, two activation fragments (zero outside the utterance)
  struct Frag {
    float4 a0, a1, x[NJ];
  };
  // (Loads are unconditional, from clamped addresses, and masked afterwards: a load under a lane
  // condition compiles to a branch whose join waits for every outstanding load, which serialises
  // the software pipeline.)
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load = [&](int c, int tap, int cc) -> Frag {
    Frag f;
    const bool in_k = c < p.NC;
    const int cl = in_k ? c : p.NC - 1;
    if (!in_k) tap = p.taps - 1, cc = chunks_per_tap - 1;
    f.a0 = W0[(size_t)cl * 64];
    f.a1 = W1[(size_t)cl * 64];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int fs = ff[j] + tap - p.pad_left;
      const int fc = min(max(fs, 0), p.F - 1);
      const float4 x = *reinterpret_cast<const float4*>(xbase[j] + (size_t)fc * p.x_pitch + cc * 8);
      f.x[j] = (in_k && fs >= 0 && fs < nf[j]) ? x : zero4;
    }
    return f;
  };
#define TTSX_MFMA4(A, X, ACC)                                         \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, X.x, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, X.y, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, X.z, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, X.w, ACC, 0, 0, 0);
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      TTSX_MFMA4(f.a0, f.x[j], acc[0][j]);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      TTSX_MFMA4(f.a1, f.x[j], acc[1][j]);
    }
  };
  // software pipeline over the K chunks (tap-major): the operands of chunk c+1 are requested before
  // the 16 MFMAs of chunk c issue
  int tap = 0, cc = 0;
  auto advance = [&]() {
    if (++cc == chunks_per_tap) cc = 0, ++tap;
  };
  Frag fa = load(0, tap, cc);
  advance();
  for (int c = 0; c < p.NC; c += 2) {
    Frag fb = load(c + 1, tap, cc);
    advance();
    mma(fa);
    fa = load(c + 2, tap, cc);
    advance();
    mma(fb);
  }
#undef TTSX_MFMA4

  // epilogue: lane holds column l31 (a frame) and rows 8g + 4*half + {0..3} of each 32-row block
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!has_cb[i]) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) store_tile(p, cb0 + i, acc[i][j], bb[j], ff[j], half);
  }
}

// Small problems (few frames): the big tiling would leave the chip empty and walk K serially in
// one wave.  Here a block owns ONE 32 x 32 output tile and its four waves split K (wave w takes
// chunks w, w+4, ...; four chunks of loads in flight per wave), partial tiles summed through LDS.
static __global__ __launch_bounds__(256) void conv_mfma_small_kernel(ConvP p) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, l31 = lane & 31;
  const int cb = blockIdx.y;
  const long n = (long)blockIdx.x * 32 + l31;
  const int bb = (int)(n / p.F), ff = (int)(n % p.F);
  const int nf = bb < p.B ? (p.n_frames ? min(max(p.n_frames[bb], 0), p.F) : p.F) : 0;
  __shared__ f32x16 red[4][64];
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float4* W = reinterpret_cast<const float4*>(p.wp) + (size_t)cb * p.NC * 64 + lane;
  const int chunks_per_tap = p.Cin / 8;
  constexpr int U = 4;
  for (int c0 = wave * U; c0 < p.NC; c0 += 4 * U) {
    float4 a[U], x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u;
      a[u] = make_float4(0.f, 0.f, 0.f, 0.f), x[u] = a[u];
      if (c < p.NC) {
        a[u] = W[(size_t)c * 64];
        const int tap = c / chunks_per_tap, cc = c - tap * chunks_per_tap;
        const int fs = ff + tap - p.pad_left;
        if (fs >= 0 && fs < nf)
          x[u] = *reinterpret_cast<const float4*>(p.x + ((size_t)bb * p.F + fs) * p.x_pitch + 4 * half + cc * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, x[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, x[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, x[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, x[u].w, acc, 0, 0, 0);
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    f32x16 v = red[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += red[w][lane];
    store_tile(p, cb, v, bb, ff, half);
  }
}

// A packed layer and its launch.
struct Layer {
  float* wp;
  float* bias;
  int cin, cout, taps, pad_left, act;
};

static inline int layer_create(Layer* L, const float* w, const float* b, const float* g, const float* beta,
                               const float* mean, const float* var, int cout, int cin, int taps, int pad_left,
                               int act, float eps, hipStream_t st) {
  TTSX_ARG(cin % 8 == 0, "conv layer: Cin=%d is not a multiple of 8", cin);
  TTSX_ARG(w, "conv layer: null weight");
  L->cin = cin, L->cout = cout, L->taps = taps, L->pad_left = pad_left, L->act = act;
  const int NC = taps * cin / 8;
  const size_t nw = (size_t)ceil_div(cout, 32) * NC * 256;
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&L->wp), nw * sizeof(float)));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&L->bias), round_up(cout, 32) * sizeof(float)));
  pack_conv_kernel<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>(L->wp, L->bias, w, b, g, beta, mean, var, cout, cin,
                                                                taps, eps);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

static inline void layer_destroy(Layer* L) {
  if (L->wp) (void)hipFree(L->wp);
  if (L->bias) (void)hipFree(L->bias);
  L->wp = L->bias = nullptr;
}

static inline int layer_run(const Layer& L, const float* x, int x_pitch, float* y, int y_pitch, int y_off,
                            const float* residual, int r_pitch, const int32_t* n_frames, int B, int F,
                            hipStream_t st) {
  ConvP c{};
  c.wp = L.wp, c.bias = L.bias, c.x = x, c.y = y, c.residual = residual, c.n_frames = n_frames;
  c.B = B, c.F = F, c.Cin = L.cin, c.Cout = L.cout, c.NC = L.taps * L.cin / 8, c.act = L.act, c.taps = L.taps;
  c.pad_left = L.pad_left, c.x_pitch = x_pitch, c.y_pitch = y_pitch, c.y_off = y_off, c.r_pitch = r_pitch;
  const long ncols = (long)B * F;
  dim3 grid((unsigned)((ncols + 127) / 128), ceil_div(L.cout, 128));
  if ((long)grid.x * grid.y < 128) {
    dim3 sgrid((unsigned)((ncols + 31) / 32), ceil_div(L.cout, 32));
    conv_mfma_small_kernel<<<sgrid, 256, 0, st>>>(c);
  } else {
    grid.x = (unsigned)((ncols + 63) / 64);
    conv_mfma_kernel<<<grid, 256, 0, st>>>(c);
  }
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

}  // namespace convmfma
