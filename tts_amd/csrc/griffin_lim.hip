// Batched Griffin-Lim for gfx950: LDS-resident radix-4 Stockham FFTs (a 2048-point real
// transform computed as a 1024-point complex one), with the analysis window, the phase
// projection onto the target magnitudes, the inverse transform and the synthesis window
// fused into one kernel per iteration.  Overlap-add is done as a gather when the next
// iteration (or the final pass) reads the windowed frames, so no atomics are needed and
// the result is bit-reproducible.
//
// Follows AudioProcessor._griffin_lim / _stft / _istft / inv_mel_spectrogram /
// inv_spectrogram / apply_inv_preemphasis (Tacotron2/TTS/utils.py:896-942, 834-878,
// 802-804; Tacotron/TTS/utils.py:109-125, 169-208) with librosa 0.7 semantics for
// stft/istft: centre padding by n_fft/2 ('constant' or 'reflect'), periodic Hann window of
// win_length zero-padded centrally to n_fft, overlap-add divided by the window sum-square
// where it exceeds float32 tiny, n_fft/2 samples trimmed at both ends.
#include <math.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace {

constexpr int NFFT = 2048, MC = 1024, NBIN = 1025, FFT_THREADS = 256;
constexpr int WP = 1104;  // row pitch of the windowed-frame buffer (>= win_length, 16-byte multiple)

struct Tables {
  float2* tw;   // W_2048^m = exp(-2 pi i m / 2048), m = 0..2047
  float* win[2];
  int win_len[2];
};
Tables g_tab = {nullptr, {nullptr, nullptr}, {0, 0}};
std::mutex g_tab_mu;

__device__ inline float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ inline float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ inline float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// 1024-point complex FFT, 256 threads, radix-4 Stockham autosort, 5 passes.  Input in `a`
// (must be complete: caller syncs), output in `b`.  SIGN = -1 forward, +1 inverse (unscaled).
template <int SIGN>
__device__ inline void fft1024(float2* a, float2* b, const float2* __restrict__ tw, int tid) {
  float2* src = a;
  float2* dst = b;
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    const int Ns = 1 << (2 * p);
    const int k = tid & (Ns - 1);
    float2 v0 = src[tid], v1 = src[tid + 256], v2 = src[tid + 512], v3 = src[tid + 768];
    if (p > 0) {
      const int m = k * (512 >> (2 * p));
      float2 w1 = tw[m], w2 = tw[2 * m], w3 = tw[3 * m];
      if (SIGN > 0) w1 = cconj(w1), w2 = cconj(w2), w3 = cconj(w3);
      v1 = cmul(v1, w1), v2 = cmul(v2, w2), v3 = cmul(v3, w3);
    }
    const float2 s0 = cadd(v0, v2), d0 = csub(v0, v2), s1 = cadd(v1, v3), d1 = csub(v1, v3);
    // forward: -i*d1 = (d1.y, -d1.x); inverse: +i*d1 = (-d1.y, d1.x)
    const float2 jd1 = SIGN < 0 ? make_float2(d1.y, -d1.x) : make_float2(-d1.y, d1.x);
    const int j0 = ((tid >> (2 * p)) << (2 * p + 2)) + k;
    dst[j0] = cadd(s0, s1);
    dst[j0 + Ns] = cadd(d0, jd1);
    dst[j0 + 2 * Ns] = csub(s0, s1);
    dst[j0 + 3 * Ns] = csub(d0, jd1);
    __syncthreads();
    float2* t = src;
    src = dst;
    dst = t;
  }
}

// unit phasor of z with np.angle(0) = 0 semantics
__device__ inline float2 unit_phase(float2 z) {
  const float r = fmaxf(fabsf(z.x), fabsf(z.y));
  if (!(r > 0.f)) return make_float2(1.f, 0.f);
  const float a = z.x / r, b = z.y / r;
  const float n = 1.0f / sqrtf(a * a + b * b);
  return make_float2(a * n, b * n);
}

struct FrameGeom {
  int hop, win, lpad;  // lpad = (n_fft - win) / 2
  int pad_mode;
};

// Overlap-added, window-normalised signal at padded index g (0 <= g < n_fft + hop*(F-1)),
// gathered from the windowed frames yw[F][WP] of one utterance.  Frames are summed in
// increasing order in fp32, as the reference's frame-by-frame accumulation does.
__device__ inline float ola_sample(const float* __restrict__ yw, const float* __restrict__ s_win, int g, int F,
                                   const FrameGeom& q) {
  const int rel = g - q.lpad;
  if (rel < 0) return 0.f;
  int hi = rel / q.hop;
  if (hi > F - 1) hi = F - 1;
  int lo = rel - q.win + 1;
  lo = lo <= 0 ? 0 : (lo + q.hop - 1) / q.hop;
  float acc = 0.f, wss = 0.f;
  for (int i = lo; i <= hi; ++i) {
    const int n = rel - i * q.hop;
    acc += yw[(size_t)i * WP + n];
    const float w = s_win[n];
    wss += w * w;
  }
  if (wss > 1.17549435e-38f) acc /= wss;
  return acc;
}

// value of the centre-padded signal at padded index g, signal length L = hop*(F-1)
template <typename SampleFn>
__device__ inline float padded_sample(int g, int L, int pad_mode, SampleFn sample) {
  int m = g - NFFT / 2;
  if (m < 0 || m >= L) {
    if (pad_mode == TTSX_PAD_CONSTANT || L <= 1) return 0.f;
    const int period = 2 * (L - 1);
    m %= period;
    if (m < 0) m += period;
    if (m >= L) m = period - m;
  }
  return sample(m);
}

enum { K_GL_INIT = 0, K_GL_ITER = 1, K_STFT = 2, K_ISTFT = 3 };

struct FrameP {
  const float* mag;      // [B][F][NBIN]           (GL)
  const float* phase01;  // [B][F][NBIN]           (GL_INIT)
  const float* yw_in;    // [B][F][WP]             (GL_ITER)
  float* yw_out;         // [B][F][WP]             (GL_*, ISTFT)
  const float* y_in;     // [B][hop*(F-1)]         (STFT)
  float* spec_re;        // [B][F][NBIN]           (STFT out / ISTFT in)
  float* spec_im;
  const int32_t* n_frames;
  const float2* tw;
  const float* win;
  int B, F;
  FrameGeom q;
};

template <int KIND>
__global__ __launch_bounds__(FFT_THREADS) void frame_kernel(FrameP p) {
  __shared__ float2 s_a[MC];
  __shared__ float2 s_b[MC];
  __shared__ float2 s_tw[NFFT];
  __shared__ float s_win[WP];
  const int tid = threadIdx.x;
  for (int i = tid; i < NFFT; i += FFT_THREADS) s_tw[i] = p.tw[i];
  for (int i = tid; i < WP; i += FFT_THREADS) s_win[i] = i < p.q.win ? p.win[i] : 0.f;
  __syncthreads();
  const FrameGeom q = p.q;
  const long total = (long)p.B * p.F;

  for (long fi = blockIdx.x; fi < total; fi += gridDim.x) {
    const int b = (int)(fi / p.F), i = (int)(fi % p.F);
    const int Fb = p.n_frames ? p.n_frames[b] : p.F;
    if (i >= Fb) continue;  // uniform per block
    const int L = q.hop * (Fb - 1);
    const size_t frow = (size_t)b * p.F + i;

    if (KIND == K_GL_ITER || KIND == K_STFT) {
      // ---- analysis: windowed frame -> packed complex -> FFT -> half spectrum
      for (int m = tid; m < MC; m += FFT_THREADS) s_a[m] = make_float2(0.f, 0.f);
      __syncthreads();
      float* s_real = reinterpret_cast<float*>(s_a);  // x[2m] = re, x[2m+1] = im
      for (int n = tid; n < q.win; n += FFT_THREADS) {
        const int g = i * q.hop + q.lpad + n;
        float v;
        if (KIND == K_GL_ITER) {
          const float* yw = p.yw_in + (size_t)b * p.F * WP;
          v = padded_sample(g, L, q.pad_mode, [&](int m) { return ola_sample(yw, s_win, m + NFFT / 2, Fb, q); });
        } else {
          const float* y = p.y_in + (size_t)b * q.hop * (p.F - 1);
          v = padded_sample(g, L, q.pad_mode, [&](int m) { return y[m]; });
        }
        s_real[q.lpad + n] = v * s_win[n];
      }
      __syncthreads();
      fft1024<-1>(s_a, s_b, s_tw, tid);  // result in s_b
    }

    // ---- per-bin work on pairs (k, 1024-k); thread handles k = tid, tid+256 and (tid==0) 512
    for (int kk = 0; kk < 3; ++kk) {
      int k;
      if (kk < 2) k = tid + kk * 256;
      else if (tid == 0) k = 512;
      else break;
      const int mk = MC - k;  // partner bin (1024 for k = 0)
      float2 Xk, Xm;
      if (KIND == K_GL_ITER || KIND == K_STFT) {
        const float2 Zk = s_b[k], Zm = s_b[mk & (MC - 1)];
        const float2 E = make_float2(0.5f * (Zk.x + Zm.x), 0.5f * (Zk.y - Zm.y));
        const float2 D = make_float2(0.5f * (Zk.x - Zm.x), 0.5f * (Zk.y + Zm.y));  // (Zk - conj(Zm))/2
        const float2 O = make_float2(D.y, -D.x);                                  // -i * D
        const float2 WO = cmul(s_tw[k], O);
        Xk = cadd(E, WO);
        Xm = cconj(csub(E, WO));
      }
      if (KIND == K_STFT) {
        p.spec_re[frow * NBIN + k] = Xk.x, p.spec_im[frow * NBIN + k] = Xk.y;
        p.spec_re[frow * NBIN + mk] = Xm.x, p.spec_im[frow * NBIN + mk] = Xm.y;
        continue;
      }
      if (KIND == K_GL_ITER) {
        const float2 uk = unit_phase(Xk), um = unit_phase(Xm);
        const float sk = p.mag[frow * NBIN + k], sm = p.mag[frow * NBIN + mk];
        Xk = make_float2(sk * uk.x, sk * uk.y);
        Xm = make_float2(sm * um.x, sm * um.y);
      } else if (KIND == K_GL_INIT) {
        const float sk = p.mag[frow * NBIN + k], sm = p.mag[frow * NBIN + mk];
        float sn, cs;
        sincospif(2.0f * p.phase01[frow * NBIN + k], &sn, &cs);
        Xk = make_float2(sk * cs, sk * sn);
        sincospif(2.0f * p.phase01[frow * NBIN + mk], &sn, &cs);
        Xm = make_float2(sm * cs, sm * sn);
      } else {  // K_ISTFT
        Xk = make_float2(p.spec_re[frow * NBIN + k], p.spec_im[frow * NBIN + k]);
        Xm = make_float2(p.spec_re[frow * NBIN + mk], p.spec_im[frow * NBIN + mk]);
      }
      if (k == 0) Xk.y = 0.f, Xm.y = 0.f;  // the inverse real transform ignores Im of DC and Nyquist
      // ---- synthesis pre-pass: Z[k] = E + iO, Z[1024-k] = conj(E) + i conj(O)
      const float2 E = make_float2(0.5f * (Xk.x + Xm.x), 0.5f * (Xk.y - Xm.y));
      const float2 D = make_float2(0.5f * (Xk.x - Xm.x), 0.5f * (Xk.y + Xm.y));
      const float2 O = cmul(D, cconj(s_tw[k]));
      s_a[k] = make_float2(E.x - O.y, E.y + O.x);
      if (k != 0) s_a[mk] = make_float2(E.x + O.y, -E.y + O.x);
    }
    if (KIND == K_STFT) {
      __syncthreads();
      continue;
    }
    __syncthreads();
    fft1024<+1>(s_a, s_b, s_tw, tid);  // result in s_b, unscaled
    {
      const float* s_real = reinterpret_cast<const float*>(s_b);
      float* dst = p.yw_out + frow * WP;
      const float scale = 1.0f / MC;
      for (int n = tid; n < q.win; n += FFT_THREADS) dst[n] = s_real[q.lpad + n] * scale * s_win[n];
    }
    __syncthreads();
  }
}

// final overlap-add: y[b][m] for m < hop*(F_b - 1), zero beyond
__global__ void ola_kernel(float* __restrict__ y, const float* __restrict__ yw, const float* __restrict__ win,
                           const int32_t* __restrict__ n_frames, int B, int F, FrameGeom q) {
  __shared__ float s_win[WP];
  for (int i = threadIdx.x; i < WP; i += blockDim.x) s_win[i] = i < q.win ? win[i] : 0.f;
  __syncthreads();
  const int b = blockIdx.y;
  const int Fb = n_frames ? n_frames[b] : F;
  const int Lmax = q.hop * (F - 1), L = q.hop * (Fb - 1);
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Lmax) return;
  float v = 0.f;
  if (m < L) v = ola_sample(yw + (size_t)b * F * WP, s_win, m + NFFT / 2, Fb, q);
  y[(size_t)b * Lmax + m] = v;
}

// de-emphasis y[n] = x[n] + a*y[n-1] (lfilter([1],[1,-a])).  The impulse response a^k falls
// below 1e-17 after 2048 taps (a = 0.98), so each thread rebuilds the recurrence for its own
// 256-sample segment from a 2048-sample warm-up, in fp64 like the reference.
constexpr int DE_SEG = 256, DE_WARM = 2048;
__global__ void deemph_kernel(float* __restrict__ out, const float* __restrict__ x, const int32_t* __restrict__ n_frames,
                              int B, int F, int hop, float coef) {
  const int b = blockIdx.y;
  const int Fb = n_frames ? n_frames[b] : F;
  const int Lmax = hop * (F - 1), L = hop * (Fb - 1);
  const int seg = blockIdx.x * blockDim.x + threadIdx.x;
  const int start = seg * DE_SEG;
  if (start >= Lmax) return;
  const float* xb = x + (size_t)b * Lmax;
  float* ob = out + (size_t)b * Lmax;
  double acc = 0.0;
  const double a = (double)coef;
  for (int n = max(0, start - DE_WARM); n < start; ++n) acc = (double)xb[n] + a * acc;
  const int end = min(start + DE_SEG, Lmax);
  for (int n = start; n < end; ++n) {
    acc = (double)xb[n] + a * acc;
    ob[n] = n < L ? (float)acc : 0.f;
  }
}

// ---- spectrogram -> Griffin-Lim target magnitudes
struct MagP {
  const float* spec;       // [B][F][C]
  const float* inv_basis;  // [n_freq][C] or null
  float* mag;              // [B][F][n_freq]
  int C, n_freq;
  long n_rows;             // B*F
  ttsx_audio_params ap;
};

__device__ inline float denorm_to_amp(float s, const ttsx_audio_params& a) {
  // _denormalize then _db_to_amp(. + ref_level_db)
  if (a.signal_norm) {
    if (a.symmetric_norm) {
      if (a.clip_norm) s = fminf(fmaxf(s, -a.max_norm), a.max_norm);
      s = ((s + a.max_norm) * -a.min_level_db / (2.f * a.max_norm)) + a.min_level_db;
    } else {
      if (a.clip_norm) s = fminf(fmaxf(s, 0.f), a.max_norm);
      s = (s * -a.min_level_db / a.max_norm) + a.min_level_db;
    }
  }
  return powf(10.0f, (s + a.ref_level_db) * 0.05f);
}

constexpr int MAG_ROWS = 8;
__global__ __launch_bounds__(256) void spec_to_mag_kernel(MagP p) {
  __shared__ float s_amp[MAG_ROWS][128];
  const long r0 = (long)blockIdx.x * MAG_ROWS;
  const int tid = threadIdx.x;
  if (p.inv_basis) {
    for (int i = tid; i < MAG_ROWS * p.C; i += 256) {
      const int rr = i / p.C, c = i % p.C;
      s_amp[rr][c] = (r0 + rr < p.n_rows) ? denorm_to_amp(p.spec[(r0 + rr) * p.C + c], p.ap) : 0.f;
    }
    __syncthreads();
    for (int k = tid; k < p.n_freq; k += 256) {
      float acc[MAG_ROWS];
#pragma unroll
      for (int rr = 0; rr < MAG_ROWS; ++rr) acc[rr] = 0.f;
      const float* brow = p.inv_basis + (size_t)k * p.C;
      for (int c = 0; c < p.C; ++c) {
        const float w = brow[c];
#pragma unroll
        for (int rr = 0; rr < MAG_ROWS; ++rr) acc[rr] += w * s_amp[rr][c];
      }
#pragma unroll
      for (int rr = 0; rr < MAG_ROWS; ++rr)
        if (r0 + rr < p.n_rows) p.mag[(r0 + rr) * p.n_freq + k] = powf(fmaxf(1e-10f, acc[rr]), p.ap.power);
    }
  } else {
    for (int rr = 0; rr < MAG_ROWS; ++rr) {
      if (r0 + rr >= p.n_rows) break;
      for (int k = tid; k < p.n_freq; k += 256)
        p.mag[(r0 + rr) * p.n_freq + k] = powf(denorm_to_amp(p.spec[(r0 + rr) * p.C + k], p.ap), p.ap.power);
    }
  }
}

int get_tables(const ttsx_stft_params* sp, const float2** tw, const float** win, hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_tab_mu);
  if (!g_tab.tw) {
    std::vector<float2> h(NFFT);
    for (int m = 0; m < NFFT; ++m) {
      const double a = -2.0 * M_PI * (double)m / (double)NFFT;
      h[m] = make_float2((float)cos(a), (float)sin(a));
    }
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&g_tab.tw), NFFT * sizeof(float2)));
    TTSX_HIP(hipMemcpy(g_tab.tw, h.data(), NFFT * sizeof(float2), hipMemcpyHostToDevice));
  }
  int slot = -1;
  for (int s = 0; s < 2; ++s)
    if (g_tab.win_len[s] == sp->win_length) slot = s;
  if (slot < 0) {
    slot = g_tab.win_len[0] == 0 ? 0 : 1;
    if (g_tab.win[slot]) (void)hipFree(g_tab.win[slot]);
    // periodic Hann: scipy.signal.get_window('hann', win, fftbins=True)
    std::vector<float> h(sp->win_length);
    for (int n = 0; n < sp->win_length; ++n)
      h[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)sp->win_length));
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&g_tab.win[slot]), sp->win_length * sizeof(float)));
    TTSX_HIP(hipMemcpy(g_tab.win[slot], h.data(), sp->win_length * sizeof(float), hipMemcpyHostToDevice));
    g_tab.win_len[slot] = sp->win_length;
  }
  *tw = g_tab.tw;
  *win = g_tab.win[slot];
  (void)st;
  return TTSX_OK;
}

int check_stft_params(const ttsx_stft_params* sp, int B, int F) {
  TTSX_ARG(sp, "stft params: null");
  TTSX_ARG(sp->n_fft == NFFT, "n_fft=%d unsupported (only %d)", sp->n_fft, NFFT);
  TTSX_ARG(sp->win_length >= 2 && sp->win_length <= WP && sp->win_length <= NFFT,
           "win_length=%d outside 2..%d", sp->win_length, WP);
  TTSX_ARG(sp->hop_length >= 1 && sp->hop_length <= sp->win_length, "hop_length=%d", sp->hop_length);
  TTSX_ARG(sp->pad_mode == TTSX_PAD_CONSTANT || sp->pad_mode == TTSX_PAD_REFLECT, "pad_mode=%d", sp->pad_mode);
  TTSX_ARG(B >= 1 && F >= 2, "bad shape B=%d F=%d (need at least 2 frames)", B, F);
  return TTSX_OK;
}

// scratch for the windowed frames: two buffers of B*F*WP floats, grown on demand
struct Scratch {
  float* yw[2];
  float* y;
  size_t cap_yw, cap_y;
};
Scratch g_scr = {{nullptr, nullptr}, nullptr, 0, 0};
std::mutex g_scr_mu;

int ensure_scratch(size_t n_yw, size_t n_y) {
  if (n_yw > g_scr.cap_yw) {
    for (int i = 0; i < 2; ++i) {
      if (g_scr.yw[i]) (void)hipFree(g_scr.yw[i]);
      g_scr.yw[i] = nullptr;
      TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&g_scr.yw[i]), n_yw * sizeof(float)));
    }
    g_scr.cap_yw = n_yw;
  }
  if (n_y > g_scr.cap_y) {
    if (g_scr.y) (void)hipFree(g_scr.y);
    g_scr.y = nullptr;
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&g_scr.y), n_y * sizeof(float)));
    g_scr.cap_y = n_y;
  }
  return TTSX_OK;
}

unsigned frame_grid(long total) { return (unsigned)std::min<long>(total, 256 * 8); }

}  // namespace

extern "C" int ttsx_spec_to_mag(const float* spec, const float* inv_basis, int B, int F, int C, int n_freq,
                                const ttsx_audio_params* ap, float* mag, void* stream) {
  TTSX_ARG(spec && mag && ap, "ttsx_spec_to_mag: null argument");
  TTSX_ARG(B >= 1 && F >= 1 && n_freq >= 1, "ttsx_spec_to_mag: bad shape");
  TTSX_ARG(inv_basis ? (C >= 1 && C <= 128) : (C == n_freq), "ttsx_spec_to_mag: C=%d invalid", C);
  MagP p{spec, inv_basis, mag, C, n_freq, (long)B * F, *ap};
  spec_to_mag_kernel<<<(unsigned)(((long)B * F + MAG_ROWS - 1) / MAG_ROWS), 256, 0, static_cast<hipStream_t>(stream)>>>(p);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

extern "C" int ttsx_griffin_lim(const float* mag, const float* init_phase, const int32_t* n_frames, int B, int F,
                                const ttsx_stft_params* sp, int iters, float preemph, float* wav, void* stream) {
  TTSX_ARG(mag && init_phase && wav, "ttsx_griffin_lim: null argument");
  TTSX_ARG(iters >= 0, "ttsx_griffin_lim: iters=%d", iters);
  int rc = check_stft_params(sp, B, F);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float2* tw;
  const float* win;
  rc = get_tables(sp, &tw, &win, st);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(g_scr_mu);
  const size_t Lmax = (size_t)sp->hop_length * (F - 1);
  rc = ensure_scratch((size_t)B * F * WP, (size_t)B * Lmax);
  if (rc) return rc;
  FrameP p{};
  p.mag = mag, p.phase01 = init_phase, p.n_frames = n_frames, p.tw = tw, p.win = win, p.B = B, p.F = F;
  p.q = FrameGeom{sp->hop_length, sp->win_length, (NFFT - sp->win_length) / 2, sp->pad_mode};
  const unsigned grid = frame_grid((long)B * F);
  p.yw_out = g_scr.yw[0];
  frame_kernel<K_GL_INIT><<<grid, FFT_THREADS, 0, st>>>(p);
  TTSX_LAUNCH_CHECK();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    p.yw_in = g_scr.yw[cur], p.yw_out = g_scr.yw[cur ^ 1];
    frame_kernel<K_GL_ITER><<<grid, FFT_THREADS, 0, st>>>(p);
    TTSX_LAUNCH_CHECK();
    cur ^= 1;
  }
  const bool de = preemph != 0.f;
  ola_kernel<<<dim3((unsigned)((Lmax + 255) / 256), B), 256, 0, st>>>(de ? g_scr.y : wav, g_scr.yw[cur], win, n_frames,
                                                                    B, F, p.q);
  TTSX_LAUNCH_CHECK();
  if (de) {
    const int nseg = (int)((Lmax + DE_SEG - 1) / DE_SEG);
    deemph_kernel<<<dim3(ceil_div(nseg, 64), B), 64, 0, st>>>(wav, g_scr.y, n_frames, B, F, sp->hop_length, preemph);
    TTSX_LAUNCH_CHECK();
  }
  // the scratch is shared by every call in the process: finish before releasing the lock
  TTSX_HIP(hipStreamSynchronize(st));
  return TTSX_OK;
}

extern "C" int ttsx_stft(const float* y, int B, int F, const ttsx_stft_params* sp, float* spec_re, float* spec_im,
                         void* stream) {
  TTSX_ARG(y && spec_re && spec_im, "ttsx_stft: null argument");
  int rc = check_stft_params(sp, B, F);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float2* tw;
  const float* win;
  rc = get_tables(sp, &tw, &win, st);
  if (rc) return rc;
  FrameP p{};
  p.y_in = y, p.spec_re = spec_re, p.spec_im = spec_im, p.tw = tw, p.win = win, p.B = B, p.F = F;
  p.q = FrameGeom{sp->hop_length, sp->win_length, (NFFT - sp->win_length) / 2, sp->pad_mode};
  frame_kernel<K_STFT><<<frame_grid((long)B * F), FFT_THREADS, 0, st>>>(p);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

extern "C" int ttsx_istft(const float* spec_re, const float* spec_im, int B, int F, const ttsx_stft_params* sp,
                          float* y, void* stream) {
  TTSX_ARG(y && spec_re && spec_im, "ttsx_istft: null argument");
  int rc = check_stft_params(sp, B, F);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float2* tw;
  const float* win;
  rc = get_tables(sp, &tw, &win, st);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(g_scr_mu);
  const size_t Lmax = (size_t)sp->hop_length * (F - 1);
  rc = ensure_scratch((size_t)B * F * WP, 0);
  if (rc) return rc;
  FrameP p{};
  p.spec_re = const_cast<float*>(spec_re), p.spec_im = const_cast<float*>(spec_im), p.tw = tw, p.win = win;
  p.B = B, p.F = F, p.yw_out = g_scr.yw[0];
  p.q = FrameGeom{sp->hop_length, sp->win_length, (NFFT - sp->win_length) / 2, sp->pad_mode};
  frame_kernel<K_ISTFT><<<frame_grid((long)B * F), FFT_THREADS, 0, st>>>(p);
  TTSX_LAUNCH_CHECK();
  ola_kernel<<<dim3((unsigned)((Lmax + 255) / 256), B), 256, 0, st>>>(y, g_scr.yw[0], win, nullptr, B, F, p.q);
  TTSX_LAUNCH_CHECK();
  TTSX_HIP(hipStreamSynchronize(st));
  return TTSX_OK;
}
