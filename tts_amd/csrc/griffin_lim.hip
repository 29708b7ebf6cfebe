// Batched Griffin-Lim for gfx950.
//
// Follows AudioProcessor._griffin_lim / _stft / _istft / inv_mel_spectrogram /
// inv_spectrogram / apply_inv_preemphasis (Tacotron2/TTS/utils.py:896-942, 834-878,
// 802-804; Tacotron/TTS/utils.py:109-125, 169-208) with librosa 0.7 semantics for
// stft/istft: centre padding by n_fft/2 ('constant' or 'reflect'), periodic Hann window of
// win_length zero-padded centrally to n_fft, overlap-add divided by the window sum-square
// where it exceeds float32 tiny, n_fft/2 samples trimmed at both ends.
//
// One Griffin-Lim iteration is ONE launch.  A workgroup owns a run of consecutive frames of one
// utterance and walks it in rounds of four frames, one WAVE per frame:
//   windowed frame -> 2048-point real FFT (as a 1024-point complex FFT, radix 16 x 16 x 4,
//   16 points per lane in registers, LDS only for the transposes between passes) -> unit phasor
//   x target magnitude -> inverse FFT -> synthesis window -> windowed frame left in LDS;
// after each round the workgroup overlap-adds the four new frames onto a carry of the previous
// ones (fixed order, no atomics: bit-reproducible), applies the window-sum-square normalisation
// and writes hop-sized slots of the next padded signal.  The windowed frames never travel to
// HBM.  A run starts with "halo" rounds that rebuild the carry of the frames before it.
// The stand-alone STFT / ISTFT entry points use the same frame code with a separate
// overlap-add launch.
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "common.h"
#include "pk_complex.h"

namespace {

// Tuning builds (-DTTSX_GL_TIMING): wave 0 of one workgroup in the middle of the grid leaves 100 MHz timestamps
// at the phase boundaries of one round (tools/gl_timing.py).
#ifdef TTSX_GL_TIMING
__device__ long long gl_marks[32];
#define GL_MARK(i)                                                                \
  do {                                                                            \
    if (mark) {                                                                   \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                 \
      if (threadIdx.x == 0) gl_marks[i] = wall_clock64();                         \
    }                                                                             \
  } while (0)
// (without waiting for the stores of the phase to be acknowledged)
#define GL_MARK_NV(i)                                                             \
  do {                                                                            \
    if (mark) {                                                                   \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
      if (threadIdx.x == 0) gl_marks[i] = wall_clock64();                         \
    }                                                                             \
  } while (0)
#else
#define GL_MARK(i) do { } while (0)
#define GL_MARK_NV(i) do { } while (0)
#endif

constexpr int NFFT = 2048, MC = 1024, NBIN = 1025;
constexpr int WP = 1104;          // row pitch of the windowed-frame buffer (floats)
constexpr int FR_PER_WG = 4;      // frames (waves) per workgroup
constexpr int LDS_PER_FR = MC + MC / 16;  // float2 per frame, padded by one every 16 against bank conflicts
constexpr int OLA_MAXK = 8;       // frames that can overlap one sample: ceil(win / hop) <= 8

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ int pidx(int i) { return i + (i >> 4); }
// *(T*)((char*)base + byte_off) with a zero-extended 32-bit offset
template <typename T>
__device__ __forceinline__ T ld_at(const void* base, unsigned byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// 4-point DFT in place; SIGN = -1 forward, +1 inverse
template <int SIGN>
__device__ __forceinline__ void dft4(float2& a, float2& b, float2& c, float2& d) {
  const float2 s0 = cadd(a, c), d0 = csub(a, c), s1 = cadd(b, d), d1 = csub(b, d);
  const float2 jd1 = SIGN < 0 ? make_float2(d1.y, -d1.x) : make_float2(-d1.y, d1.x);  // -+ i * d1
  a = cadd(s0, s1);
  b = cadd(d0, jd1);
  c = csub(s0, s1);
  d = csub(d0, jd1);
}

// 16-point DFT of v[r] (r = time index) in registers; result X[q] is left in v[q].
template <int SIGN>
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
  // r = r1 + 4 r2: DFT over r2 for each r1
#pragma unroll
  for (int r1 = 0; r1 < 4; ++r1) dft4<SIGN>(v[r1], v[r1 + 4], v[r1 + 8], v[r1 + 12]);
  // now v[r1 + 4 q2] = A[r1][q2]; multiply by W16^(r1 q2)
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, C2 = 0.70710678118654752f;
  const float sg = SIGN < 0 ? -1.f : 1.f;
  // W16^m = (cos(pi m/8), sg*sin(pi m/8))
  const float2 w1 = make_float2(C1, sg * S1), w2 = make_float2(C2, sg * C2), w3 = make_float2(S1, sg * C1);
  const float2 w4 = make_float2(0.f, sg), w6 = make_float2(-C2, sg * C2), w9 = make_float2(-C1, -sg * S1);
  v[1 + 4] = cmul(v[1 + 4], w1);
  v[2 + 4] = cmul(v[2 + 4], w2);
  v[3 + 4] = cmul(v[3 + 4], w3);
  v[1 + 8] = cmul(v[1 + 8], w2);
  v[2 + 8] = cmul(v[2 + 8], w4);
  v[3 + 8] = cmul(v[3 + 8], w6);
  v[1 + 12] = cmul(v[1 + 12], w3);
  v[2 + 12] = cmul(v[2 + 12], w6);
  v[3 + 12] = cmul(v[3 + 12], w9);
  // DFT over r1 for each q2: X[4 q1 + q2]
#pragma unroll
  for (int q2 = 0; q2 < 4; ++q2) dft4<SIGN>(v[4 * q2], v[4 * q2 + 1], v[4 * q2 + 2], v[4 * q2 + 3]);
  // v[4 q2 + q1] holds X[4 q1 + q2]: transpose the 4x4 index
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b) {
      const float2 t = v[4 * a + b];
      v[4 * a + b] = v[4 * b + a];
      v[4 * b + a] = t;
    }
}

// W_1024^m = exp(-2 pi i m / 1024) (m < TW_MAIN) from the LDS table; conjugated for the inverse.  Entries
// TW_MAIN + 16 r + k hold W_256^(k r) (r, k < 16), the twiddles of the second radix-16 pass, in the order the lanes
// read them: lanes with different k of one r hit consecutive words (read as W_1024^(4 k r) from the main table,
// strides of 4 r entries put up to sixteen different addresses on one bank).
constexpr int TW_MAIN = 768, TW_ALL = TW_MAIN + 256;
template <int SIGN>
__device__ __forceinline__ float2 tw1024(const float2* s_tw, int m) {
  const float2 w = s_tw[m];
  return SIGN < 0 ? w : cconj(w);
}

// Passes 2 and 3 of the 1024-point FFT of one wave.  On entry v[r] holds, for lane j, the
// output of the first radix-16 pass (butterfly j, Stockham order); on exit v[4 i + r] holds
// element (lane + 64 i) + 256 r of the result.  LDS addresses are written as a per-lane base
// plus a compile-time constant (pidx(lane + 64 r) = lane + (lane >> 4) + 68 r, etc.) so that
// they fold into instruction offsets; `z` is an opaque zero that keeps the twiddle loads
// inside the frame loop (hoisting them costs ~60 registers and an occupancy step).
template <int SIGN>
__device__ __forceinline__ void fft1024_tail(float2 (&v)[16], float2* s, const float2* s_tw, int lane, int z) {
  const int bl = lane + (lane >> 4);  // pidx(lane)
  // pass 1 store: dst[j*16 + r] -> pidx = 17 j + r
#pragma unroll
  for (int r = 0; r < 16; ++r) s[17 * lane + r] = v[r];
  // pass 2: Ns = 16, k = j % 16, twiddle W_256^(k r) = W_1024^(4 k r)
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = s[bl + 68 * r];
  const int k = lane & 15;
#pragma unroll
  for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], tw1024<SIGN>(s_tw, TW_MAIN + 16 * r + k + z));
  dft16<SIGN>(v);
  const int j0 = (lane >> 4) * 256 + k;
  const int bj = j0 + (j0 >> 4);  // pidx(j0 + 16 r) = bj + 17 r
#pragma unroll
  for (int r = 0; r < 16; ++r) s[bj + 17 * r] = v[r];
  // pass 3: radix 4, Ns = 256, four butterflies per lane: j = lane + 64 i; pidx(j + 256 r) = bl + 68 i + 272 r
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[4 * i + r] = s[bl + 68 * i + 272 * r];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = lane + 64 * i + z;
    v[4 * i + 1] = cmul(v[4 * i + 1], tw1024<SIGN>(s_tw, j));
    v[4 * i + 2] = cmul(v[4 * i + 2], tw1024<SIGN>(s_tw, 2 * j));
    v[4 * i + 3] = cmul(v[4 * i + 3], tw1024<SIGN>(s_tw, 3 * j));
    dft4<SIGN>(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  }
}

// unit phasor of z with np.angle(0) = 0 semantics
__device__ __forceinline__ float2 unit_phase(float2 z) {
  const float s = z.x * z.x + z.y * z.y;
  if (s > 1e-30f && s < 1e30f) {  // the usual case: |z|^2 neither underflows nor overflows
    const float n = __builtin_amdgcn_rsqf(s);
    return make_float2(z.x * n, z.y * n);
  }
  const float r = fmaxf(fabsf(z.x), fabsf(z.y));
  if (!(r > 0.f)) return make_float2(1.f, 0.f);
  const float ir = __builtin_amdgcn_rcpf(r);   // prescale by the larger component: no overflow / underflow
  const float a = z.x * ir, b = z.y * ir;
  const float n = __builtin_amdgcn_rsqf(a * a + b * b);
  return make_float2(a * n, b * n);
}

struct FrameGeom {
  int hop, win, lpad;  // lpad = (n_fft - win) / 2
  int pad_mode;
  int LP;              // padded signal length per utterance: n_fft + hop*(F-1), rounded up to 4
};

enum { K_GL_INIT = 0, K_GL_ITER = 1, K_STFT = 2, K_ISTFT = 3 };

struct FrameP {
  const float* mag;      // [B][F][NBIN]           (GL)
  const float* phase01;  // [B][F][NBIN]           (GL_INIT)
  const float* ypad;     // [B][LP]                (GL_ITER, STFT)
  float* yw_out;         // [B][F][WP]             (GL_*, ISTFT): samples lpad_even .. lpad_even+WP of each frame
  float* spec_re;        // [B][F][NBIN]           (STFT out / ISTFT in)
  float* spec_im;
  const int32_t* n_frames;
  const float2* tw;
  const float* winpad;   // [2048]
  int B, F;
  FrameGeom q;
};

__device__ __forceinline__ void load_tables(float2* s_tw, float2* s_tw2, const float2* tw, int tid) {
  for (int i = tid; i < TW_ALL; i += 64 * FR_PER_WG) {
    const int e = i < TW_MAIN ? i : (4 * ((i - TW_MAIN) & 15) * ((i - TW_MAIN) >> 4)) & 1023;
    const float2 w = tw[(2 * e) & 1023];  // W_1024^e = W_2048^(2e), and W_1024^(e+512) = -W_1024^e
    s_tw[i] = e >= 512 ? make_float2(-w.x, -w.y) : w;
  }
  for (int i = tid; i <= 512; i += 64 * FR_PER_WG) s_tw2[i] = tw[i];
}

// One frame (utterance b, frame i) by one wave.  `s` is the wave's LDS region.  TO_LDS: the
// windowed synthesis frame (samples lpad_even .. lpad_even + WP of the frame) is left in the
// wave's own region, viewed as floats, instead of being written to yw_out.
template <int KIND, bool TO_LDS>
__device__ __forceinline__ void frame_body(const FrameP& p, int b, int i, float2* s, const float2* s_tw,
                                           const float2* s_tw2, int lane) {
  {
    const FrameGeom q = p.q;
    // the padded window (8 KiB) stays L1-resident: read it through the vector cache, not LDS
    const float2* s_win2 = reinterpret_cast<const float2*>(p.winpad);
    const size_t frow = (size_t)b * p.F + i;
    float2 v[16];
    int z = 0;
    asm volatile("" : "+v"(z));  // opaque zero: pins loop-invariant table loads inside the loop
    // opaque per-phase copies of the lane id: addresses derived from them are recomputed inside
    // their phase instead of staying live across the whole loop body
    int lane_a = lane, lane_b = lane, lane_s = lane;
    asm volatile("" : "+v"(lane_a), "+v"(lane_b), "+v"(lane_s));
    // (ranges restated for the compiler: with them every index is provably a small non-negative number and the
    // global accesses take the scalar-base + 32-bit lane offset form instead of 64-bit per-lane address arithmetic)
    __builtin_assume((unsigned)lane_a < 64u && (unsigned)lane_b < 64u && (unsigned)lane_s < 64u);
    __builtin_assume((unsigned)z < 8u);

    // target magnitudes of this lane's bin pairs, requested now so that they land during the analysis FFT
    float mk[8], mm[8], m512 = 0.f;
    if (KIND == K_GL_ITER) {
      const float* mg0 = p.mag + frow * NBIN;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        mk[c] = mg0[lane_a + 64 * c];
        mm[c] = mg0[MC - (lane_a + 64 * c)];
      }
      m512 = mg0[MC / 2];
    }
    if (KIND == K_GL_ITER || KIND == K_STFT) {
      // ---- analysis: z[m] = (x[2m], x[2m+1]) windowed, first radix-16 pass straight from HBM
      const float* src = p.ypad + (size_t)b * q.LP + (size_t)i * q.hop;
      // second bases 4 KiB further on, opaque: the constant part of every address then fits the instruction offset
      unsigned hi_off = 1024;
      asm volatile("" : "+s"(hi_off));
      const float* src_hi = src + hi_off;
      const float2* win_hi = s_win2 + (hi_off >> 1);
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        // four groups of loads, fenced so that at most one group's operands are live at a time and every load of a
        // group has left before its first product waits
        float2 w[4];
        float x0[4], x1[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r = 4 * h + g;
          // (uniform base + constant, then a 32-bit unsigned lane offset: the scalar-base addressing form)
          w[g] = ld_at<float2>((r < 8 ? s_win2 : win_hi) + 64 * (r & 7), (unsigned)(lane_a + z) * 8u);
          // the hop is odd in general: two 4-byte loads
          const float* sr = (r < 8 ? src : src_hi) + 128 * (r & 7);
          x0[g] = ld_at<float>(sr, (unsigned)lane_a * 8u), x1[g] = ld_at<float>(sr + 1, (unsigned)lane_a * 8u);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) v[4 * h + g] = make_float2(x0[g] * w[g].x, x1[g] * w[g].y);
        __builtin_amdgcn_sched_barrier(0);
      }
      dft16<-1>(v);
      fft1024_tail<-1>(v, s, s_tw, lane_a, z);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[lane_a + (lane_a >> 4) + 68 * c + 272 * r] = v[4 * c + r];
    }

    // ---- per-bin work on pairs (k, 1024-k): k = lane + 64 c (c < 8), and bin 512 on lane 0
    const float* mg = p.mag + frow * NBIN;
    // operands of the eight pairs first (LDS reads cannot be moved across the LDS writes of an earlier pair by
    // the compiler, so read-compute-write per pair would expose one LDS latency per pair)
    float2 Zk8[8], Zm8[8], W8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int k = lane_b + 64 * c;
      if (KIND == K_GL_ITER || KIND == K_STFT) Zk8[c] = s[pidx(k)], Zm8[c] = s[pidx((MC - k) & (MC - 1))];
      W8[c] = s_tw2[k + z];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int k = lane_b + 64 * c;
      const int mkk = MC - k;  // partner bin (1024 for k = 0)
      const float2 W = W8[c];
      float2 Xk, Xm;
      if (KIND == K_GL_ITER || KIND == K_STFT) {
        const float2 Zk = Zk8[c], Zm = Zm8[c];
        const float2 E = make_float2(0.5f * (Zk.x + Zm.x), 0.5f * (Zk.y - Zm.y));
        const float2 D = make_float2(0.5f * (Zk.x - Zm.x), 0.5f * (Zk.y + Zm.y));  // (Zk - conj(Zm))/2
        const float2 O = make_float2(D.y, -D.x);                                  // -i * D
        const float2 WO = cmul(W, O);
        Xk = cadd(E, WO);
        Xm = cconj(csub(E, WO));
      }
      if (KIND == K_STFT) {
        p.spec_re[frow * NBIN + k] = Xk.x, p.spec_im[frow * NBIN + k] = Xk.y;
        p.spec_re[frow * NBIN + mkk] = Xm.x, p.spec_im[frow * NBIN + mkk] = Xm.y;
        continue;
      }
      float sk = 0.f, sm = 0.f;
      if (KIND == K_GL_ITER) sk = mk[c], sm = mm[c];
      if (KIND == K_GL_INIT) sk = mg[k], sm = mg[mkk];
      if (KIND == K_GL_ITER) {
        const float2 uk = unit_phase(Xk), um = unit_phase(Xm);
        Xk = make_float2(sk * uk.x, sk * uk.y);
        Xm = make_float2(sm * um.x, sm * um.y);
      } else if (KIND == K_GL_INIT) {
        float sn, cs;
        sincospif(2.0f * p.phase01[frow * NBIN + k], &sn, &cs);
        Xk = make_float2(sk * cs, sk * sn);
        sincospif(2.0f * p.phase01[frow * NBIN + mkk], &sn, &cs);
        Xm = make_float2(sm * cs, sm * sn);
      } else {  // K_ISTFT
        Xk = make_float2(p.spec_re[frow * NBIN + k], p.spec_im[frow * NBIN + k]);
        Xm = make_float2(p.spec_re[frow * NBIN + mkk], p.spec_im[frow * NBIN + mkk]);
      }
      if (k == 0) Xk.y = 0.f, Xm.y = 0.f;  // the inverse real transform ignores Im of DC and Nyquist
      // synthesis pre-pass: Z[k] = E + iO, Z[1024-k] = conj(E) + i conj(O)
      const float2 E = make_float2(0.5f * (Xk.x + Xm.x), 0.5f * (Xk.y - Xm.y));
      const float2 D = make_float2(0.5f * (Xk.x - Xm.x), 0.5f * (Xk.y + Xm.y));
      const float2 O = cmul(D, cconj(W));
      const float2 zk = make_float2(E.x - O.y, E.y + O.x), zm = make_float2(E.x + O.y, -E.y + O.x);
      s[pidx(k)] = zk;
      // (k = 0 has no partner slot: it stores its own value a second time, which keeps the store unconditional
      // and the eight pairs in one basic block)
      s[pidx(k ? mkk : 0)] = k ? zm : zk;
    }
    if (lane_b == 0) {
      // Bin 512 is its own partner and its twiddle W_2048^512 is -i: the post-pass of the real transform reduces
      // to X[512] = conj(Z[512]) and the pre-pass of the inverse to Z'[512] = conj(X'[512]).
      float2 X = make_float2(0.f, 0.f);
      if (KIND == K_GL_ITER || KIND == K_STFT) X = cconj(s[pidx(512)]);
      if (KIND == K_STFT) {
        p.spec_re[frow * NBIN + 512] = X.x, p.spec_im[frow * NBIN + 512] = X.y;
      } else {
        if (KIND == K_GL_ITER) {
          const float2 u = unit_phase(X);
          X = make_float2(m512 * u.x, m512 * u.y);
        } else if (KIND == K_GL_INIT) {
          float sn, cs;
          sincospif(2.0f * p.phase01[frow * NBIN + 512], &sn, &cs);
          const float sk = mg[512];
          X = make_float2(sk * cs, sk * sn);
        } else {  // K_ISTFT
          X = make_float2(p.spec_re[frow * NBIN + 512], p.spec_im[frow * NBIN + 512]);
        }
        s[pidx(512)] = cconj(X);
      }
    }
    if (KIND == K_STFT) return;

    // ---- synthesis: inverse FFT, window, store the windowed frame
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = s[lane_s + (lane_s >> 4) + 68 * r];
    dft16<+1>(v);
    fft1024_tail<+1>(v, s, s_tw, lane_s, z);
    {
      const int lpe = q.lpad & ~1;
      // (all lanes have read their last pass from `s` before the first of these stores issues)
      float* dst = TO_LDS ? reinterpret_cast<float*>(s) : p.yw_out + frow * WP;
      const float scale = 1.0f / MC;
      // the window values of all sixteen sample pairs first (always inside the table), as one burst: a load
      // inside the range check below would cost one cache round trip per pair
      float2 wn[16];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) wn[4 * c + r] = s_win2[lane_s + 64 * c + 256 * r + z];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = lane_s + 64 * c + 256 * r;  // complex index -> samples 2m, 2m+1
          const int n = 2 * m - lpe;
          if (n >= 0 && n < WP) {
            const float2 w = wn[4 * c + r];
            *reinterpret_cast<float2*>(dst + n) = make_float2(v[4 * c + r].x * scale * w.x, v[4 * c + r].y * scale * w.y);
          }
        }
    }
  }
}

// ---- The Griffin-Lim iteration's frame in packed arithmetic (pk_complex.h): the same transforms, LDS layout and
// pass structure as frame_body, with every complex value in one register pair.  INV = inverse transform.
template <bool INV>
__device__ __forceinline__ void pdft4(f2& a, f2& b, f2& c, f2& d) {
  const f2 s0 = a + c, d0 = a - c, s1 = b + d, d1 = b - d;
  a = s0 + s1;
  c = s0 - s1;
  if (!INV) b = pk_addmj(d0, d1), d = pk_addpj(d0, d1);  // forward: b = d0 - i d1
  else b = pk_addpj(d0, d1), d = pk_addmj(d0, d1);
}

// ZIN: inputs 0, 1, 2, 13, 14 and 15 are zero (forward transform of a frame whose window covers the middle of the
// transform length only) and their registers are not read: the first stage takes 20 additions instead of 32.
template <bool INV, bool ZIN = false>
__device__ __forceinline__ void pdft16(f2 (&v)[16]) {
  if constexpr (ZIN) {
    static_assert(!INV, "the pruned first stage is the forward one");
    {
      // a = 0
      const f2 c = v[8], s1 = v[4] + v[12], d1 = v[4] - v[12];
      v[0] = c + s1, v[8] = c - s1, v[4] = pk_neg_addmj(c, d1), v[12] = pk_neg_addpj(c, d1);
    }
#pragma unroll
    for (int r1 = 1; r1 < 3; ++r1) {
      // a = d = 0
      const f2 b = v[r1 + 4], c = v[r1 + 8];
      v[r1] = c + b, v[r1 + 8] = c - b, v[r1 + 4] = pk_neg_addmj(c, b), v[r1 + 12] = pk_neg_addpj(c, b);
    }
    {
      // d = 0
      const f2 b = v[7], s0 = v[3] + v[11], d0 = v[3] - v[11];
      v[3] = s0 + b, v[11] = s0 - b, v[7] = pk_addmj(d0, b), v[15] = pk_addpj(d0, b);
    }
  } else {
#pragma unroll
    for (int r1 = 0; r1 < 4; ++r1) pdft4<INV>(v[r1], v[r1 + 4], v[r1 + 8], v[r1 + 12]);
  }
  // W16^m at positive angle; the forward transform multiplies by the conjugates
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, C2 = 0.70710678118654752f;
  const f2 b1 = {C1, S1}, b2 = {C2, C2}, b3 = {S1, C1}, b6 = {-C2, C2}, b9 = {-C1, -S1};
  v[1 + 4] = pk_cmulc<!INV>(v[1 + 4], b1);
  v[2 + 4] = pk_cmulc<!INV>(v[2 + 4], b2);
  v[3 + 4] = pk_cmulc<!INV>(v[3 + 4], b3);
  v[1 + 8] = pk_cmulc<!INV>(v[1 + 8], b2);
  v[2 + 8] = pk_mulj<!INV>(v[2 + 8]);
  v[3 + 8] = pk_cmulc<!INV>(v[3 + 8], b6);
  v[1 + 12] = pk_cmulc<!INV>(v[1 + 12], b3);
  v[2 + 12] = pk_cmulc<!INV>(v[2 + 12], b6);
  v[3 + 12] = pk_cmulc<!INV>(v[3 + 12], b9);
#pragma unroll
  for (int q2 = 0; q2 < 4; ++q2) pdft4<INV>(v[4 * q2], v[4 * q2 + 1], v[4 * q2 + 2], v[4 * q2 + 3]);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b) {
      const f2 t = v[4 * a + b];
      v[4 * a + b] = v[4 * b + a];
      v[4 * b + a] = t;
    }
}

// pass 2 and pass 3 (see fft1024_tail)
template <bool INV>
__device__ __forceinline__ void pfft1024_pass2(f2 (&v)[16], f2* s, const f2* s_tw, int lane, int z) {
  const int bl = lane + (lane >> 4);
#pragma unroll
  for (int r = 0; r < 16; ++r) s[17 * lane + r] = v[r];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = s[bl + 68 * r];
  const int k = lane & 15;
#pragma unroll
  for (int r = 1; r < 16; ++r) v[r] = pk_cmul<INV>(v[r], s_tw[TW_MAIN + 16 * r + k + z]);
  pdft16<INV>(v);
  const int j0 = (lane >> 4) * 256 + k;
  const int bj = j0 + (j0 >> 4);
#pragma unroll
  for (int r = 0; r < 16; ++r) s[bj + 17 * r] = v[r];
}
template <bool INV>
__device__ __forceinline__ void pfft1024_pass3(f2 (&v)[16], f2* s, const f2* s_tw, int lane, int z) {
  const int bl = lane + (lane >> 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[4 * i + r] = s[bl + 68 * i + 272 * r];
  }
  // Twiddles W_1024^(m j), j = lane + 64 i, m = 1, 2, 3: W^(m lane) from the table (three reads per lane instead of
  // twelve - the LDS pipe is the busier one in this kernel) times W_16^(i m), a constant of the unrolled code.
  const int jl = lane + z;
  const f2 t1 = s_tw[jl], t2 = s_tw[2 * jl], t3 = s_tw[3 * jl];
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, C2 = 0.70710678118654752f;
  const f2 b1 = {C1, S1}, b2 = {C2, C2}, b3 = {S1, C1}, b6 = {-C2, C2}, b9 = {-C1, -S1};  // W_16^m at positive angle
  v[1] = pk_cmul<INV>(v[1], t1), v[2] = pk_cmul<INV>(v[2], t2), v[3] = pk_cmul<INV>(v[3], t3);
  v[5] = pk_cmul<INV>(pk_cmulc<!INV>(v[5], b1), t1);
  v[6] = pk_cmul<INV>(pk_cmulc<!INV>(v[6], b2), t2);
  v[7] = pk_cmul<INV>(pk_cmulc<!INV>(v[7], b3), t3);
  v[9] = pk_cmul<INV>(pk_cmulc<!INV>(v[9], b2), t1);
  v[10] = pk_cmul<INV>(pk_mulj<!INV>(v[10]), t2);
  v[11] = pk_cmul<INV>(pk_cmulc<!INV>(v[11], b6), t3);
  v[13] = pk_cmul<INV>(pk_cmulc<!INV>(v[13], b3), t1);
  v[14] = pk_cmul<INV>(pk_cmulc<!INV>(v[14], b6), t2);
  v[15] = pk_cmul<INV>(pk_cmulc<!INV>(v[15], b9), t3);
#pragma unroll
  for (int i = 0; i < 4; ++i) pdft4<INV>(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

// One Griffin-Lim iteration of one frame (utterance b, frame i) by one wave; the windowed synthesis frame (samples
// lpad_even .. lpad_even + WP) is left in the wave's LDS region `s`, viewed as floats.
// Factors of two are not applied where they fall: the analysis side works on 2 X (only its phase is used), the
// synthesis side on 2 Z, and the target magnitudes carry 1/2 and the 1/1024 of the inverse transform (powers of two:
// the same bits as scaling at the end).
// The frame's input: x[r] = samples 2m, 2m+1 at m = lane + 64 r, requested at the start of the frame.  (Requested one
// round ahead, across the overlap-add, the 32 registers it holds there cost more in spilled registers than the
// memory round trip it hides: the other workgroups of the CU fill that gap.)
struct FrameIn {
  f2 x[16];
};
// PRUNE: the window is zero at the sample pairs m < 192 and m >= 832 (r = 0, 1, 2, 13, 14, 15 of every lane), which
// are neither loaded nor multiplied nor fed to the first pass; and the synthesis side computes only the outputs
// that the windowed frame keeps (gl_prunable below).
template <bool PRUNE>
__device__ __forceinline__ void frame_prefetch(const FrameP& p, int b, int i, int lane, FrameIn& in) {
  const FrameGeom q = p.q;
  int lane_a = lane;
  asm volatile("" : "+v"(lane_a));
  __builtin_assume((unsigned)lane_a < 64u);
  // second base 4 KiB further on: the constant part of every global address then fits the instruction offset
  unsigned hi_off = 1024;
  asm volatile("" : "+s"(hi_off));
  const float* src = p.ypad + (size_t)b * q.LP + (size_t)i * q.hop;
  const float* src_hi = src + hi_off;
#pragma unroll
  for (int r = PRUNE ? 3 : 0; r < (PRUNE ? 13 : 16); ++r) {
    // (uniform base + constant, then a 32-bit unsigned lane offset: the scalar-base addressing form)
    // the hop is odd in general: two 4-byte loads
    const float* sr = (r < 8 ? src : src_hi) + 128 * (r & 7);
    in.x[r].x = ld_at<float>(sr, (unsigned)lane_a * 8u), in.x[r].y = ld_at<float>(sr + 1, (unsigned)lane_a * 8u);
  }
}

template <bool PRUNE>
__device__ __forceinline__ void frame_iter(const FrameP& p, int b, int i, float2* s_, const float2* s_tw_,
                                           const float2* s_tw2_, int lane, const FrameIn& in, bool mark) {
  const FrameGeom q = p.q;
  GL_MARK(0);
  f2* s = reinterpret_cast<f2*>(s_);
  const f2* s_tw = reinterpret_cast<const f2*>(s_tw_);
  const f2* s_tw2 = reinterpret_cast<const f2*>(s_tw2_);
  const f2* win2 = reinterpret_cast<const f2*>(p.winpad);
  const size_t frow = (size_t)b * p.F + i;
  int z = 0;
  asm volatile("" : "+v"(z));  // opaque zero: pins loop-invariant table loads inside the loop
  int lane_a = lane, lane_b = lane, lane_s = lane;
  asm volatile("" : "+v"(lane_a), "+v"(lane_b), "+v"(lane_s));
  __builtin_assume((unsigned)lane_a < 64u && (unsigned)lane_b < 64u && (unsigned)lane_s < 64u);
  __builtin_assume((unsigned)z < 8u);
  unsigned hi_off = 1024;
  asm volatile("" : "+s"(hi_off));
  const f2* win_hi = win2 + (hi_off >> 1);

  // the window at this lane's samples first, then the target magnitudes of its bin pairs (k, 1024 - k), k = lane +
  // 64 c: loads return in order, so the window products wait for the (cached) window only and the magnitudes, which
  // come from HBM, land during the analysis transform
  constexpr int R0 = PRUNE ? 3 : 0, R1 = PRUNE ? 13 : 16;
  f2 w[16];
#pragma unroll
  for (int r = R0; r < R1; ++r) w[r] = ld_at<f2>((r < 8 ? win2 : win_hi) + 64 * (r & 7), (unsigned)(lane_a + z) * 8u);
  __builtin_amdgcn_sched_barrier(0);
  f2 mg[8];
  float m512;
  {
    const float* mg0 = p.mag + frow * NBIN;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      mg[c].x = ld_at<float>(mg0 + 64 * c, (unsigned)lane_a * 4u);
      mg[c].y = ld_at<float>(mg0 + (MC - 63) - 64 * c, (unsigned)(63 - lane_a) * 4u);
    }
    m512 = mg0[MC / 2];
  }
  __builtin_amdgcn_sched_barrier(0);
  f2 v[16];
  {
    // ---- analysis: z[m] = (x[2m], x[2m+1]) windowed, first radix-16 pass on the prefetched input
#pragma unroll
    for (int r = R0; r < R1; ++r) v[r] = in.x[r] * w[r];
    GL_MARK(1);
    pdft16<false, PRUNE>(v);
    pfft1024_pass2<false>(v, s, s_tw, lane_a, z);
    pfft1024_pass3<false>(v, s, s_tw, lane_a, z);
    // v[4 c + r] holds bin (lane + 64 c) + 256 r.  Bins below 512 (r < 2) are this lane's own k of the pairs
    // (k, 1024 - k) and stay in registers; only the upper half, which other lanes read as partners, goes to LDS.
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 2; r < 4; ++r) s[lane_a + (lane_a >> 4) + 68 * c + 272 * r] = v[4 * c + r];
  }
  const f2 Z512 = v[2];  // (lane 0: bin 512)
  GL_MARK(2);
  constexpr float SCALE = 1.0f / MC;
#pragma unroll
  for (int c = 0; c < 8; ++c) mg[c] *= 0.5f * SCALE;
  m512 *= SCALE;

  // ---- per-bin work on pairs (k, 1024-k); operands of the eight pairs first (one LDS latency, not eight)
  f2 Zk8[8], Zm8[8], W8[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int k = lane_b + 64 * c;
    Zk8[c] = v[4 * (c & 3) + (c >> 2)], Zm8[c] = s[pidx((MC - k) & (MC - 1))];
    W8[c] = s_tw2[k + z];
  }
  Zm8[0] = lane_b == 0 ? Zk8[0] : Zm8[0];  // bin 0 is its own partner (and a lower bin: not in LDS)
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int k = lane_b + 64 * c;
    const f2 W = W8[c];
    // real-transform post-pass: with E = (Zk + conj Zm)/2, D = (Zk - conj Zm)/2: X[k] = E - i W D,
    // conj X[1024-k] = E + i W D
    const f2 S = pk_addconj(Zk8[c], Zm8[c]), Dd = pk_subconj(Zk8[c], Zm8[c]);
    const f2 P = pk_cmul<false>(Dd, W);
    const f2 Xk = pk_addmj(S, P), Ym = pk_addpj(S, P);  // 2 X[k], 2 conj X[1024-k]
    // unit phasors times the target magnitudes
    const f2 qk = Xk * Xk, qm = Ym * Ym;
    f2 ss;
    ss.x = qk.x + qk.y, ss.y = qm.x + qm.y;
    f2 Xk2, Ym2;
    if (__builtin_expect(ss.x > 1e-30f && ss.x < 1e30f && ss.y > 1e-30f && ss.y < 1e30f, 1)) {
      f2 n;
      n.x = __builtin_amdgcn_rsqf(ss.x), n.y = __builtin_amdgcn_rsqf(ss.y);
      n *= mg[c];
      Xk2 = pk_scale_lo(Xk, n), Ym2 = pk_scale_hi(Ym, n);
    } else {
      // |X|^2 under- or overflows, or X = 0 (np.angle(0) = 0)
      const float2 uk = unit_phase(make_float2(Xk.x, Xk.y)), um = unit_phase(make_float2(Ym.x, Ym.y));
      Xk2 = (f2){uk.x, uk.y} * mg[c].x, Ym2 = (f2){um.x, um.y} * mg[c].y;
    }
    // inverse pre-pass: Z[k] = E' + i O', Z[1024-k] = conj(E' - i O') with E' = (X'k + Y'm)/2,
    // O' = conj(W) (X'k - Y'm)/2
    const f2 A = Xk2 + Ym2, Bd = Xk2 - Ym2;
    const f2 O = pk_cmul<true>(Bd, W);
    const f2 zk = pk_addpj(A, O), zm = pk_conj_addmj(A, O);
    // the lower bin is input lane + 64 c of the inverse transform's first pass: it stays in this lane's registers;
    // the upper bin travels to its lane through LDS
    v[c] = zk;
    // (k = 0 has no partner: it stores its own value into the unused slot 0, which keeps the store unconditional)
    s[pidx(k ? MC - k : 0)] = k ? zm : zk;
  }
  if (lane_b == 0) {
    // Bin 512 is its own partner and its twiddle is -i: X[512] = conj Z[512], Z'[512] = conj X'[512], so the new
    // value is the old one's unit phasor times the magnitude
    const float2 u = unit_phase(make_float2(Z512.x, Z512.y));
    s[pidx(512)] = (f2){u.x, u.y} * m512;
  }

  GL_MARK(3);
  // ---- synthesis: inverse transform, window, windowed frame into the wave's LDS region
#pragma unroll
  for (int r = 8; r < 16; ++r) v[r] = s[lane_s + (lane_s >> 4) + 68 * r];
  pdft16<true>(v);
  pfft1024_pass2<true>(v, s, s_tw, lane_s, z);
  // the window values of the sixteen sample pairs (always inside the table) are requested before the last pass
  // and land during it
  f2 wn[16];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (!(PRUNE && ((r == 0 && c < 3) || (r == 3 && c > 0))))
        wn[4 * c + r] = ld_at<f2>((r < 2 ? win2 : win_hi) + 64 * c + 256 * (r & 1), (unsigned)(lane_s + z) * 8u);
  __builtin_amdgcn_sched_barrier(0);
  pfft1024_pass3<true>(v, s, s_tw, lane_s, z);
  GL_MARK(4);
  {
    const int lpe = q.lpad & ~1;
    // (all lanes have read their last pass from `s` before the first of these stores issues)
    float* dst = reinterpret_cast<float*>(s);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (PRUNE: m < 192 and m >= 832 lie outside the windowed frame; the last pass's sums for them fall away)
        if (PRUNE && ((r == 0 && c < 3) || (r == 3 && c > 0))) continue;
        const int m = lane_s + 64 * c + 256 * r;  // complex index -> samples 2m, 2m+1
        const int n = 2 * m - lpe;
        if (n >= 0 && n < WP) *reinterpret_cast<f2*>(dst + n) = v[4 * c + r] * wn[4 * c + r];
      }
  }
  GL_MARK_NV(5);
}

// Frames strided over the grid, windowed synthesis frames to HBM (stand-alone STFT / ISTFT).
template <int KIND>
__global__ __launch_bounds__(64 * FR_PER_WG, 3) void frame_kernel(FrameP p) {
  __shared__ float2 s_tw[TW_ALL];  // W_1024^m and the second-pass table
  __shared__ float2 s_tw2[516];   // W_2048^k, k <= 512 (real-transform post/pre-processing)
  __shared__ float2 s_data[FR_PER_WG][LDS_PER_FR];
  // (the wave index through readfirstlane: frame index and row pointers are then wave-uniform for the compiler)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  load_tables(s_tw, s_tw2, p.tw, tid);
  __syncthreads();
  const long total = (long)p.B * p.F;
  for (long fi = (long)blockIdx.x * FR_PER_WG + wave; fi < total; fi += (long)gridDim.x * FR_PER_WG) {
    const int b = (int)(fi / p.F), i = (int)(fi % p.F);
    const int Fb = p.n_frames ? min(max(p.n_frames[b], 0), p.F) : p.F;
    if (i >= Fb) continue;  // wave-uniform
    frame_body<KIND, false>(p, b, i, s_data[wave], s_tw, s_tw2, lane);
  }
}

// One Griffin-Lim iteration (or the initial synthesis from the given phase) with the overlap-add
// fused in.  Workgroup (x, b) owns rounds x*C .. x*C+C-1 of utterance b; round j holds frames
// 4j .. 4j+3 (one per wave).  In sample positions a = padded index - lpad, frame i covers
// [i*hop, i*hop + win).  After round j every position below (4j+4)*hop has received all its
// frames: those 4*hop positions are normalised and written, the next win-hop positions become the
// carry.  Sums run in increasing frame order from zero, as the reference's frame-by-frame
// accumulation does.  The run is preceded by `n_halo` rounds that only rebuild the carry - or, with one round per
// workgroup, the carry comes from the left neighbour of the same launch: the carry a round leaves behind holds
// its own frames only (carry_len <= 4 hop), summed from zero in frame order, so it is the same bits either way.
// A workgroup publishes before it waits and its neighbour has the lower block index (dispatched first): no
// wait can be circular.
// PADDED = 1: out[b][g] over the centre-padded signal (what the next iteration reads; the zero
// pad of 'constant' mode is written too); PADDED = 0: out[b][m] = trimmed signal.
struct FusedP {
  FrameP f;
  float* out;
  int C, n_halo, n_rounds;  // rounds per workgroup, halo rounds, rounds per utterance
  int carry_len;            // win - hop
  // One round per workgroup (small workloads, where the halo round would double the work): no halo; a workgroup
  // publishes the carry its frames leave behind as tagged words [b][x][carry_len] and takes the carry of its left
  // neighbour from there.  Null: halo rounds.
  unsigned long long* xbuf;
  unsigned xtag;            // this launch's tag (never repeats on a handle)
  // Several rounds per workgroup without the halo round (`stash` non-null; the window is four hops of 256 .. 292
  // samples): every workgroup publishes the carry of its LAST round in xbuf.  A workgroup whose first round lies in
  // the interior of the utterance starts without a carry: the three hop blocks of that round that need one are
  // not emitted but their frame terms are kept in `stash` ([b][x][6][hop]: block 0 has one term, block 1 two,
  // block 2 three), and when its run is done it takes the left neighbour's carry and adds carry and terms in the
  // usual order - the same bits as with a halo round, one round of frames less per run.
  float* stash;
  // 'reflect' centre padding written by the launch itself: the thread that emits sample m of the signal also
  // writes its images in the head pad (n_fft/2 - m, 1 <= m <= n_fft/2) and in the tail pad (n_fft/2 + 2 L - 2 - m,
  // L - 1 - n_fft/2 <= m <= L - 2); positions outside the signal are not written by their slot's owner.  Only for
  // utterances of full length with L >= n_fft/2 + 2 (one image per pad position): the REFL forms of the kernel.
  // Otherwise the pad is constant, or reflect_pad_kernel fills it between the launches.
};

// ONE: one round per workgroup with the carry from the left neighbour (FusedP::xbuf; batch 1 and other small
// workloads).  Such a launch is as long as its chain of memory round trips, so the iteration form requests the
// frame's samples before the tables, not after them.
template <int KIND, int PADDED, bool PRUNE = false, bool REFL = false, bool ONE = false>
__global__ __launch_bounds__(64 * FR_PER_WG, 3) void gl_fused_kernel(FusedP P) {
  __shared__ float2 s_tw[TW_ALL];
  __shared__ float2 s_tw2[516];
  __shared__ float2 s_data[FR_PER_WG][LDS_PER_FR];
  extern __shared__ float s_carry[];  // [2][carry_len]
  const FrameP& p = P.f;
  const FrameGeom q = p.q;
  // (the wave index through readfirstlane: the compiler then knows that the frame index and every pointer
  // derived from it are wave-uniform and keeps them in scalar registers)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int b = blockIdx.y;
  const int Fb = p.n_frames ? min(max(p.n_frames[b], 0), p.F) : p.F;
  const int L = q.hop * (Fb - 1), Lmax = q.hop * (p.F - 1);
  const int shift = q.lpad - (q.lpad & ~1);
  const int K = (q.win + q.hop - 1) / q.hop;
  [[maybe_unused]] FrameIn fin_early;
  if constexpr (ONE && KIND == K_GL_ITER) {
    if (FR_PER_WG * (int)blockIdx.x + wave < Fb) frame_prefetch<PRUNE>(p, b, FR_PER_WG * (int)blockIdx.x + wave, lane, fin_early);
  }
  load_tables(s_tw, s_tw2, p.tw, tid);
  for (int t = tid; t < 2 * P.carry_len; t += 64 * FR_PER_WG) s_carry[t] = 0.f;
  const int j_first = blockIdx.x * P.C, j_end = min(j_first + P.C, P.n_rounds);
  // A thread's output positions t = tid + 256 u sit at the same offset inside a hop in every round
  // (a round starts on a hop boundary), so the window sum-square of a fully overlapped position
  // is computed once per thread.
  constexpr int WSS_U = 5;
  float wss_full[WSS_U];
#pragma unroll
  for (int u = 0; u < WSS_U; ++u) {
    const int rho = (tid + 64 * FR_PER_WG * u) % q.hop;
    float acc = 0.f;
#pragma unroll
    for (int kk = OLA_MAXK - 1; kk >= 0; --kk) {
      // unconditional (clamped) loads: all of them leave in one burst; behind a condition each would be a
      // branch with its own wait, 40 dependent round trips before the first frame
      const int n = kk * q.hop + rho;
      const float w = p.winpad[min(q.lpad + n, NFFT - 1)];
      acc += (kk < K && n < q.win) ? w * w : 0.f;
    }
    wss_full[u] = acc;
  }
  const float inv_hop = 1.0f / (float)q.hop;
  int cur = 0;
  const bool dmode = !ONE && KIND == K_GL_ITER && P.stash != nullptr;
  constexpr bool xmode = ONE;
  // Interior rounds of the usual geometry (window = 4 hops, 256 <= hop <= 292) take combine_interior below: thread
  // tid owns offset rho = tid of each of the seven hop blocks of the round's span, and the 7 (hop - 256) offsets
  // beyond 256 are dealt one per thread (block tid / (hop - 256), offset 256 + tid % (hop - 256)).
  constexpr int NTHR = 64 * FR_PER_WG;
  const int n_left = q.hop - NTHR;  // offsets of a hop block beyond the first 256
  const bool fast_geom = !xmode && !REFL && q.win == FR_PER_WG * q.hop && P.carry_len == (FR_PER_WG - 1) * q.hop &&
                         n_left >= 0 && (2 * FR_PER_WG - 1) * n_left <= NTHR;
  float wss_left = 0.f;
  if (fast_geom && n_left > 0) {
    const int left_rho = NTHR + tid % n_left;
#pragma unroll
    for (int kk = FR_PER_WG - 1; kk >= 0; --kk) {
      const float w = p.winpad[min(q.lpad + kk * q.hop + min(left_rho, q.hop - 1), NFFT - 1)];
      wss_left += w * w;
    }
  }
  // (dmode) this workgroup starts without a carry and takes its left neighbour's when its run is done
  const bool defer_me = dmode && fast_geom && blockIdx.x > 0 && j_first >= 1 && j_first < j_end &&
                        FR_PER_WG * j_first + FR_PER_WG - 1 < Fb;
  float* stash = defer_me ? P.stash + ((size_t)b * gridDim.x + blockIdx.x) * 6 * q.hop : nullptr;
  const int j_start = max(j_first - ((xmode || defer_me) ? 0 : P.n_halo), 0);  // (nothing before the first frame)
  // (several rounds per workgroup) the next round's samples are requested while this round is combined
  [[maybe_unused]] FrameIn fin_next;
  if constexpr (!ONE && KIND == K_GL_ITER) {
    if (j_start < j_end && FR_PER_WG * j_start + wave < Fb) frame_prefetch<PRUNE>(p, b, FR_PER_WG * j_start + wave, lane, fin_next);
  }
  for (int j = j_start; j < j_end; ++j) {
    __syncthreads();      // tables / carry ready; the previous round's combine has read the frames
    const int i = FR_PER_WG * j + wave;
#ifdef TTSX_GL_TIMING
    const bool mark = KIND == K_GL_ITER && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && j == j_first + 1;
#else
    constexpr bool mark = false;
#endif
    if (i < Fb) {
      if constexpr (KIND == K_GL_ITER) {
        if constexpr (ONE) {
          frame_iter<PRUNE>(p, b, i, s_data[wave], s_tw, s_tw2, lane, fin_early, mark);
        } else {
          frame_iter<PRUNE>(p, b, i, s_data[wave], s_tw, s_tw2, lane, fin_next, mark);
        }
      }
      else frame_body<KIND, true>(p, b, i, s_data[wave], s_tw, s_tw2, lane);
    }
    if constexpr (!ONE && KIND == K_GL_ITER) {
      if (j + 1 < j_end && i + FR_PER_WG < Fb) frame_prefetch<PRUNE>(p, b, i + FR_PER_WG, lane, fin_next);
    }
    __syncthreads();
    GL_MARK_NV(6);
    // ---- combine: positions t = a - 4j*hop in [0, 4*hop + carry_len)
    const float* c_old = s_carry + cur * P.carry_len;
    float* c_new = s_carry + (cur ^ 1) * P.carry_len;
    const int n_out = FR_PER_WG * q.hop;
    constexpr int NT = 64 * FR_PER_WG;
    const int n_tot = n_out + P.carry_len;
    // emit: the round's own 4 hop positions are finished and written; carry_part: the positions behind them
    // become the next carry
    auto combine = [&](const bool emit, const bool carry_part) {
      // U positions per thread at a time: their LDS reads (carry + one sample of each of the four frames, clamped
      // indices, selected afterwards) are all issued before the first use
      constexpr int U = 4;
      const int t_end = carry_part ? n_tot : n_out;
#pragma unroll 1
      for (int t0 = emit ? 0 : (n_out / NT) * NT; t0 < t_end; t0 += NT * U) {
        float a[U][FR_PER_WG + 1];
#pragma unroll
        for (int x = 0; x < U; ++x) {
          const int t = t0 + NT * x + tid;
          a[x][0] = c_old[max(min(t, P.carry_len - 1), 0)];
#pragma unroll
          for (int w = 0; w < FR_PER_WG; ++w)
            a[x][1 + w] = reinterpret_cast<const float*>(s_data[w])[max(min(t - w * q.hop, q.win - 1), 0) + shift];
        }
#pragma unroll
        for (int x = 0; x < U; ++x) {
          const int t = t0 + NT * x + tid;
          if (t >= t_end || (!emit && t < n_out)) continue;
          float acc = t < P.carry_len ? a[x][0] : 0.f;
#pragma unroll
          for (int w = 0; w < FR_PER_WG; ++w) {
            const int n = t - w * q.hop;
            acc += (n >= 0 && n < q.win && FR_PER_WG * j + w < Fb) ? a[x][1 + w] : 0.f;
          }
          if (t >= n_out) {
            c_new[t - n_out] = acc;
            continue;
          }
          const int a_pos = FR_PER_WG * j * q.hop + t;
          const int g = q.lpad + a_pos, m = g - NFFT / 2;
          float y = 0.f;
          if (m >= 0 && m < L) {
            // window sum-square over the frames that exist at this position
            const int th = (int)(((float)t + 0.5f) * inv_hop);  // t / hop (exact: the quotient is below 16)
            const int sl = FR_PER_WG * j + th;
            const int u = t0 / NT + x;
            float wss = 0.f;
            if (u < WSS_U && sl >= K - 1 && sl < Fb) {
              // (a select chain, not an index: keeps the table in registers)
              wss = u == 0 ? wss_full[0] : u == 1 ? wss_full[1] : u == 2 ? wss_full[2] : u == 3 ? wss_full[3] : wss_full[4];
            } else {
              const int rho = t - th * q.hop;
#pragma unroll
              for (int kk = OLA_MAXK - 1; kk >= 0; --kk) {
                const int fi = sl - kk, n = kk * q.hop + rho;
                if (kk < K && fi >= 0 && fi < Fb && n < q.win) {
                  const float w = p.winpad[q.lpad + n];
                  wss += w * w;
                }
              }
            }
            y = wss > 1.17549435e-38f ? acc / wss : acc;
          }
          if (PADDED) {
            if constexpr (REFL) {
              if (m >= 0 && m < L) {
                float* yb = P.out + (size_t)b * q.LP;
                yb[g] = y;
                if (m >= 1 && m <= NFFT / 2) yb[NFFT / 2 - m] = y;
                const int k = L - 2 - m;
                if (k >= 0 && k < NFFT / 2) yb[NFFT / 2 + L + k] = y;
              }
            } else if (g < q.LP) {
              P.out[(size_t)b * q.LP + g] = y;
            }
          } else {
            if (m >= 0 && m < Lmax) P.out[(size_t)b * Lmax + m] = y;
          }
        }
      }
    };
    // ---- the same for a round in the interior of an utterance (all four frames exist, every emitted position is
    // fully overlapped and inside the signal): hop block h = 0 .. 6 of the span receives the carry (h < 3) and
    // frames max(0, h - 3) .. min(3, h), frame w at in-window position (h - w) hop + rho - no range checks, no
    // selects, the block index is a compile-time constant of each pass, and only the operands that exist are read.
    // Same sums in the same order as combine().
    // keep: the round is the first of a run that started without a carry - the blocks that need one (h < 3) are
    // not emitted, their frame terms go to the stash (term w of block h in slot h (h + 1) / 2 + w)
    auto combine_interior = [&](const bool emit, const bool keep) {
      const int H = q.hop;
      const float* fr[FR_PER_WG];
#pragma unroll
      for (int w = 0; w < FR_PER_WG; ++w) fr[w] = reinterpret_cast<const float*>(s_data[w]) + shift;
      float* outp = PADDED ? P.out + (size_t)b * q.LP + q.lpad + (size_t)FR_PER_WG * j * H
                           : P.out + (size_t)b * Lmax + ((size_t)FR_PER_WG * j * H + q.lpad - NFFT / 2);
      auto block_sum = [&](const int h, const int rho) {
        float acc = h < FR_PER_WG - 1 ? c_old[h * H + rho] : 0.f;
#pragma unroll
        for (int w = 0; w < FR_PER_WG; ++w)
          if (w <= h && h - w < FR_PER_WG) acc += fr[w][(h - w) * H + rho];
        return acc;
      };
#pragma unroll
      for (int h = 0; h < 2 * FR_PER_WG - 1; ++h) {
        if (!(emit || h >= FR_PER_WG)) continue;
        if (keep && h < FR_PER_WG - 1) {
#pragma unroll
          for (int w = 0; w <= h; ++w) stash[(h * (h + 1) / 2 + w) * H + tid] = fr[w][(h - w) * H + tid];
          continue;
        }
        const float v = block_sum(h, tid);
        if (h >= FR_PER_WG) c_new[(h - FR_PER_WG) * H + tid] = v;
        else outp[h * H + tid] = wss_full[0] > 1.17549435e-38f ? v / wss_full[0] : v;
      }
      // the offsets beyond 256: one (block, offset) per thread
      const int left_h = n_left > 0 ? tid / n_left : 0, left_rho = NTHR + (n_left > 0 ? tid % n_left : 0);
      if (keep && tid < (FR_PER_WG - 1) * n_left) {
#pragma unroll
        for (int w = 0; w < FR_PER_WG - 1; ++w)
          if (w <= left_h) stash[(left_h * (left_h + 1) / 2 + w) * H + left_rho] = fr[w][(left_h - w) * H + left_rho];
      } else if (tid < (2 * FR_PER_WG - 1) * n_left && (emit || left_h >= FR_PER_WG)) {
        const int h = left_h, rho = left_rho;
        float acc = h < FR_PER_WG - 1 ? c_old[h * H + rho] : 0.f;
#pragma unroll
        for (int w = 0; w < FR_PER_WG; ++w) {
          const int d = h - w;
          const float f = fr[w][min(max(d, 0), FR_PER_WG - 1) * H + rho];
          acc += (d >= 0 && d < FR_PER_WG) ? f : 0.f;
        }
        if (h >= FR_PER_WG) c_new[(h - FR_PER_WG) * H + rho] = acc;
        else outp[h * H + rho] = wss_left > 1.17549435e-38f ? acc / wss_left : acc;
      }
    };
    if (!xmode) {
      if (fast_geom && FR_PER_WG * j + FR_PER_WG - 1 < Fb && (j >= 1 || j < j_first))
        combine_interior(j >= j_first, defer_me && j == j_first);
      else combine(j >= j_first, true);
      cur ^= 1;
      GL_MARK_NV(7);
      continue;
    }
    // ---- one round per workgroup: what the frames leave behind goes to the right neighbour, the carry comes
    // from the left one (zero for the first round of an utterance)
    typedef unsigned long long u64;
    combine(false, true);
    __syncthreads();
    u64* mine = P.xbuf + ((size_t)b * gridDim.x + blockIdx.x) * P.carry_len;
    for (int k = tid; k < P.carry_len; k += NT)
      __hip_atomic_store(mine + k, ((u64)P.xtag << 32) | __float_as_uint(c_new[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float* c_in = s_carry + cur * P.carry_len;
    if (blockIdx.x == 0) {
      for (int k = tid; k < P.carry_len; k += NT) c_in[k] = 0.f;
    } else {
      const u64* left = mine - P.carry_len;
      constexpr int XU = (WP + NT - 1) / NT;  // carry_len < win <= WP
      for (long spin = 0;; ++spin) {
        u64 v[XU];
        bool ok = true;
#pragma unroll
        for (int m = 0; m < XU; ++m) {
          const int k = tid + NT * m;
          v[m] = __hip_atomic_load(left + min(k, P.carry_len - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && (k >= P.carry_len || (unsigned)(v[m] >> 32) == P.xtag);
        }
        // (the left neighbour was dispatched before this workgroup and publishes before it waits; the bound is
        // a last resort that makes a broken run visible instead of hanging it)
        if (__all(ok) || spin > 40000000) {
#pragma unroll
          for (int m = 0; m < XU; ++m)
            if (tid + NT * m < P.carry_len) c_in[tid + NT * m] = ok ? __uint_as_float((unsigned)v[m]) : __builtin_nanf("");
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    combine(true, false);
  }
  if (dmode) {
    typedef unsigned long long u64;
    constexpr int NT = 64 * FR_PER_WG;
    const int H = q.hop;
    // the carry of the last round, for the right neighbour
    __syncthreads();
    const float* c_fin = s_carry + cur * P.carry_len;
    u64* mine = P.xbuf + ((size_t)b * gridDim.x + blockIdx.x) * P.carry_len;
    for (int k = tid; k < P.carry_len; k += NT)
      __hip_atomic_store(mine + k, ((u64)P.xtag << 32) | __float_as_uint(c_fin[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (defer_me) {
      // the left neighbour's (it has the lower block index and publishes before it waits; bounded all the same)
      float* c_in = s_carry + (cur ^ 1) * P.carry_len;
      const u64* left = mine - P.carry_len;
      constexpr int XU = (WP + NT - 1) / NT;  // carry_len < win <= WP
      for (long spin = 0;; ++spin) {
        u64 v[XU];
        bool ok = true;
#pragma unroll
        for (int m = 0; m < XU; ++m) {
          const int k = tid + NT * m;
          v[m] = __hip_atomic_load(left + min(k, P.carry_len - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && (k >= P.carry_len || (unsigned)(v[m] >> 32) == P.xtag);
        }
        if (__all(ok) || spin > 40000000) {
#pragma unroll
          for (int m = 0; m < XU; ++m)
            if (tid + NT * m < P.carry_len) c_in[tid + NT * m] = ok ? __uint_as_float((unsigned)v[m]) : __builtin_nanf("");
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      // the three blocks of the first round that were kept back: carry, then the frame terms in frame order
      float* outp = PADDED ? P.out + (size_t)b * q.LP + q.lpad + (size_t)FR_PER_WG * j_first * H
                           : P.out + (size_t)b * Lmax + ((size_t)FR_PER_WG * j_first * H + q.lpad - NFFT / 2);
#pragma unroll
      for (int h = 0; h < FR_PER_WG - 1; ++h) {
        float acc = c_in[h * H + tid];
#pragma unroll
        for (int w = 0; w <= h; ++w) acc += stash[(h * (h + 1) / 2 + w) * H + tid];
        outp[h * H + tid] = wss_full[0] > 1.17549435e-38f ? acc / wss_full[0] : acc;
      }
      if (tid < (FR_PER_WG - 1) * n_left) {
        const int h = tid / n_left, rho = NTHR + tid % n_left;
        float acc = c_in[h * H + rho];
#pragma unroll
        for (int w = 0; w < FR_PER_WG - 1; ++w)
          if (w <= h) acc += stash[(h * (h + 1) / 2 + w) * H + rho];
        outp[h * H + rho] = wss_left > 1.17549435e-38f ? acc / wss_left : acc;
      }
    }
  }
  if (PADDED) {
    // the first lpad samples and the tail beyond the last round belong to nobody's slot
    // (with the reflect images written above: only what lies beyond the tail pad)
    if (blockIdx.x == 0 && !REFL)
      for (int g = tid; g < q.lpad; g += 64 * FR_PER_WG) P.out[(size_t)b * q.LP + g] = 0.f;
    if (j_end == P.n_rounds)
      for (int g = (REFL ? max(q.lpad + P.n_rounds * FR_PER_WG * q.hop, NFFT + L) : q.lpad + P.n_rounds * FR_PER_WG * q.hop) + tid;
           g < q.LP; g += 64 * FR_PER_WG)
        P.out[(size_t)b * q.LP + g] = 0.f;
  }
}

// Overlap-add of the windowed frames, window-sum-square normalisation (where it exceeds float32
// tiny) and the zero part of the centre padding.  One workgroup per hop-sized slot s: padded
// indices g = lpad + s*hop + rho, rho < hop, receive frame s-k at in-window position k*hop + rho
// for k = 0 .. ceil(win/hop)-1.  Frames are summed in increasing order in fp32, as the reference's
// frame-by-frame accumulation does.  No divisions, every access coalesced.
// PADDED = 1: out[b][g] over the centre-padded signal (what the next STFT reads), zeros in the pad;
// PADDED = 0: out[b][m] = trimmed signal (m = g - n_fft/2), zero beyond the utterance's own length.
constexpr int OLA_SLOTS = 8;
template <int PADDED>
__global__ __launch_bounds__(320) void ola_kernel(float* __restrict__ out, const float* __restrict__ yw,
                                                  const float* __restrict__ winpad,
                                                  const int32_t* __restrict__ n_frames, int B, int F, FrameGeom q,
                                                  int n_slots) {
  const int b = blockIdx.y;
  const int Fb = n_frames ? min(max(n_frames[b], 0), F) : F;
  const int L = q.hop * (Fb - 1), Lmax = q.hop * (F - 1);
  const float* ywb = yw + (size_t)b * F * WP;
  const int K = (q.win + q.hop - 1) / q.hop;
  const int shift = q.lpad - (q.lpad & ~1);
  for (int s = blockIdx.x * OLA_SLOTS; s < min((int)(blockIdx.x + 1) * OLA_SLOTS, n_slots); ++s)
  for (int rho = threadIdx.x; rho < q.hop; rho += blockDim.x) {
    const int g = q.lpad + s * q.hop + rho;  // padded-buffer index
    const int m = g - NFFT / 2;              // index in the trimmed signal
    float acc = 0.f, wss = 0.f;
    if (m >= 0 && m < L) {
#pragma unroll
      for (int kk = OLA_MAXK - 1; kk >= 0; --kk) {  // descending k = ascending frame index
        const int i = s - kk, n = kk * q.hop + rho;
        if (kk < K && i >= 0 && i < Fb && n < q.win) {
          acc += ywb[(size_t)i * WP + n + shift];
          const float w = winpad[q.lpad + n];
          wss += w * w;
        }
      }
      if (wss > 1.17549435e-38f) acc /= wss;
    }
    if (PADDED) {
      if (g < q.LP) out[(size_t)b * q.LP + g] = acc;
    } else {
      if (m >= 0 && m < Lmax) out[(size_t)b * Lmax + m] = acc;
    }
  }
  if (PADDED && blockIdx.x == 0) {
    // the first lpad samples and the tail beyond the last slot belong to nobody's slot
    for (int g = threadIdx.x; g < q.lpad; g += blockDim.x) out[(size_t)b * q.LP + g] = 0.f;
    for (int g = q.lpad + n_slots * q.hop + threadIdx.x; g < q.LP; g += blockDim.x) out[(size_t)b * q.LP + g] = 0.f;
  }
}

// 'reflect' centre padding: pad samples mirror the interior of the signal (np.pad mode='reflect')
__global__ void reflect_pad_kernel(float* __restrict__ ypad, const int32_t* __restrict__ n_frames, int B, int F,
                                   FrameGeom q) {
  const int b = blockIdx.y;
  const int Fb = n_frames ? min(max(n_frames[b], 0), F) : F;
  const int L = q.hop * (Fb - 1);
  if (L <= 1) return;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. n_fft-1: first half = head pad, second = tail pad
  if (t >= NFFT) return;
  const int g = t < NFFT / 2 ? t : NFFT / 2 + L + (t - NFFT / 2);
  int m = g - NFFT / 2;
  const int period = 2 * (L - 1);
  m %= period;
  if (m < 0) m += period;
  if (m >= L) m = period - m;
  float* yb = ypad + (size_t)b * q.LP;
  yb[g] = yb[NFFT / 2 + m];
}

// centre padding of a given signal y[B][hop*(F-1)] (unit STFT entry point)
__global__ void pad_signal_kernel(float* __restrict__ ypad, const float* __restrict__ y, int B, int F, FrameGeom q) {
  const int b = blockIdx.y;
  const int L = q.hop * (F - 1);
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= q.LP) return;
  int m = g - NFFT / 2;
  float v = 0.f;
  if (m >= 0 && m < L) {
    v = y[(size_t)b * L + m];
  } else if (q.pad_mode == TTSX_PAD_REFLECT && L > 1 && g < NFFT + L) {
    const int period = 2 * (L - 1);
    m %= period;
    if (m < 0) m += period;
    if (m >= L) m = period - m;
    v = y[(size_t)b * L + m];
  }
  ypad[(size_t)b * q.LP + g] = v;
}

// de-emphasis y[n] = x[n] + a*y[n-1] (lfilter([1],[1,-a])), in fp64 like the reference, as a blocked scan:
//   1. every thread runs the recurrence over its own 64-sample segment from a zero state; the value it ends on is
//      what the segment contributes to the state right behind it;
//   2. the state a segment starts from is the sum of the 32 segments before it, each damped by a^64 per segment
//      of distance (the impulse response a^k is below 1e-17 after 2048 taps for a = 0.98, so older input is
//      beyond fp64 resolution);
//   3. the recurrence again from that state, this time storing.
// A workgroup (256 segments + 32 halo segments, whose end values the first 32 threads compute as well) stages
// its 2048 + 16384 input samples in LDS with coalesced loads, one pad word per segment so that lanes at the same
// offset of their segments hit different banks, overwrites them in place in step 3 and writes them back coalesced.
// Serial chain per thread: 64 + 32 + 64 steps (a thread walking a 2048-sample warm-up before its segment, as an
// earlier version did, pays 2304 dependent steps: 300 us for 64 x 220 000 samples against 20 us).
constexpr int DE_SEG = 64, DE_HALO = 32, DE_SEGS_PER_WG = 256;
constexpr int DE_STAGE = (DE_HALO + DE_SEGS_PER_WG) * DE_SEG;          // staged samples
constexpr int DE_LDS_FLOATS = (DE_HALO + DE_SEGS_PER_WG) * (DE_SEG + 1);
__device__ __forceinline__ int de_idx(int j) { return j + (j >> 6); }
__device__ __forceinline__ double de_zero_state_end(const float* s_de, int j0, double a) {
  double acc = 0.0;
#pragma unroll 16
  for (int k = 0; k < DE_SEG; ++k) acc = (double)s_de[de_idx(j0 + k)] + a * acc;
  return acc;
}
__global__ __launch_bounds__(DE_SEGS_PER_WG) void deemph_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                               size_t x_pitch, const int32_t* __restrict__ n_frames,
                                                               int B, int F, int hop, float coef) {
  extern __shared__ float s_de[];
  __shared__ double s_end[DE_HALO + DE_SEGS_PER_WG];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int Fb = n_frames ? min(max(n_frames[b], 0), F) : F;
  const int Lmax = hop * (F - 1), L = hop * (Fb - 1);
  const int base = (blockIdx.x * DE_SEGS_PER_WG - DE_HALO) * DE_SEG;  // first sample of the staged window
  const float* xb = x + (size_t)b * x_pitch;
  float* ob = out + (size_t)b * Lmax;
  for (int j = tid; j < DE_STAGE; j += DE_SEGS_PER_WG) {
    const int n = base + j;
    s_de[de_idx(j)] = (n >= 0 && n < Lmax) ? xb[n] : 0.f;   // (before the signal: zeros, i.e. a zero initial state)
  }
  __syncthreads();
  const double a = (double)coef;
  double r = a;  // a^64
#pragma unroll
  for (int i = 0; i < 6; ++i) r *= r;
  const int j0 = (DE_HALO + tid) * DE_SEG;  // this thread's segment inside the window
  s_end[DE_HALO + tid] = de_zero_state_end(s_de, j0, a);
  if (tid < DE_HALO) s_end[tid] = de_zero_state_end(s_de, tid * DE_SEG, a);
  __syncthreads();
  double acc = 0.0;
#pragma unroll 8
  for (int m = DE_HALO; m >= 1; --m) acc = r * acc + s_end[DE_HALO + tid - m];
#pragma unroll 16
  for (int k = 0; k < DE_SEG; ++k) {
    acc = (double)s_de[de_idx(j0 + k)] + a * acc;
    s_de[de_idx(j0 + k)] = (float)acc;   // (only this thread reads or writes its segment from here on)
  }
  __syncthreads();
  for (int j = DE_HALO * DE_SEG + tid; j < DE_STAGE; j += DE_SEGS_PER_WG) {
    const int n = base + j;
    if (n < Lmax) ob[n] = n < L ? s_de[de_idx(j)] : 0.f;
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
  return exp2f((s + a.ref_level_db) * (0.05f * 3.32192809488736f));  // 10^(dB/20)
}

// x^p for x > 0 (the library powf spends most of its instructions on cases that cannot occur here; the common
// exponent 1.5 is a multiply and a square root, anything else goes through exp2 / log2: relative error ~2e-6)
__device__ __forceinline__ float pow_pos(float x, float p) {
  if (p == 1.5f) return x * sqrtf(x);
  if (p == 1.0f) return x;
  return exp2f(p * log2f(x));
}

constexpr int MAG_ROWS = 8;
__global__ __launch_bounds__(256) void spec_to_mag_kernel(MagP p) {
  // amplitudes of the workgroup's rows, channel-major: the MAG_ROWS values of a channel are two 16-byte
  // broadcast reads in the contraction below instead of MAG_ROWS scalar ones
  __shared__ __attribute__((aligned(16))) float s_amp[128][MAG_ROWS];
  static_assert(MAG_ROWS == 8, "the contraction reads the rows of a channel as two float4");
  const long r0 = (long)blockIdx.x * MAG_ROWS;
  const int tid = threadIdx.x;
  if (p.inv_basis) {
    for (int i = tid; i < MAG_ROWS * p.C; i += 256) {
      const int rr = i / p.C, c = i % p.C;
      s_amp[c][rr] = (r0 + rr < p.n_rows) ? denorm_to_amp(p.spec[(r0 + rr) * p.C + c], p.ap) : 0.f;
    }
    __syncthreads();
    for (int k = tid; k < p.n_freq; k += 256) {
      float acc[MAG_ROWS];
#pragma unroll
      for (int rr = 0; rr < MAG_ROWS; ++rr) acc[rr] = 0.f;
      const float* brow = p.inv_basis + (size_t)k * p.C;
      auto channel = [&](float w, int c) {
        const float4 a0 = *reinterpret_cast<const float4*>(&s_amp[c][0]);
        const float4 a1 = *reinterpret_cast<const float4*>(&s_amp[c][4]);
        acc[0] += w * a0.x, acc[1] += w * a0.y, acc[2] += w * a0.z, acc[3] += w * a0.w;
        acc[4] += w * a1.x, acc[5] += w * a1.y, acc[6] += w * a1.z, acc[7] += w * a1.w;
      };
      if ((p.C & 3) == 0) {
        // (a row of the basis is C floats from a 16-byte boundary: four channels per load)
        const float4* brow4 = reinterpret_cast<const float4*>(brow);
#pragma unroll 4
        for (int c4 = 0; c4 < p.C / 4; ++c4) {
          const float4 w = brow4[c4];
          channel(w.x, 4 * c4), channel(w.y, 4 * c4 + 1), channel(w.z, 4 * c4 + 2), channel(w.w, 4 * c4 + 3);
        }
      } else {
        for (int c = 0; c < p.C; ++c) channel(brow[c], c);
      }
#pragma unroll
      for (int rr = 0; rr < MAG_ROWS; ++rr)
        if (r0 + rr < p.n_rows) p.mag[(r0 + rr) * p.n_freq + k] = pow_pos(fmaxf(1e-10f, acc[rr]), p.ap.power);
    }
  } else {
    for (int rr = 0; rr < MAG_ROWS; ++rr) {
      if (r0 + rr >= p.n_rows) break;
      for (int k = tid; k < p.n_freq; k += 256)
        p.mag[(r0 + rr) * p.n_freq + k] = pow_pos(denorm_to_amp(p.spec[(r0 + rr) * p.C + k], p.ap), p.ap.power);
    }
  }
}

// Twiddles W_2048^m (m < 1024) and the periodic Hann window of win_length (scipy.signal.get_window('hann', win,
// fftbins=True)) zero-padded centrally to n_fft, evaluated in fp64 and rounded once.  Every entry point builds
// its tables with this kernel, so that they are the same bits whichever way a caller comes in.
__global__ void init_tables_kernel(float2* __restrict__ tw, float* __restrict__ winpad, int win_length) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < MC) {
    const double a = -2.0 * M_PI * (double)i / (double)NFFT;
    tw[i] = make_float2((float)cos(a), (float)sin(a));
  }
  if (i < NFFT) {
    const int n = i - (NFFT - win_length) / 2;
    winpad[i] = (n >= 0 && n < win_length) ? (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)win_length)) : 0.f;
  }
}

int check_stft_params(const ttsx_stft_params* sp, int B, int F) {
  TTSX_ARG(sp, "stft params: null");
  TTSX_ARG(sp->n_fft == NFFT, "n_fft=%d unsupported (only %d)", sp->n_fft, NFFT);
  TTSX_ARG(sp->win_length >= 2 && sp->win_length <= WP - 2, "win_length=%d outside 2..%d", sp->win_length, WP - 2);
  TTSX_ARG(sp->hop_length >= 1 && sp->hop_length <= sp->win_length, "hop_length=%d", sp->hop_length);
  TTSX_ARG(ceil_div(sp->win_length, sp->hop_length) <= OLA_MAXK, "hop_length=%d: more than %d frames overlap",
           sp->hop_length, OLA_MAXK);
  TTSX_ARG(sp->pad_mode == TTSX_PAD_CONSTANT || sp->pad_mode == TTSX_PAD_REFLECT, "pad_mode=%d", sp->pad_mode);
  TTSX_ARG(B >= 1 && F >= 2, "bad shape B=%d F=%d (need at least 2 frames)", B, F);
  return TTSX_OK;
}

FrameGeom make_geom(const ttsx_stft_params* sp, int F) {
  FrameGeom q;
  q.hop = sp->hop_length, q.win = sp->win_length, q.lpad = (NFFT - sp->win_length) / 2, q.pad_mode = sp->pad_mode;
  q.LP = round_up(NFFT + sp->hop_length * (F - 1), 4);
  return q;
}

int check_gl_args(const float* mag, const float* init_phase, const float* wav, int iters, float preemph) {
  TTSX_ARG(mag && init_phase && wav, "ttsx_griffin_lim: null argument");
  TTSX_ARG(iters >= 0, "ttsx_griffin_lim: iters=%d", iters);
  // (the de-emphasis kernel cuts the filter's impulse response off after 2048 taps)
  TTSX_ARG(pow(fabs((double)preemph), 2048.0) < 1e-9,
           "ttsx_griffin_lim: preemphasis %g too close to 1 (|a|^2048 must be below 1e-9, i.e. |a| <= 0.9899)", preemph);
  return TTSX_OK;
}

// The launches of a whole Griffin-Lim run on `st`: initial synthesis from the given phase, `iters` iterations
// ping-ponging between the two padded signals ybuf[0] / ybuf[1] (B * q.LP floats each), de-emphasis into wav.
// Every padded position of an utterance is written by each launch before the next one reads it, so the buffers
// need not be cleared.  Nothing here allocates or synchronises.
// How a run is cut into workgroups.
struct GlPlan {
  int n_rounds, n_halo, carry_len, C;
  bool exchange;      // one round per workgroup and the carry fits the neighbour scheme (see FusedP::xbuf)
  bool defer;         // several rounds per workgroup, no halo round (see FusedP::stash)
  size_t xbuf_words;  // carry words of such a launch
  size_t stash_floats;
};
GlPlan gl_plan(const ttsx_stft_params* sp, int B, int F) {
  const FrameGeom q = make_geom(sp, F);
  const size_t Lmax = (size_t)sp->hop_length * (F - 1);
  GlPlan pl{};
  // slots that can hold signal samples: padded indices lpad .. n_fft/2 + Lmax; a round = 4 slots
  const int n_slots = ceil_div(NFFT / 2 + (int)Lmax - q.lpad, q.hop);
  pl.n_rounds = ceil_div(n_slots, FR_PER_WG);
  pl.n_halo = ceil_div(ceil_div(q.win, q.hop) - 1, FR_PER_WG);
  pl.carry_len = q.win - q.hop;
  // rounds per workgroup: long runs amortise the halo, short ones fill the chip; as many workgroups as are
  // resident at once (three per CU, bounded by LDS), so that no second, mostly empty wave of them runs
  constexpr int WGS_PER_CU = 3, CUS = 256;
  pl.C = std::max(1, ceil_div(pl.n_rounds, std::max(1, CUS * WGS_PER_CU / B)));
  pl.exchange = pl.C == 1 && pl.n_halo >= 1 && pl.carry_len >= 1 && pl.carry_len <= FR_PER_WG * q.hop;
  pl.xbuf_words = (size_t)B * pl.n_rounds * pl.carry_len;
  pl.defer = pl.C > 1 && pl.n_halo == 1 && q.win == FR_PER_WG * q.hop && q.hop >= 64 * FR_PER_WG &&
             (2 * FR_PER_WG - 1) * (q.hop - 64 * FR_PER_WG) <= 64 * FR_PER_WG;
  if (pl.defer) {
    const size_t runs = (size_t)B * ceil_div(pl.n_rounds, pl.C);
    pl.xbuf_words = runs * pl.carry_len;
    pl.stash_floats = round_up((int)(runs * 6 * q.hop), 2);
  }
  return pl;
}

// The iteration's frames can skip the sample pairs m < 192 and m >= 832 of the 1024: the window (samples lpad ..
// lpad + win of the transform length) is zero there on the analysis side, and the synthesis side keeps samples
// lpad_even .. lpad_even + WP only.
bool gl_prunable(const FrameGeom& q) {
  const int lpe = q.lpad & ~1;
  return q.lpad >= 2 * 192 && q.lpad + q.win <= 2 * 832 && lpe + WP <= 2 * 832;
}

// One padded iteration launch in the form the plan asks for.
void gl_iter_padded(bool one, bool refl, bool prune, dim3 grid, size_t smem, hipStream_t st, const FusedP& P) {
  constexpr int NT = 64 * FR_PER_WG;
  const int form = (one ? 4 : 0) | (refl ? 2 : 0) | (prune ? 1 : 0);
  switch (form) {
    case 0: gl_fused_kernel<K_GL_ITER, 1, false, false, false><<<grid, NT, smem, st>>>(P); break;
    case 1: gl_fused_kernel<K_GL_ITER, 1, true, false, false><<<grid, NT, smem, st>>>(P); break;
    case 2: gl_fused_kernel<K_GL_ITER, 1, false, true, false><<<grid, NT, smem, st>>>(P); break;
    case 3: gl_fused_kernel<K_GL_ITER, 1, true, true, false><<<grid, NT, smem, st>>>(P); break;
    case 4: gl_fused_kernel<K_GL_ITER, 1, false, false, true><<<grid, NT, smem, st>>>(P); break;
    case 5: gl_fused_kernel<K_GL_ITER, 1, true, false, true><<<grid, NT, smem, st>>>(P); break;
    case 6: gl_fused_kernel<K_GL_ITER, 1, false, true, true><<<grid, NT, smem, st>>>(P); break;
    default: gl_fused_kernel<K_GL_ITER, 1, true, true, true><<<grid, NT, smem, st>>>(P); break;
  }
}

// `xbuf` / `xtag`: carry words and launch counter of a handle (null: always halo rounds).
int enqueue_griffin_lim(const float2* tw, const float* winpad, float* const ybuf[2], const float* mag,
                        const float* init_phase, const int32_t* n_frames, int B, int F, const ttsx_stft_params* sp,
                        int iters, float preemph, float* wav, hipStream_t st, unsigned long long* xbuf = nullptr,
                        unsigned* xtag = nullptr) {
  const FrameGeom q = make_geom(sp, F);
  const size_t Lmax = (size_t)sp->hop_length * (F - 1);
  FusedP P{};
  P.f.mag = mag, P.f.phase01 = init_phase, P.f.n_frames = n_frames, P.f.tw = tw, P.f.winpad = winpad;
  P.f.B = B, P.f.F = F, P.f.q = q;
  const GlPlan pl = gl_plan(sp, B, F);
  P.n_rounds = pl.n_rounds, P.n_halo = pl.n_halo, P.carry_len = pl.carry_len, P.C = pl.C;
  const bool exchange = pl.exchange && xbuf && xtag;
  const bool defer = pl.defer && xbuf && xtag;  // (the iteration launches; the initial synthesis keeps its halo round)
  if (exchange) P.xbuf = xbuf;
  const dim3 grid(ceil_div(P.n_rounds, P.C), B), refl_grid(NFFT / 256, B);
  const size_t smem = 2 * (size_t)std::max(P.carry_len, 1) * sizeof(float);
  const bool reflect = sp->pad_mode == TTSX_PAD_REFLECT;
  // (utterances of full length that are long enough for one image per pad position: the launches write the pad)
  const bool own_pad = reflect && !n_frames && !defer && Lmax >= (size_t)NFFT / 2 + 2;
  const bool direct = preemph == 0.f;  // without de-emphasis the last launch writes the trimmed signal itself
  const bool prune = gl_prunable(q);
  int cur = 0;
  P.out = (direct && iters == 0) ? wav : ybuf[cur];
  if (exchange) P.xtag = ++*xtag;
  if (direct && iters == 0) {
    if (exchange) gl_fused_kernel<K_GL_INIT, 0, false, false, true><<<grid, 64 * FR_PER_WG, smem, st>>>(P);
    else gl_fused_kernel<K_GL_INIT, 0><<<grid, 64 * FR_PER_WG, smem, st>>>(P);
  } else if (own_pad) {
    if (exchange) gl_fused_kernel<K_GL_INIT, 1, false, true, true><<<grid, 64 * FR_PER_WG, smem, st>>>(P);
    else gl_fused_kernel<K_GL_INIT, 1, false, true><<<grid, 64 * FR_PER_WG, smem, st>>>(P);
  } else {
    if (exchange) gl_fused_kernel<K_GL_INIT, 1, false, false, true><<<grid, 64 * FR_PER_WG, smem, st>>>(P);
    else gl_fused_kernel<K_GL_INIT, 1><<<grid, 64 * FR_PER_WG, smem, st>>>(P);
  }
  TTSX_LAUNCH_CHECK();
  for (int it = 0; it < iters; ++it) {
    if (reflect && !own_pad) reflect_pad_kernel<<<refl_grid, 256, 0, st>>>(ybuf[cur], n_frames, B, F, q);
    P.f.ypad = ybuf[cur];
    if (defer) P.xbuf = xbuf, P.stash = reinterpret_cast<float*>(xbuf + pl.xbuf_words);
    if (exchange || defer) P.xtag = ++*xtag;
    const bool last = it + 1 == iters;
    if (last && direct) {
      P.out = wav;
      if (exchange && prune) gl_fused_kernel<K_GL_ITER, 0, true, false, true><<<grid, 64 * FR_PER_WG, smem, st>>>(P);
      else if (exchange) gl_fused_kernel<K_GL_ITER, 0, false, false, true><<<grid, 64 * FR_PER_WG, smem, st>>>(P);
      else if (prune) gl_fused_kernel<K_GL_ITER, 0, true><<<grid, 64 * FR_PER_WG, smem, st>>>(P);
      else gl_fused_kernel<K_GL_ITER, 0><<<grid, 64 * FR_PER_WG, smem, st>>>(P);
    } else {
      P.out = ybuf[cur ^ 1];
      gl_iter_padded(exchange, own_pad, prune, grid, smem, st, P);
    }
    TTSX_LAUNCH_CHECK();
    cur ^= 1;
  }
  if (!direct) {
    // the IIR reads the trimmed part of the last padded signal
    const int nseg = (int)((Lmax + DE_SEG - 1) / DE_SEG);
    deemph_kernel<<<dim3(ceil_div(nseg, DE_SEGS_PER_WG), B), DE_SEGS_PER_WG, DE_LDS_FLOATS * sizeof(float), st>>>(
        wav, ybuf[cur] + NFFT / 2, (size_t)q.LP, n_frames, B, F, sp->hop_length, preemph);
    TTSX_LAUNCH_CHECK();
  }
  return TTSX_OK;
}

// (the de-emphasis kernel stages ~74 KiB of samples per workgroup in dynamic LDS: the limit is raised for the
// current device whenever a handle is created on it)
int raise_deemph_lds() {
  TTSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deemph_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(DE_LDS_FLOATS * sizeof(float))));
  return TTSX_OK;
}

unsigned frame_grid(long total) { return (unsigned)std::min<long>((total + FR_PER_WG - 1) / FR_PER_WG, 256 * 12); }

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


// ------------------------------------------------------------------ vocoder handle
// Everything a Griffin-Lim / STFT / ISTFT call needs beyond its arguments lives in the handle: the twiddle and
// window tables of its STFT geometry and the scratch signals, all on the device that was current at create.
// Calls on one handle are stream-ordered and do not synchronise; two handles never share anything, so two
// request threads with a handle each can overlap their calls.
struct ttsx_vocoder {
  ttsx_stft_params sp;
  int device;
  float2* tw;      // W_2048^m, m < 1024
  float* winpad;   // [2048]
  float* yw;       // windowed frames of the stand-alone ISTFT
  float* ypad;     // two padded signals back to back (an iteration reads one and writes the other)
  size_t cap_yw, cap_ypad;
  // small workloads: carries handed from workgroup to workgroup inside a launch (FusedP::xbuf), as words tagged
  // with a launch counter that never repeats between two clears of the buffer
  unsigned long long* xbuf;
  size_t cap_xbuf;
  unsigned xtag;
};

namespace {

// Grow a scratch buffer.  The old one may still be read by work already enqueued on `st`: drain it first.
int grow(float** buf, size_t* cap, size_t n, hipStream_t st) {
  if (n <= *cap) return TTSX_OK;
  if (*buf) {
    TTSX_HIP(hipStreamSynchronize(st));
    (void)hipFree(*buf);
  }
  *buf = nullptr, *cap = 0;
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(buf), n * sizeof(float)));
  *cap = n;
  return TTSX_OK;
}

int check_device(const ttsx_vocoder* v, const char* who) {
  int dev = -1;
  TTSX_HIP(hipGetDevice(&dev));
  TTSX_ARG(dev == v->device, "%s: the handle belongs to device %d, the current device is %d", who, v->device, dev);
  return TTSX_OK;
}

}  // namespace

extern "C" int ttsx_vocoder_create(const ttsx_stft_params* sp, void* stream, ttsx_vocoder** out) {
  TTSX_ARG(out, "ttsx_vocoder_create: null argument");
  int rc = check_stft_params(sp, 1, 2);
  if (rc) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  ttsx_vocoder* v = new ttsx_vocoder();
  memset(v, 0, sizeof(*v));
  ScopeExit on_fail([&] { (void)ttsx_vocoder_destroy(v); });
  v->sp = *sp;
  TTSX_HIP(hipGetDevice(&v->device));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&v->tw), MC * sizeof(float2)));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&v->winpad), NFFT * sizeof(float)));
  init_tables_kernel<<<NFFT / 256, 256, 0, st>>>(v->tw, v->winpad, sp->win_length);
  TTSX_LAUNCH_CHECK();
  if ((rc = raise_deemph_lds())) return rc;
  on_fail.dismiss();
  *out = v;
  return TTSX_OK;
}

extern "C" int ttsx_vocoder_destroy(ttsx_vocoder* v) {
  if (!v) return TTSX_OK;
  float* bufs[] = {reinterpret_cast<float*>(v->tw), v->winpad, v->yw, v->ypad, reinterpret_cast<float*>(v->xbuf)};
  for (float* b : bufs)
    if (b) (void)hipFree(b);   // (hipFree waits for the device: nothing enqueued on the handle is cut short)
  delete v;
  return TTSX_OK;
}

extern "C" int ttsx_vocoder_griffin_lim(ttsx_vocoder* v, const float* mag, const float* init_phase,
                                        const int32_t* n_frames, int B, int F, int iters, float preemph, float* wav,
                                        void* stream) {
  TTSX_ARG(v, "ttsx_vocoder_griffin_lim: null handle");
  int rc = check_gl_args(mag, init_phase, wav, iters, preemph);
  if (rc || (rc = check_stft_params(&v->sp, B, F)) || (rc = check_device(v, "ttsx_vocoder_griffin_lim"))) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t n_pad = (size_t)B * make_geom(&v->sp, F).LP;
  if ((rc = grow(&v->ypad, &v->cap_ypad, 2 * n_pad, st))) return rc;
  float* const ybuf[2] = {v->ypad, v->ypad + n_pad};
  const GlPlan pl = gl_plan(&v->sp, B, F);
  if (pl.exchange || pl.defer) {
    const bool wrap = v->xtag > 0xfffffff0u - (unsigned)(iters + 1);
    const size_t words = pl.xbuf_words + pl.stash_floats / 2;  // (the stash of the no-halo scheme behind the carries)
    if (words > v->cap_xbuf || wrap) {
      if (v->xbuf) {
        TTSX_HIP(hipStreamSynchronize(st));
        if (words > v->cap_xbuf) (void)hipFree(v->xbuf), v->xbuf = nullptr, v->cap_xbuf = 0;
      }
      if (!v->xbuf) {
        TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&v->xbuf), words * sizeof(unsigned long long)));
        v->cap_xbuf = words;
      }
      TTSX_HIP(hipMemsetAsync(v->xbuf, 0, v->cap_xbuf * sizeof(unsigned long long), st));  // tag 0 matches no launch
      v->xtag = 0;
    }
  }
  return enqueue_griffin_lim(v->tw, v->winpad, ybuf, mag, init_phase, n_frames, B, F, &v->sp, iters, preemph, wav, st,
                             (pl.exchange || pl.defer) ? v->xbuf : nullptr, &v->xtag);
}

extern "C" int ttsx_vocoder_stft(ttsx_vocoder* v, const float* y, int B, int F, float* spec_re, float* spec_im,
                                 void* stream) {
  TTSX_ARG(v && y && spec_re && spec_im, "ttsx_vocoder_stft: null argument");
  int rc = check_stft_params(&v->sp, B, F);
  if (rc || (rc = check_device(v, "ttsx_vocoder_stft"))) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const FrameGeom q = make_geom(&v->sp, F);
  if ((rc = grow(&v->ypad, &v->cap_ypad, (size_t)B * q.LP, st))) return rc;
  pad_signal_kernel<<<dim3((unsigned)((q.LP + 255) / 256), B), 256, 0, st>>>(v->ypad, y, B, F, q);
  TTSX_LAUNCH_CHECK();
  FrameP p{};
  p.ypad = v->ypad, p.spec_re = spec_re, p.spec_im = spec_im, p.tw = v->tw, p.winpad = v->winpad, p.B = B, p.F = F;
  p.q = q;
  frame_kernel<K_STFT><<<frame_grid((long)B * F), 64 * FR_PER_WG, 0, st>>>(p);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

extern "C" int ttsx_vocoder_istft(ttsx_vocoder* v, const float* spec_re, const float* spec_im, int B, int F, float* y,
                                  void* stream) {
  TTSX_ARG(v && y && spec_re && spec_im, "ttsx_vocoder_istft: null argument");
  int rc = check_stft_params(&v->sp, B, F);
  if (rc || (rc = check_device(v, "ttsx_vocoder_istft"))) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const FrameGeom q = make_geom(&v->sp, F);
  const size_t Lmax = (size_t)v->sp.hop_length * (F - 1);
  if ((rc = grow(&v->yw, &v->cap_yw, (size_t)B * F * WP, st))) return rc;
  FrameP p{};
  p.spec_re = const_cast<float*>(spec_re), p.spec_im = const_cast<float*>(spec_im), p.tw = v->tw, p.winpad = v->winpad;
  p.B = B, p.F = F, p.yw_out = v->yw, p.q = q;
  frame_kernel<K_ISTFT><<<frame_grid((long)B * F), 64 * FR_PER_WG, 0, st>>>(p);
  TTSX_LAUNCH_CHECK();
  const int n_slots = ceil_div(NFFT / 2 + (int)Lmax - q.lpad, q.hop);
  ola_kernel<0><<<dim3(ceil_div(n_slots, OLA_SLOTS), B), 320, 0, st>>>(y, v->yw, v->winpad, nullptr, B, F, q, n_slots);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

// ------------------------------------------------------------------ one-shot forms
// The same three operations without a handle: tables and scratch are allocated for the call and released before
// it returns (which waits for the device).  For occasional calls and unit tests; a serving loop keeps a handle.
namespace {
template <class F>
int with_temporary_vocoder(const ttsx_stft_params* sp, void* stream, F body) {
  ttsx_vocoder* v = nullptr;
  int rc = ttsx_vocoder_create(sp, stream, &v);
  if (rc) return rc;
  rc = body(v);
  if (!rc && hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) {
    ttsx_set_error("hipStreamSynchronize failed after a one-shot vocoder call");
    rc = TTSX_ERR_HIP;
  }
  (void)ttsx_vocoder_destroy(v);
  return rc;
}
}  // namespace

extern "C" int ttsx_griffin_lim(const float* mag, const float* init_phase, const int32_t* n_frames, int B, int F,
                                const ttsx_stft_params* sp, int iters, float preemph, float* wav, void* stream) {
  int rc = check_gl_args(mag, init_phase, wav, iters, preemph);
  if (rc || (rc = check_stft_params(sp, B, F))) return rc;
  return with_temporary_vocoder(sp, stream, [&](ttsx_vocoder* v) {
    return ttsx_vocoder_griffin_lim(v, mag, init_phase, n_frames, B, F, iters, preemph, wav, stream);
  });
}

extern "C" int ttsx_stft(const float* y, int B, int F, const ttsx_stft_params* sp, float* spec_re, float* spec_im,
                         void* stream) {
  TTSX_ARG(y && spec_re && spec_im, "ttsx_stft: null argument");
  int rc = check_stft_params(sp, B, F);
  if (rc) return rc;
  return with_temporary_vocoder(sp, stream,
                                [&](ttsx_vocoder* v) { return ttsx_vocoder_stft(v, y, B, F, spec_re, spec_im, stream); });
}

extern "C" int ttsx_istft(const float* spec_re, const float* spec_im, int B, int F, const ttsx_stft_params* sp,
                          float* y, void* stream) {
  TTSX_ARG(y && spec_re && spec_im, "ttsx_istft: null argument");
  int rc = check_stft_params(sp, B, F);
  if (rc) return rc;
  return with_temporary_vocoder(sp, stream,
                                [&](ttsx_vocoder* v) { return ttsx_vocoder_istft(v, spec_re, spec_im, B, F, y, stream); });
}

// ------------------------------------------------------------------ allocation-free form
// ttsx_vocoder_griffin_lim for callers that cannot allocate device memory (code under HIP graph capture, tight
// serving loops, harnesses that forbid it).  Tables and both padded signals live in one file-scope device array:
//   scratch = [ twiddles: 2048 floats | padded window: 2048 floats | signal 0 | signal 1 ]
// with room for NOALLOC_MAX_PADDED floats per signal (B * (2048 + hop * (F - 1)), rounded up to 4 per
// utterance): 4 Mi floats, e.g. 16 utterances of 900 frames.  The array costs 32 MiB + 16 KiB on every GPU
// the library runs on, for the life of the process.  Every call on a GPU uses that GPU's copy: calls that
// overlap on one GPU race.  Nothing has to start at zero: the tables are rebuilt by every call (a win_length may
// differ from the last caller's) and each launch writes every padded position of its utterances before the next
// one reads it.
constexpr size_t NOALLOC_TABLE_FLOATS = 2 * MC + NFFT;
constexpr size_t NOALLOC_MAX_PADDED = (size_t)1 << 22;
static __device__ __align__(256) float scratch[NOALLOC_TABLE_FLOATS + 2 * NOALLOC_MAX_PADDED];

namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
float* g_scratch_ptr[NOALLOC_MAX_DEVICES];  // device address of `scratch` on each GPU (the array's name is only a
std::mutex g_scratch_mu;                    // host-side stand-in for it), found once per device
}  // namespace

extern "C" int ttsx_griffin_lim_noalloc(const float* mag, const float* init_phase, const int32_t* n_frames, int B,
                                        int F, const ttsx_stft_params* sp, int iters, float preemph, float* wav,
                                        void* stream) {
  int rc = check_gl_args(mag, init_phase, wav, iters, preemph);
  if (rc || (rc = check_stft_params(sp, B, F))) return rc;
  const size_t n_pad = (size_t)B * make_geom(sp, F).LP;
  TTSX_ARG(n_pad <= NOALLOC_MAX_PADDED,
           "ttsx_griffin_lim_noalloc: B=%d x F=%d needs %zu floats per padded signal, the static scratch holds %zu "
           "(use ttsx_vocoder_griffin_lim for larger batches)", B, F, n_pad, NOALLOC_MAX_PADDED);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int dev = -1;
  TTSX_HIP(hipGetDevice(&dev));
  TTSX_ARG(dev >= 0 && dev < NOALLOC_MAX_DEVICES, "ttsx_griffin_lim_noalloc: device %d", dev);
  float* base = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
      float* ptr = nullptr;
      TTSX_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
      if ((rc = raise_deemph_lds())) return rc;
      g_scratch_ptr[dev] = ptr;
    }
    base = g_scratch_ptr[dev];
  }
  float2* tw = reinterpret_cast<float2*>(base);
  float* winpad = base + 2 * MC;
  init_tables_kernel<<<NFFT / 256, 256, 0, st>>>(tw, winpad, sp->win_length);
  TTSX_LAUNCH_CHECK();
  float* const ybuf[2] = {base + NOALLOC_TABLE_FLOATS, base + NOALLOC_TABLE_FLOATS + n_pad};
  return enqueue_griffin_lim(tw, winpad, ybuf, mag, init_phase, n_frames, B, F, sp, iters, preemph, wav, st);
}

#ifdef TTSX_GL_TIMING
extern "C" int ttsx_debug_gl_timing(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gl_marks), sizeof(long long) * 32) == hipSuccess ? 0 : 1;
}
#endif
