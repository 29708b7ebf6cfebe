// Tacotron2 greedy decoder for ONE utterance: one persistent launch per run, every decoder weight resident on chip.
//
// What a step computes follows Decoder.decode (Tacotron2/TTS/models.py:578-617) with Prenet (:76-82),
// OriginalAttention.forward (:314-361), LocationLayer (:84-104) and the greedy loop of Decoder.inference
// (:643-673), exactly as the launch chain in t2_decoder.hip evaluates it (composed stop and prenet-1 rows).
//
// The two LSTMs are 71 MB of fp32 weights; the chip has 128 MB of vector registers and 40 MB of LDS.  A TEAM of
// 216 workgroups (one per CU, 512 threads each) keeps all of it on chip for the whole utterance:
//   *  64 attention-LSTM workgroups, 16 hidden units each (a wave owns 2 units = 8 gate rows): the columns over
//      the prenet output and over h_a(t-1) in registers (160 per thread), the columns over the context in LDS;
//   * 128 decoder-LSTM workgroups, 8 hidden units each (a wave owns one unit): the columns over h_a in registers
//      (64 per thread), the columns over h_d(t-1) in LDS, and - instead of the columns over the context - their
//      products with the utterance's encoder outputs, one per position (see below);
//   *  16 projection workgroups: the mel rows, the composed stop-logit row, the composed prenet-layer-1 rows and
//      prenet layer 2, all in registers;
//   *   4 energy workgroups, one per 32 attention dimensions: their rows of the query layer in registers, the
//      processed inputs of the utterance in registers, location layers in LDS; they publish partial energies;
//   *   4 context workgroups, one per 128 encoder columns, with the utterance's encoder outputs in registers: they
//      add the partial energies, normalise and publish their slice of the context.
// Texts beyond 256 symbols (up to 1024) run with 16 + 16 attention workgroups (240 in all): the encoder axis is cut
// into four chunks of 256 positions, every energy / context workgroup owns one chunk of one slice, all of them
// normalise over all positions from the same partial energies, and the context travels as four partial vectors
// that its consumers add in a fixed order.
// Only activations cross the chip, as in the Tacotron (v1) team kernel (t1_team.hip): 64-bit words (step tag,
// fp32 value) in small exchange buffers, written with one contiguous store per workgroup and vector, polled once
// per workgroup (thread i polls word i) and handed on through LDS.  Five exchanges lie on a step's serial chain:
// x1 | x2 | h_a | partial energies | h_d.  Everything that depends on the previous step only (the
// recurrent columns of both LSTMs, the location features) is evaluated while the chain is elsewhere.
// The context is off that chain: it is a weighted sum of encoder outputs, ctx = sum_p a[p] enc[p], so what the
// decoder LSTM needs of it is W_c ctx = sum_p a[p] (W_c enc[p]).  The products G[row][p] = W_c[row] . enc[p] are
// formed once per utterance (ctx_products_kernel) and held in registers; a decoder-LSTM workgroup takes the partial
// energies themselves, normalises them like every attention workgroup does (same operations, same order) and
// adds sum_p a[p] G[row][p] - it never waits for the context workgroups, whose output goes to the projection and
// to the next step's attention LSTM, both of which need it later than the decoder LSTM did.
// A free run ends through the data: the workgroup that owns the stop row marks the words it publishes, and every
// workgroup that sees a marked word marks what it publishes and leaves.  Every wait is bounded: a team that is not
// resident together (other kernels hold CUs) raises an abort word after 50 ms of wall-clock time and the host reports it.
#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"
#include "ctx_products.h"
#include "t2_team.h"

namespace {

typedef unsigned long long u64;

constexpr int NT = 512;
constexpr int QD = 1024, DD = 1024, ED = 512, PD = 256, AD = 128, MEL = 80;
constexpr int LOCF = 32, LOCK = 31, LPAD = 15, TMAX = 256;  // (attention roles are built for 128 and for 256 positions)
constexpr int N_ATT = 64, N_DEC = 128, N_PROJ = 16, N_EN = 4;  // (and as many context workgroups as energy workgroups)
constexpr int G_DEC0 = N_ATT, G_PROJ0 = G_DEC0 + N_DEC, G_EN0 = G_PROJ0 + N_PROJ;
constexpr unsigned DONE_BIT = 0x80000000u;
constexpr int PROJ_K = DD + ED, PROJ_ROWS_MAX = 2 * MEL + 16 + PD;

// register images: slot s of thread tid of workgroup g of a role at img[(g * NS + s) * NT + tid]
constexpr int A_X2 = 0, A_HA = 32, A_B = 160, NS_ATT = 168;   // row ri = 4 * unit_in_wave + gate
constexpr int D_HA = 0, D_B = 64, NS_DEC = 68;   // row = gate
constexpr int P_ROW = 0, P_B = 80, P_W2 = 85, P_B2 = 93, NS_PROJ = 95;  // (rows over h_d only: sixteen slots each)
constexpr int NS_EN = 64;
constexpr int LDS_W = 32768;  // floats of weights in the LDS of an LSTM workgroup

// exchange buffers (u64 words); one block per step parity.  Vectors that one workgroup polls together are neighbours.
// The context travels as NC partial vectors (one per chunk of 256 encoder positions; NC = 1 up to 256 positions).
constexpr int NC_MAX = 4;
constexpr int X_HA = 0, X_CTX = X_HA + QD, X_HD = X_CTX + NC_MAX * ED, X_X1 = X_HD + DD, X_X1_N = 272 /* 256 + control */,
              X_X2 = X_X1 + X_X1_N, X_EP = X_X2 + PD, X_PAR = X_EP + N_EN * NC_MAX * 256;

struct T2P {
  const float *img_att, *lds_att, *img_dec, *lds_dec, *img_proj, *img_en;
  const float* ctx_gates;  // [1024 units][4 gates][TS positions]: decoder-LSTM context columns x encoder outputs
  const float* ctx_proj;   // [80 r + 272 rows][TS]: the projection rows' context columns x encoder outputs
  const float *conv_t /* [62][32] */, *loc_dense /* [128][32] */, *v_w, *v_b, *b1f;
  const float *enc, *P;
  const int32_t* len;
  u64* xch;
  int* abort_word;
  float *mel, *align, *stop;
  int32_t* n_steps;
  T2TeamState st;
  int T, r, S, free_run, norm;
  float thr;
};

#ifdef TTSX_TEAM_TIMING
// Tuning build: the first workgroup of every role leaves 100 MHz timestamps of step 50.
__device__ long long t2_team_marks[64];
#define T2_MARK(cond, i)                                                                \
  do {                                                                                  \
    if ((cond) && t == 50 && threadIdx.x == 0) t2_team_marks[i] = wall_clock64();        \
  } while (0)
#else
#define T2_MARK(cond, i) \
  do {                   \
  } while (0)
#endif

__device__ __forceinline__ void publish(u64* X, int idx, float v, unsigned tag) {
  __hip_atomic_store(X + idx, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Words tid + 512 j (j < N, index < n) of a span of published vectors into LDS, once all of them carry `tag`.
// Reads past n go to the span's last word and are ignored (a read inside a branch would wait for its own round
// trip).  Returns bit 0: aborted, bit 1: a word carried the end-of-run mark.
template <int N>
__device__ __forceinline__ int poll_span(const u64* X, int n, unsigned tag, float* dst, int* abort_word) {
  const int tid = threadIdx.x;
  long long t_wait = 0;
  for (long spin = 0;; ++spin) {
    u64 v[N];
    bool ok = true, dn = false;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int idx = tid + NT * j;
      v[j] = __hip_atomic_load(X + min(idx, n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned tf = (unsigned)(v[j] >> 32);
      ok = ok && (idx >= n || (tf & ~DONE_BIT) == tag);
      dn = dn || (idx < n && (tf & DONE_BIT) != 0);
    }
    if (__all(ok)) {
#pragma unroll
      for (int j = 0; j < N; ++j)
        if (tid + NT * j < n) dst[tid + NT * j] = __uint_as_float((unsigned)v[j]);
      return __any(dn) ? 2 : 0;
    }
    if ((spin & 127) == 127 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 1;
    if (team_wait_expired(spin, t_wait)) {
      __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return 1;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// poll_span by the whole workgroup, then a barrier: the words are in LDS for everyone.  Returns the union of the
// waves' status bits (sticky: a workgroup that has seen one leaves).
template <int N>
__device__ __forceinline__ int fetch_span(const u64* X, int n, unsigned tag, float* dst, int* abort_word, int* s_flag) {
  const int st = poll_span<N>(X, n, tag, dst, abort_word);
  if (st && (threadIdx.x & 63) == 0) atomicOr(s_flag, st);
  __syncthreads();
  return *s_flag;
}

// The same over two spans that are not neighbours in memory: words [0, na) of XA, then words [0, nb) of XB.
template <int N>
__device__ __forceinline__ int fetch_span2(const u64* XA, int na, const u64* XB, int nb, unsigned tag, float* dst,
                                           int* abort_word, int* s_flag) {
  const int tid = threadIdx.x, n = na + nb;
  int st = -1;
  long long t_wait = 0;
  for (long spin = 0; st < 0; ++spin) {
    u64 v[N];
    bool ok = true, dn = false;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int idx = min(tid + NT * j, n - 1);
      v[j] = __hip_atomic_load(idx < na ? XA + idx : XB + (idx - na), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned tf = (unsigned)(v[j] >> 32);
      ok = ok && (tid + NT * j >= n || (tf & ~DONE_BIT) == tag);
      dn = dn || (tid + NT * j < n && (tf & DONE_BIT) != 0);
    }
    if (__all(ok)) {
#pragma unroll
      for (int j = 0; j < N; ++j)
        if (tid + NT * j < n) dst[tid + NT * j] = __uint_as_float((unsigned)v[j]);
      st = __any(dn) ? 2 : 0;
    } else if ((spin & 127) == 127 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      st = 1;
    } else if (team_wait_expired(spin, t_wait)) {
      __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      st = 1;
    } else {
      __builtin_amdgcn_s_sleep(1);
    }
  }
  if (st && (threadIdx.x & 63) == 0) atomicOr(s_flag, st);
  __syncthreads();
  return *s_flag;
}

// Sums / maxima over the 64 lanes on the DPP network; every lane gets the result, in a fixed order.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wsum(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror
  v = dpp_add<0x142, 0xa>(v);  // row_bcast15
  v = dpp_add<0x143, 0xc>(v);  // row_bcast31: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max(float v) {
  const int i = __float_as_int(v);
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xf, false)));
}
__device__ __forceinline__ float wmax(float v) {
  v = dpp_max<0xB1, 0xf>(v);
  v = dpp_max<0x4E, 0xf>(v);
  v = dpp_max<0x141, 0xf>(v);
  v = dpp_max<0x140, 0xf>(v);
  v = dpp_max<0x142, 0xa>(v);
  v = dpp_max<0x143, 0xc>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// tanh through one hardware exponential, as the energies of the launch-chain decoder compute it
__device__ __forceinline__ float fast_tanhf(float x) {
  const float t = __expf(2.f * x);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}

// The logistic function through the hardware exponential and reciprocal (1 ulp each; the product x log2(e) inside
// __expf adds |x| 6e-8 relative to exp(-x), which the factor s (1 - s) of the derivative scales down again:
// absolute error below 2e-7 everywhere, against 6e-8 for the library's expf and an IEEE division - and four
// instructions on the step's serial chain instead of some twenty-five).
__device__ __forceinline__ float fast_sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// torch.nn.LSTMCell from the four gate pre-activations (i, f, g, o) of one hidden unit
__device__ __forceinline__ float lstm_cell(float gi, float gf, float gg, float go, float* c) {
  const float c_new = fast_sigmoidf(gf) * *c + fast_sigmoidf(gi) * fast_tanhf(gg);
  *c = c_new;
  return fast_sigmoidf(go) * fast_tanhf(c_new);
}

// Alignment weights of the utterance's own positions from the summed energies (threads < TM hold one position
// each): softmax or sigmoid normalisation (OriginalAttention.forward, models.py:343-353).  Called by all threads
// of the workgroup.  The energy and the context workgroups run exactly this on the same numbers.
template <int TM>
__device__ __forceinline__ float normalise(float e, bool valid, int norm, float* s_red) {
  constexpr int NW = TM / 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool softmax = norm == TTSX_NORM_SOFTMAX;
  float m = 0.f;
  if (softmax) {
    const float x = wmax(valid ? e : -INFINITY);
    if (lane == 0 && wave < NW) s_red[wave] = x;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) m = fmaxf(m, s_red[i]);
  }
  const float v = valid ? (softmax ? expf(e - m) : sigmoidf_(e)) : 0.f;
  const float part = wsum(v);
  if (lane == 0 && wave < NW) s_red[4 + wave] = part;
  __syncthreads();
  float sum = s_red[4];
#pragma unroll
  for (int i = 1; i < NW; ++i) sum += s_red[4 + i];
  return v / sum;
}

// Alignment weights of ALL NC * 256 positions from the partial energies in LDS ([4][NC * 256]): every energy and
// context workgroup of a long-text team runs exactly this, then uses its own chunk.  A thread holds positions
// tid + 512 j.  Called by all threads; the caller synchronises before s_al is read.
template <int NC>
__device__ __forceinline__ void normalise_all(const float* s_e4, float vb, int len, int norm, float* s_al, float* s_red) {
  constexpr int TS = NC * 256, NPT = TS / NT;
  static_assert(TS % NT == 0, "two chunks per 512 threads");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool softmax = norm == TTSX_NORM_SOFTMAX;
  float en[NPT];
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int i = tid + NT * j;
    en[j] = ((s_e4[i] + s_e4[TS + i]) + (s_e4[2 * TS + i] + s_e4[3 * TS + i])) + vb;
  }
  float m = 0.f;
  if (softmax) {
    float x = -INFINITY;
#pragma unroll
    for (int j = 0; j < NPT; ++j) x = fmaxf(x, tid + NT * j < len ? en[j] : -INFINITY);
    x = wmax(x);
    if (lane == 0) s_red[wave] = x;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) m = fmaxf(m, s_red[i]);
  }
  float v[NPT], part = 0.f;
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    v[j] = tid + NT * j < len ? (softmax ? expf(en[j] - m) : sigmoidf_(en[j])) : 0.f;
    part += v[j];
  }
  part = wsum(part);
  if (lane == 0) s_red[8 + wave] = part;
  __syncthreads();
  float sum = s_red[8];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) sum += s_red[8 + i];
#pragma unroll
  for (int j = 0; j < NPT; ++j) s_al[tid + NT * j] = v[j] / sum;
}

// Row of the composed projection matrix behind slot rs of wave `wave` of projection workgroup q (or -1): the composed
// prenet-1 rows 16 q + 2 w, + 1 (slots 0, 1), the mel rows q + 16 (w + 8 i) below 80 r (slots 2, 3), and - wave 7 of
// the last workgroup - the composed stop row (slot 4).
__host__ __device__ __forceinline__ int proj_src_row(int rs, int q, int wave, int n_mel) {
  if (rs < 2) return n_mel + 16 + 16 * q + 2 * wave + rs;
  if (rs < 4) {
    const int m = q + 16 * (wave + 8 * (rs - 2));
    return m < n_mel ? m : -1;
  }
  return q == N_PROJ - 1 && wave == 7 ? n_mel : -1;
}

// Alignment weights of positions lane + 64 j (j < TS / 64) formed by ONE wave from the partial energies in LDS
// ([4][TS]), with the operations and the order of summation of normalise<TS>: wave j of an attention workgroup adds
// positions 64 j ... 64 j + 63, and the wave sums are added in order.  For consumers of the alignment that are not
// attention workgroups (up to 256 positions: no barrier).
template <int TS>
__device__ __forceinline__ void wave_alignments(const float* s_e4, float vb, int len, bool softmax, float* a) {
  constexpr int NJ = TS / 64;
  const int lane = threadIdx.x & 63;
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = lane + 64 * j;
    a[j] = ((s_e4[i] + s_e4[TS + i]) + (s_e4[2 * TS + i] + s_e4[3 * TS + i])) + vb;
  }
  if (softmax) {
    float x = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j) x = fmaxf(x, lane + 64 * j < len ? a[j] : -INFINITY);
    m = wmax(x);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) a[j] = lane + 64 * j < len ? (softmax ? expf(a[j] - m) : sigmoidf_(a[j])) : 0.f;
  float sum = wsum(a[0]);
#pragma unroll
  for (int j = 1; j < NJ; ++j) sum += wsum(a[j]);
#pragma unroll
  for (int j = 0; j < NJ; ++j) a[j] = a[j] / sum;
}

// ------------------------------------------------------------------ attention LSTM: 16 hidden units
template <int NC>
__device__ void att_role(const T2P& p, int a, float* sm) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* s_wc = sm;                 // [wave][8 rows][8][64] columns over the context
  float* s_x2 = s_wc + LDS_W;       // [256]
  float* s_hc = s_x2 + PD;          // [1024] h_a(t-1), [NC][512] partial contexts of step t-1
  float* s_g = s_hc + QD + NC * ED;  // [16 units][4] gate pre-activations
  int* s_flag = reinterpret_cast<int*>(s_g + 64);
  float* s_w15 = s_g + 64 + 8;        // [8 rows][512]: the last of a row's sixteen slots over h_a(t-1) (off the serial
                                    // chain; eight registers the allocator would otherwise keep in scratch)
  float w[NS_ATT];
  {
    const float* img = p.img_att + ((size_t)a * NS_ATT) * NT + tid;
#pragma unroll
    for (int s = 0; s < NS_ATT; ++s) {
      if (s >= A_HA && s < A_B && (s - A_HA) % 16 == 15) s_w15[(s - A_HA) / 16 * NT + tid] = img[(size_t)s * NT], w[s] = 0.f;
      else w[s] = img[(size_t)s * NT];
    }
    // (the biases are the same on every lane: scalar registers)
#pragma unroll
    for (int s = A_B; s < NS_ATT; ++s) w[s] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w[s])));
  }
  for (int i = tid; i < LDS_W; i += NT) s_wc[i] = p.lds_att[(size_t)a * LDS_W + i];
  if (tid == 0) *s_flag = 0;
  __syncthreads();
  const float* wc = s_wc + (size_t)wave * 8 * 8 * 64 + lane;
  float cc = 0.f, hh = 0.f;  // lanes 0-15 of the first wave: the workgroup's sixteen units
  for (int t = 0; t < p.S; ++t) {
    const unsigned tag = (unsigned)t + 1u;
    u64* Xc = p.xch + (size_t)(t & 1) * X_PAR;
    const u64* Xp = p.xch + (size_t)((t & 1) ^ 1) * X_PAR;
    // ---- the columns over h_a(t-1) and ctx(t-1): off the serial chain
    float acc[8];
#pragma unroll
    for (int ri = 0; ri < 8; ++ri) acc[ri] = 0.f;
    if (t > 0) {
      if constexpr (NC == 1) {
        if (fetch_span<(QD + ED) / NT>(Xp + X_HA, QD + ED, tag - 1, s_hc, p.abort_word, s_flag) & 1) return;
      } else {
        // (two polls: six words per thread in flight next to the weight registers is what the allocator spills)
        static_assert(X_CTX == X_HA + QD, "h_a and the partial contexts are neighbours");
        if (fetch_span<QD / NT>(Xp + X_HA, QD, tag - 1, s_hc, p.abort_word, s_flag) & 1) return;
        if (fetch_span<NC * ED / NT>(Xp + X_CTX, NC * ED, tag - 1, s_hc + QD, p.abort_word, s_flag) & 1) return;
      }
      if constexpr (NC > 1) {
        // the partial contexts added up front, one column per thread (left to the loop below, the 32 operands in
        // flight pushed weight registers into scratch, and their reloads sat on the serial chain: 0.8 us)
        static_assert(NT == ED, "one context column per thread");
        float cs = s_hc[QD + tid];
#pragma unroll
        for (int c = 1; c < NC; ++c) cs += s_hc[QD + c * ED + tid];
        s_hc[QD + tid] = cs;
        __syncthreads();
      }
      float hv[16], cv[8];
#pragma unroll
      for (int j = 0; j < 16; ++j) hv[j] = s_hc[lane + 64 * j];
#pragma unroll
      for (int j = 0; j < 8; ++j) cv[j] = s_hc[QD + lane + 64 * j];
#pragma unroll
      for (int ri = 0; ri < 8; ++ri) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 15; ++j) s += w[A_HA + ri * 16 + j] * hv[j];
        s += s_w15[ri * NT + tid] * hv[15];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += wc[(ri * 8 + j) * 64] * cv[j];
        acc[ri] = s;
      }
    }
    // ---- the columns over the prenet output, the cell update
    const int st = fetch_span<1>(Xc + X_X2, PD, tag, s_x2, p.abort_word, s_flag);
    T2_MARK(a == 0, 8);
    if (st & 1) return;
    const bool done = (st & 2) != 0;
    if (!done) {
      float xv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = s_x2[lane + 64 * j];
      float g[8];
#pragma unroll
      for (int ri = 0; ri < 8; ++ri) {
        float s = acc[ri];
#pragma unroll
        for (int j = 0; j < 4; ++j) s += w[A_X2 + ri * 4 + j] * xv[j];
        g[ri] = wsum(s) + w[A_B + ri];
      }
      if (lane == 0) {
        reinterpret_cast<float4*>(s_g)[2 * wave] = make_float4(g[0], g[1], g[2], g[3]);
        reinterpret_cast<float4*>(s_g)[2 * wave + 1] = make_float4(g[4], g[5], g[6], g[7]);
      }
      T2_MARK(a == 0, 10);
    }
    __syncthreads();
    T2_MARK(a == 0, 12);
    // The cell updates of all sixteen units side by side on sixteen lanes of one wave: as one instruction stream
    // instead of sixteen (two per wave, two waves per SIMD taking turns on the transcendental unit: 0.4 us for
    // the first wave and 0.6 more until the last one reached the barrier).
    if (wave == 0 && lane < 16) {
      if (!done) {
        const float4 gv = reinterpret_cast<const float4*>(s_g)[lane];
        hh = lstm_cell(gv.x, gv.y, gv.z, gv.w, &cc);
      }
      publish(Xc + X_HA, 16 * a + lane, done ? 0.f : hh, done ? tag | DONE_BIT : tag);
    }
    T2_MARK(a == 0, 9);
    if (done) break;
  }
  if (wave == 0 && lane < 16) {
    const int u = 16 * a + lane;
    if (p.st.h_a) p.st.h_a[u] = hh;
    if (p.st.h_a_alt) p.st.h_a_alt[u] = hh;
    if (p.st.c_a) p.st.c_a[u] = cc;
  }
}

// ------------------------------------------------------------------ decoder LSTM: 8 hidden units
// TS = positions the team's attention roles are built for (128, 256 or NC * 256).  Lane l of every wave holds
// positions l + 64 j.  Up to 256 positions every wave forms the alignment weights by itself (no barrier), with
// the operations and the order of summation of normalise<TM>; long texts share the work over the workgroup
// through normalise_all<NC> (sixteen sigmoids per lane and wave would cost more than its three barriers: 4.3 us
// from the partial energies to h_d against 2.0).  Either way they are the numbers the attention workgroups get.
template <int TM, int NC>
__device__ void dec_role(const T2P& p, int d, float* sm) {
  constexpr int TS = NC == 1 ? TM : NC * 256, NJ = TS / 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* s_wh = sm;                 // [wave][4 rows][16][64] columns over h_d(t-1)
  float* s_hd = s_wh + LDS_W;       // [1024]
  float* s_ha = s_hd + DD;          // [1024]
  float* s_e4 = s_ha + QD;          // [4][TS] partial energies
  float* s_al = s_e4 + N_EN * TS;   // [TS] alignment weights (long texts)
  float* s_red = s_al + (NC == 1 ? 0 : TS);  // [16]
  float* s_g = s_red + 16;          // [8 units][4] gate pre-activations
  int* s_flag = reinterpret_cast<int*>(s_g + 32);
  const int len = min(max(p.len[0], 0), p.T);
  float w[NS_DEC], wm[4 * NJ];
  {
    const float* img = p.img_dec + ((size_t)d * NS_DEC) * NT + tid;
#pragma unroll
    for (int s = 0; s < NS_DEC; ++s) w[s] = img[(size_t)s * NT];
#pragma unroll
    for (int s = D_B; s < NS_DEC; ++s) w[s] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w[s])));
    const float* G = p.ctx_gates + (size_t)(8 * d + wave) * 4 * TS + lane;
#pragma unroll
    for (int s = 0; s < 4 * NJ; ++s) wm[s] = G[(size_t)(s / NJ) * TS + 64 * (s % NJ)];
  }
  for (int i = tid; i < LDS_W; i += NT) s_wh[i] = p.lds_dec[(size_t)d * LDS_W + i];
  if (tid == 0) *s_flag = 0;
  const float vb = p.v_b[0];
  const bool softmax = p.norm == TTSX_NORM_SOFTMAX;
  __syncthreads();
  const float* wh = s_wh + (size_t)wave * 4 * 16 * 64 + lane;
  float cc = 0.f, hh = 0.f;  // lanes 0-7 of the first wave: the workgroup's eight units
  for (int t = 0; t < p.S; ++t) {
    const unsigned tag = (unsigned)t + 1u;
    u64* Xc = p.xch + (size_t)(t & 1) * X_PAR;
    const u64* Xp = p.xch + (size_t)((t & 1) ^ 1) * X_PAR;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // ---- the columns over h_d(t-1)
    if (t > 0) {
      if (fetch_span<2>(Xp + X_HD, DD, tag - 1, s_hd, p.abort_word, s_flag) & 1) return;
      float hv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) hv[j] = s_hd[lane + 64 * j];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += wh[(g * 16 + j) * 64] * hv[j];
        acc[g] = s;
      }
    }
    // ---- the columns over h_a(t): while the attention runs
    {
      const int st = fetch_span<2>(Xc + X_HA, QD, tag, s_ha, p.abort_word, s_flag);
    T2_MARK(d == 0, 32);
      if (st & 1) return;
      if (st & 2) break;
      float hv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) hv[j] = s_ha[lane + 64 * j];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float s = acc[g];
#pragma unroll
        for (int j = 0; j < 16; ++j) s += w[D_HA + g * 16 + j] * hv[j];
        acc[g] = s;
      }
    }
    // ---- the context's share from the partial energies, the cell update
    {
      const int st = fetch_span<N_EN * TS / NT>(Xc + X_EP, N_EN * TS, tag, s_e4, p.abort_word, s_flag);
    T2_MARK(d == 0, 33);
      if (st & 1) return;
      if (st & 2) break;
    }
    {
      float a[NJ];
      if constexpr (NC == 1) {
        wave_alignments<TS>(s_e4, vb, len, softmax, a);
      } else {
        normalise_all<NC>(s_e4, vb, len, p.norm, s_al, s_red);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) a[j] = s_al[lane + 64 * j];
      }
      float g4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float s = acc[g];
#pragma unroll
        for (int j = 0; j < NJ; ++j) s += wm[g * NJ + j] * a[j];
        g4[g] = wsum(s) + w[D_B + g];
      }
      if (lane == 0) reinterpret_cast<float4*>(s_g)[wave] = make_float4(g4[0], g4[1], g4[2], g4[3]);
    }
    __syncthreads();
    if (wave == 0 && lane < 8) {  // (the eight cell updates as one instruction stream, as in the attention LSTM)
      const float4 gv = reinterpret_cast<const float4*>(s_g)[lane];
      hh = lstm_cell(gv.x, gv.y, gv.z, gv.w, &cc);
      publish(Xc + X_HD, 8 * d + lane, hh, tag);
    }
    T2_MARK(d == 0, 34);
  }
  if (wave == 0 && lane < 8) {
    const int u = 8 * d + lane;
    if (p.st.h_d) p.st.h_d[u] = hh;
    if (p.st.h_d_alt) p.st.h_d_alt[u] = hh;
    if (p.st.c_d) p.st.c_d[u] = cc;
  }
}

// ------------------------------------------------------------------ projection + prenet layer 2
// Wave w of workgroup q owns: the composed prenet-1 rows 16 q + 2 w, + 1 (slots 0, 1), the mel rows
// q + 16 (w + 8 i), i = 0, 1 (slots 2, 3; below 80 r), and - wave 7 of workgroup 15 - the composed stop row (slot 4);
// prenet-2 rows 16 q + 2 w, + 1.
// The rows' columns over the context are held as their products with the utterance's encoder outputs, like the
// decoder LSTM's (ctx_proj): the workgroup forms the alignment weights of step t - 1 from the partial energies while
// the decoder LSTM is still at work, and what it waits for on the chain is h_d alone.
template <int TM, int NC>
__device__ void proj_role(const T2P& p, int q, float* sm) {
  constexpr int TS = NC == 1 ? TM : NC * 256, NJ = TS / 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* s_hd = sm;                 // [1024] h_d; after the run: [NC][512] the last context
  float* s_e4 = s_hd + (NC * ED > DD ? NC * ED : DD);  // [4][TS] partial energies
  float* s_al = s_e4 + N_EN * TS;   // [TS] alignment weights (long texts)
  float* s_red = s_al + (NC == 1 ? 0 : TS);  // [16]
  float* s_x1 = s_red + 16;         // [257]
  float* s_pub = s_x1 + X_X1_N;     // [16] x1 slice, [16] x2 slice
  int* s_flag = reinterpret_cast<int*>(s_pub + 32);
  const bool owner = q == N_PROJ - 1 && wave == 7;  // of the stop row
  const int n_mel = MEL * p.r;
  const int len = min(max(p.len[0], 0), p.T);
  float w[NS_PROJ], wg[5 * NJ];
  {
    const float* img = p.img_proj + ((size_t)q * NS_PROJ) * NT + tid;
#pragma unroll
    for (int s = 0; s < NS_PROJ; ++s) w[s] = img[(size_t)s * NT];
#pragma unroll
    for (int s = P_B; s < NS_PROJ; ++s)
      if (s < P_W2 || s >= P_B2) w[s] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w[s])));
#pragma unroll
    for (int rs = 0; rs < 5; ++rs) {
      const int row = proj_src_row(rs, q, wave, n_mel);
#pragma unroll
      for (int j = 0; j < NJ; ++j) wg[rs * NJ + j] = row >= 0 ? p.ctx_proj[(size_t)row * TS + lane + 64 * j] : 0.f;
    }
  }
  if (tid == 0) *s_flag = 0;
  const float vb = p.v_b[0];
  const bool softmax = p.norm == TTSX_NORM_SOFTMAX;
  __syncthreads();
  int n_stop = 0;  // 1-based step of the first stop token above the threshold (owner wave)
  int t = 0;
  for (;; ++t) {
    const unsigned tag = (unsigned)t + 1u;
    u64* Xc = p.xch + (size_t)(t & 1) * X_PAR;
    const u64* Xp = p.xch + (size_t)((t & 1) ^ 1) * X_PAR;
    float xin[16], cs[5];
    if (t > 0) {
      // ---- the projection of step t - 1.  Off the chain: the context's share of every row from the partial
      // energies (no end-of-run mark can arrive here: the marks start from this role)
      if (fetch_span<N_EN * TS / NT>(Xp + X_EP, N_EN * TS, tag - 1, s_e4, p.abort_word, s_flag) & 1) return;
      {
        float a[NJ];
        if constexpr (NC == 1) {
          wave_alignments<TS>(s_e4, vb, len, softmax, a);
        } else {
          normalise_all<NC>(s_e4, vb, len, p.norm, s_al, s_red);
          __syncthreads();
#pragma unroll
          for (int j = 0; j < NJ; ++j) a[j] = s_al[lane + 64 * j];
        }
#pragma unroll
        for (int rs = 0; rs < 5; ++rs) {
          float c = 0.f;
#pragma unroll
          for (int j = 0; j < NJ; ++j) c += wg[rs * NJ + j] * a[j];
          cs[rs] = c;
          asm volatile("" : "+v"(cs[rs])::"memory");  // formed here, not behind the next wait
        }
      }
      // ---- on the chain: h_d, then prenet layer 1 of the fed-back frame (and, in its owner's wave, the stop token)
      if (fetch_span<DD / NT>(Xp + X_HD, DD, tag - 1, s_hd, p.abort_word, s_flag) & 1) return;
    T2_MARK(q == 0, 0);
#pragma unroll
      for (int j = 0; j < 16; ++j) xin[j] = s_hd[lane + 64 * j];
      float val[2];
#pragma unroll
      for (int rs = 0; rs < 2; ++rs) {
        float s = cs[rs];
#pragma unroll
        for (int j = 0; j < 16; ++j) s += w[P_ROW + rs * 16 + j] * xin[j];
        val[rs] = wsum(s) + w[P_B + rs];
      }
      if (lane == 0) s_pub[2 * wave] = fmaxf(val[0], 0.f), s_pub[2 * wave + 1] = fmaxf(val[1], 0.f);
      if (owner) {
        float s = cs[4];
#pragma unroll
        for (int j = 0; j < 16; ++j) s += w[P_ROW + 4 * 16 + j] * xin[j];
        s = sigmoidf_(wsum(s) + w[P_B + 4]);
        if (lane == 0) p.stop[t - 1] = s;
        if (s > p.thr && n_stop == 0) n_stop = t;
      }
    } else if (lane == 0) {
      // the first step is fed an all-zero frame: layer 1 gives relu(bias)
      s_pub[2 * wave] = fmaxf(p.b1f[16 * q + 2 * wave], 0.f), s_pub[2 * wave + 1] = fmaxf(p.b1f[16 * q + 2 * wave + 1], 0.f);
    }
    const bool fin = t == p.S || (p.free_run && n_stop != 0);  // (the second term is the owner wave's knowledge)
    __syncthreads();
    if (t < p.S) {
      if (wave == 0 && lane < 16) publish(Xc + X_X1, 16 * q + lane, s_pub[lane], tag);
      if (owner && lane == 0) publish(Xc + X_X1, PD, 0.f, fin ? tag | DONE_BIT : tag);
    }
    T2_MARK(q == 0, 1);
    if (t > 0) {
      // ---- then, while x1 travels, the frame rows of step t - 1 (the inputs pass through an empty statement so
      // that these sums are not formed ahead of the publication)
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(xin[j]));
      const size_t row0 = (size_t)(t - 1) * p.r * MEL;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float s = cs[2 + i];
#pragma unroll
        for (int j = 0; j < 16; ++j) s += w[P_ROW + (2 + i) * 16 + j] * xin[j];
        s = wsum(s) + w[P_B + 2 + i];
        const int m = q + 16 * (wave + 8 * i);
        if (lane == 0 && m < n_mel) p.mel[row0 + m] = s;
      }
    }
    if (t == p.S) break;  // every workgroup knows the step limit
    // ---- prenet layer 2
    const int st = fetch_span<1>(Xc + X_X1, PD + 1, tag, s_x1, p.abort_word, s_flag);
    T2_MARK(q == 0, 2);
    if (st & 1) return;
    const bool done = (st & 2) != 0;
    if (!done) {
      float xv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = s_x1[lane + 64 * j];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += w[P_W2 + i * 4 + j] * xv[j];
        const float x2 = fmaxf(wsum(s) + w[P_B2 + i], 0.f);
        if (lane == 0) s_pub[16 + 2 * wave + i] = x2;
      }
    }
    __syncthreads();
    if (wave == 0 && lane < 16) publish(Xc + X_X2, 16 * q + lane, done ? 0.f : s_pub[16 + lane], done ? tag | DONE_BIT : tag);
    T2_MARK(q == 0, 3);
    if (done) break;
  }
  // what the next step would have been fed, the step count, and the context of the last step (t >= 1 here; the
  // context workgroups published it with the tag of that step, and nothing has overwritten it since)
  if (p.st.x1 && wave == 0 && lane < 16) p.st.x1[16 * q + lane] = s_pub[lane];
  if (owner && lane == 0) p.n_steps[0] = n_stop;
  if (p.st.ctx && q == 0) {
    const u64* Xl = p.xch + (size_t)((t - 1) & 1) * X_PAR;
    if (fetch_span<NC * ED / NT>(Xl + X_CTX, NC * ED, (unsigned)t, s_hd, p.abort_word, s_flag) & 1) return;
    if (tid < ED) {
      float c = s_hd[tid];
#pragma unroll
      for (int k = 1; k < NC; ++k) c += s_hd[k * ED + tid];
      p.st.ctx[tid] = c;
    }
  }
}

// ------------------------------------------------------------------ energies: 32 attention dimensions
// TM = positions the role is built for (128 or 256): a thread owns one position and 32 * TM / 512 of the dims.
template <int TM>
__device__ void energy_role(const T2P& p, int e, float* sm) {
  constexpr int NG = NT / TM, ND = 32 / NG;  // position groups of threads; own dims per thread
  constexpr int CP = TM / 16;                // positions per thread in the location conv
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.T, len = min(max(p.len[0], 0), T);
  float* s_wl = sm;                              // [62][32] location conv, filter fastest
  float* s_wld = s_wl + 2 * LOCK * LOCF;         // [32 own dims][32]
  float* s_v = s_wld + 32 * LOCF;                // [32]
  float* s_a = s_v + 32;                         // [2][TM + 32] previous / cumulative alignment, 15 zeros in front
  float* s_cf = s_a + 2 * (TM + 32);             // [TM][33]
  float* s_ha = s_cf + TM * (LOCF + 1);          // [1024]
  float* s_q = s_ha + QD;                        // [32]
  float* s_ep = s_q + 32;                        // [NG][TM]
  float* s_e4 = s_ep + NG * TM;                  // [4][TM] partial energies of the four workgroups
  float* s_red = s_e4 + N_EN * TM;               // [8]
  int* s_flag = reinterpret_cast<int*>(s_red + 8);
  float wq[NS_EN];
  {
    const float* img = p.img_en + ((size_t)e * NS_EN) * NT + tid;
#pragma unroll
    for (int s = 0; s < NS_EN; ++s) wq[s] = img[(size_t)s * NT];
  }
  for (int i = tid; i < 2 * LOCK * LOCF; i += NT) s_wl[i] = p.conv_t[i];
  for (int i = tid; i < 32 * LOCF; i += NT) s_wld[i] = p.loc_dense[(size_t)32 * e * LOCF + i];
  if (tid < 32) s_v[tid] = p.v_w[32 * e + tid];
  for (int i = tid; i < 2 * (TM + 32); i += NT) s_a[i] = 0.f;
  if (tid == 0) *s_flag = 0;
  const int pos = tid & (TM - 1), dq = tid / TM;
  float pp[ND], pl[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) pp[i] = pos < len ? p.P[(size_t)pos * AD + 32 * e + ND * dq + i] : 0.f;
  const float vb = p.v_b[0];
  __syncthreads();
  // location conv, thread (filter f, CP positions), then the location-dense rows of the own dims:
  // pl[i] = P[pos][d] + (W_ld . conv features of pos)[d].  Needs the previous alignments only.
  auto location_terms = [&]() {
    {
      const int f = tid & 31, tp = tid >> 5;
      float acc[CP];
#pragma unroll
      for (int i = 0; i < CP; ++i) acc[i] = 0.f;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const float* a = s_a + ch * (TM + 32) + CP * tp;
        float win[LOCK + CP - 1];
#pragma unroll
        for (int j = 0; j < LOCK + CP - 1; ++j) win[j] = a[j];
#pragma unroll
        for (int j = 0; j < LOCK; ++j) {
          const float wv = s_wl[(ch * LOCK + j) * LOCF + f];
#pragma unroll
          for (int i = 0; i < CP; ++i) acc[i] += wv * win[j + i];
        }
      }
#pragma unroll
      for (int i = 0; i < CP; ++i) s_cf[(CP * tp + i) * (LOCF + 1) + f] = acc[i];
    }
    __syncthreads();
    float cf[LOCF];
#pragma unroll
    for (int f = 0; f < LOCF; ++f) cf[f] = s_cf[pos * (LOCF + 1) + f];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const float4* wl = reinterpret_cast<const float4*>(s_wld + (ND * dq + i) * LOCF);
      float loc = 0.f;
#pragma unroll
      for (int f4 = 0; f4 < LOCF / 4; ++f4) {
        const float4 w4 = wl[f4];
        loc += w4.x * cf[4 * f4] + w4.y * cf[4 * f4 + 1] + w4.z * cf[4 * f4 + 2] + w4.w * cf[4 * f4 + 3];
      }
      pl[i] = pp[i] + loc;
      asm volatile("" : "+v"(pl[i])::"memory");  // evaluated here, not behind the next wait
    }
  };
  location_terms();
  for (int t = 0; t < p.S; ++t) {
    const unsigned tag = (unsigned)t + 1u;
    u64* Xc = p.xch + (size_t)(t & 1) * X_PAR;
    // ---- own rows of the query layer on h_a(t)
    const int st = fetch_span<2>(Xc + X_HA, QD, tag, s_ha, p.abort_word, s_flag);
    T2_MARK(e == 0, 16);
    if (st & 1) return;
    if (st & 2) {
      if (tid < TM) publish(Xc + X_EP, e * TM + tid, 0.f, tag | DONE_BIT);
      return;
    }
    {
      float hv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) hv[j] = s_ha[lane + 64 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += wq[i * 16 + j] * hv[j];
        const float qv = wsum(s);
        if (lane == 0) s_q[4 * wave + i] = qv;
      }
    }
    __syncthreads();
    // ---- partial energies over the own dims
    {
      const float4* q4 = reinterpret_cast<const float4*>(s_q + ND * dq);
      const float4* v4 = reinterpret_cast<const float4*>(s_v + ND * dq);
      float e0 = 0.f, e1 = 0.f;
#pragma unroll
      for (int i = 0; i < ND / 4; ++i) {
        const float4 qv = q4[i], v = v4[i];
        e0 += v.x * fast_tanhf(qv.x + pl[4 * i]) + v.z * fast_tanhf(qv.z + pl[4 * i + 2]);
        e1 += v.y * fast_tanhf(qv.y + pl[4 * i + 1]) + v.w * fast_tanhf(qv.w + pl[4 * i + 3]);
      }
      s_ep[dq * TM + pos] = e0 + e1;
    }
    __syncthreads();
    if (tid < TM) {
      float ep = s_ep[tid];
#pragma unroll
      for (int g = 1; g < NG; ++g) ep += s_ep[g * TM + tid];
      publish(Xc + X_EP, e * TM + tid, ep, tag);
    }
    T2_MARK(e == 0, 17);
    // ---- off the serial chain: this step's alignment (state of the location layer; output row)
    if (fetch_span<N_EN * TM / NT>(Xc + X_EP, N_EN * TM, tag, s_e4, p.abort_word, s_flag) & 1) return;
    T2_MARK(e == 0, 18);
    {
      const int i = tid & (TM - 1);
      const float en = ((s_e4[i] + s_e4[TM + i]) + (s_e4[2 * TM + i] + s_e4[3 * TM + i])) + vb;
      const float a = normalise<TM>(en, tid < len, p.norm, s_red);
      if (tid < TM) {
        s_a[LPAD + tid] = a;
        s_a[(TM + 32) + LPAD + tid] += a;
        if (e == 0 && tid < T) p.align[(size_t)t * T + tid] = a;
      }
    }
    __syncthreads();
    location_terms();
    T2_MARK(e == 0, 19);
  }
}

// ------------------------------------------------------------------ context: 128 encoder columns
template <int TM>
__device__ void ctx_role(const T2P& p, int c, float* sm) {
  constexpr int NP = TM / 4;  // positions per thread
  const int tid = threadIdx.x;
  const int T = p.T, len = min(max(p.len[0], 0), T);
  float* s_e4 = sm;                    // [4][TM]
  float* s_al = s_e4 + N_EN * TM;      // [TM]
  float* s_cp = s_al + TM;             // [4][128]
  float* s_red = s_cp + 4 * 128;       // [8]
  int* s_flag = reinterpret_cast<int*>(s_red + 8);
  const int col = tid & 127, tq = tid >> 7;
  float er[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int ps = NP * tq + i;
    er[i] = ps < len ? p.enc[(size_t)ps * ED + 128 * c + col] : 0.f;
  }
  const float vb = p.v_b[0];
  if (tid == 0) *s_flag = 0;
  __syncthreads();
  for (int t = 0; t < p.S; ++t) {
    const unsigned tag = (unsigned)t + 1u;
    u64* Xc = p.xch + (size_t)(t & 1) * X_PAR;
    const int st = fetch_span<N_EN * TM / NT>(Xc + X_EP, N_EN * TM, tag, s_e4, p.abort_word, s_flag);
    T2_MARK(c == 0, 24);
    if (st & 1) return;
    if (st & 2) break;
    {
      const int i = tid & (TM - 1);
      const float en = ((s_e4[i] + s_e4[TM + i]) + (s_e4[2 * TM + i] + s_e4[3 * TM + i])) + vb;
      const float a = normalise<TM>(en, tid < len, p.norm, s_red);
      if (tid < TM) s_al[tid] = a;
    }
    __syncthreads();
    {
      const float4* a4 = reinterpret_cast<const float4*>(s_al + NP * tq);
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int i = 0; i < NP / 4; ++i) {
        const float4 av = a4[i];
        a0 += av.x * er[4 * i] + av.z * er[4 * i + 2];
        a1 += av.y * er[4 * i + 1] + av.w * er[4 * i + 3];
      }
      s_cp[tq * 128 + col] = a0 + a1;
    }
    __syncthreads();
    if (tid < 128) publish(Xc + X_CTX, 128 * c + tid, (s_cp[tid] + s_cp[128 + tid]) + (s_cp[256 + tid] + s_cp[384 + tid]), tag);
    T2_MARK(c == 0, 25);
  }
}

// ------------------------------------------------------------------ long texts: NC chunks of 256 positions
// Energies of one chunk of 256 positions over 32 attention dimensions: workgroup ge = 4 * chunk + slice.
template <int NC>
__device__ void energy_long_role(const T2P& p, int ge, float* sm) {
  constexpr int TS = NC * 256, NPT = TS / NT, ND = 16, CP = 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int e = ge & 3, base = 256 * (ge >> 2);
  const int T = p.T, len = min(max(p.len[0], 0), T);
  float* s_wl = sm;                              // [62][32] location conv, filter fastest
  float* s_wld = s_wl + 2 * LOCK * LOCF;         // [32 own dims][32]
  float* s_v = s_wld + 32 * LOCF;                // [32]
  float* s_prev = s_v + 32;                      // [TS + 32] previous alignment of ALL positions, 15 zeros in front
  float* s_cum = s_prev + TS + 32;               // [TS + 32] cumulative alignment
  float* s_cf = s_cum + TS + 32;                 // [256][33] conv features of the own chunk
  float* s_ha = s_cf + 256 * (LOCF + 1);         // [1024]
  float* s_q = s_ha + QD;                        // [32]
  float* s_ep = s_q + 32;                        // [2][256]
  float* s_e4 = s_ep + 2 * 256;                  // [4][TS] partial energies of all positions
  float* s_al = s_e4 + 4 * TS;                   // [TS]
  float* s_red = s_al + TS;                      // [16]
  int* s_flag = reinterpret_cast<int*>(s_red + 16);
  float wq[NS_EN];
  {
    const float* img = p.img_en + ((size_t)e * NS_EN) * NT + tid;
#pragma unroll
    for (int s = 0; s < NS_EN; ++s) wq[s] = img[(size_t)s * NT];
  }
  for (int i = tid; i < 2 * LOCK * LOCF; i += NT) s_wl[i] = p.conv_t[i];
  for (int i = tid; i < 32 * LOCF; i += NT) s_wld[i] = p.loc_dense[(size_t)32 * e * LOCF + i];
  if (tid < 32) s_v[tid] = p.v_w[32 * e + tid];
  for (int i = tid; i < 2 * (TS + 32); i += NT) s_prev[i] = 0.f;  // (s_prev and s_cum are neighbours)
  if (tid == 0) *s_flag = 0;
  const int pos = tid & 255, dq = tid >> 8, gpos = base + pos;
  float pp[ND], pl[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) pp[i] = gpos < len ? p.P[(size_t)gpos * AD + 32 * e + ND * dq + i] : 0.f;
  const float vb = p.v_b[0];
  __syncthreads();
  auto location_terms = [&]() {
    {
      const int f = tid & 31, tp = tid >> 5;
      float acc[CP];
#pragma unroll
      for (int i = 0; i < CP; ++i) acc[i] = 0.f;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const float* a = (ch ? s_cum : s_prev) + base + CP * tp;
        float win[LOCK + CP - 1];
#pragma unroll
        for (int j = 0; j < LOCK + CP - 1; ++j) win[j] = a[j];
#pragma unroll
        for (int j = 0; j < LOCK; ++j) {
          const float wv = s_wl[(ch * LOCK + j) * LOCF + f];
#pragma unroll
          for (int i = 0; i < CP; ++i) acc[i] += wv * win[j + i];
        }
      }
#pragma unroll
      for (int i = 0; i < CP; ++i) s_cf[(CP * tp + i) * (LOCF + 1) + f] = acc[i];
    }
    __syncthreads();
    float cf[LOCF];
#pragma unroll
    for (int f = 0; f < LOCF; ++f) cf[f] = s_cf[pos * (LOCF + 1) + f];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const float4* wl = reinterpret_cast<const float4*>(s_wld + (ND * dq + i) * LOCF);
      float loc = 0.f;
#pragma unroll
      for (int f4 = 0; f4 < LOCF / 4; ++f4) {
        const float4 w4 = wl[f4];
        loc += w4.x * cf[4 * f4] + w4.y * cf[4 * f4 + 1] + w4.z * cf[4 * f4 + 2] + w4.w * cf[4 * f4 + 3];
      }
      pl[i] = pp[i] + loc;
      asm volatile("" : "+v"(pl[i])::"memory");
    }
  };
  location_terms();
  for (int t = 0; t < p.S; ++t) {
    const unsigned tag = (unsigned)t + 1u;
    u64* Xc = p.xch + (size_t)(t & 1) * X_PAR;
    const int st = fetch_span<2>(Xc + X_HA, QD, tag, s_ha, p.abort_word, s_flag);
    T2_MARK(ge == 0, 16);
    if (st & 1) return;
    if (st & 2) {
      if (tid < 256) publish(Xc + X_EP, e * TS + base + tid, 0.f, tag | DONE_BIT);
      return;
    }
    {
      float hv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) hv[j] = s_ha[lane + 64 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += wq[i * 16 + j] * hv[j];
        const float qv = wsum(s);
        if (lane == 0) s_q[4 * wave + i] = qv;
      }
    }
    __syncthreads();
    {
      const float4* q4 = reinterpret_cast<const float4*>(s_q + ND * dq);
      const float4* v4 = reinterpret_cast<const float4*>(s_v + ND * dq);
      float e0 = 0.f, e1 = 0.f;
#pragma unroll
      for (int i = 0; i < ND / 4; ++i) {
        const float4 qv = q4[i], v = v4[i];
        e0 += v.x * fast_tanhf(qv.x + pl[4 * i]) + v.z * fast_tanhf(qv.z + pl[4 * i + 2]);
        e1 += v.y * fast_tanhf(qv.y + pl[4 * i + 1]) + v.w * fast_tanhf(qv.w + pl[4 * i + 3]);
      }
      s_ep[dq * 256 + pos] = e0 + e1;
    }
    __syncthreads();
    if (tid < 256) publish(Xc + X_EP, e * TS + base + tid, s_ep[tid] + s_ep[256 + tid], tag);
    T2_MARK(ge == 0, 17);
    // ---- off the serial chain: this step's alignment over all positions
    if (fetch_span<4 * TS / NT>(Xc + X_EP, 4 * TS, tag, s_e4, p.abort_word, s_flag) & 1) return;
    T2_MARK(ge == 0, 18);
    normalise_all<NC>(s_e4, vb, len, p.norm, s_al, s_red);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const int i = tid + NT * j;
      const float a = s_al[i];
      s_prev[LPAD + i] = a;
      s_cum[LPAD + i] += a;
      if (ge == 0 && i < T) p.align[(size_t)t * T + i] = a;
    }
    __syncthreads();
    location_terms();
    T2_MARK(ge == 0, 19);
  }
}

// Partial context of one chunk of 256 positions over 128 encoder columns: workgroup gc = 4 * chunk + column slice.
template <int NC>
__device__ void ctx_long_role(const T2P& p, int gc, float* sm) {
  constexpr int TS = NC * 256, NP = 64;
  const int tid = threadIdx.x;
  const int cs = gc & 3, chunk = gc >> 2, base = 256 * chunk;
  const int T = p.T, len = min(max(p.len[0], 0), T);
  float* s_e4 = sm;                    // [4][TS]
  float* s_al = s_e4 + 4 * TS;         // [TS]
  float* s_cp = s_al + TS;             // [4][128]
  float* s_red = s_cp + 4 * 128;       // [16]
  int* s_flag = reinterpret_cast<int*>(s_red + 16);
  const int col = tid & 127, tq = tid >> 7;
  float er[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int ps = base + NP * tq + i;
    er[i] = ps < len ? p.enc[(size_t)ps * ED + 128 * cs + col] : 0.f;
  }
  const float vb = p.v_b[0];
  if (tid == 0) *s_flag = 0;
  __syncthreads();
  for (int t = 0; t < p.S; ++t) {
    const unsigned tag = (unsigned)t + 1u;
    u64* Xc = p.xch + (size_t)(t & 1) * X_PAR;
    const int st = fetch_span<4 * TS / NT>(Xc + X_EP, 4 * TS, tag, s_e4, p.abort_word, s_flag);
    T2_MARK(gc == 0, 24);
    if (st & 1) return;
    if (st & 2) break;
    normalise_all<NC>(s_e4, vb, len, p.norm, s_al, s_red);
    __syncthreads();
    {
      const float4* a4 = reinterpret_cast<const float4*>(s_al + base + NP * tq);
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int i = 0; i < NP / 4; ++i) {
        const float4 av = a4[i];
        a0 += av.x * er[4 * i] + av.z * er[4 * i + 2];
        a1 += av.y * er[4 * i + 1] + av.w * er[4 * i + 3];
      }
      s_cp[tq * 128 + col] = a0 + a1;
    }
    __syncthreads();
    if (tid < 128)
      publish(Xc + X_CTX, chunk * ED + 128 * cs + tid, (s_cp[tid] + s_cp[128 + tid]) + (s_cp[256 + tid] + s_cp[384 + tid]), tag);
    T2_MARK(gc == 0, 25);
  }
}

template <int TM, int NC>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void t2_team_kernel(T2P p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int g = blockIdx.x;
  constexpr int g_ctx0 = G_EN0 + N_EN * NC;
  if (g < G_DEC0) att_role<NC>(p, g, sm);
  else if (g < G_PROJ0) dec_role<TM, NC>(p, g - G_DEC0, sm);
  else if (g < G_EN0) proj_role<TM, NC>(p, g - G_PROJ0, sm);
  else if constexpr (NC == 1) {
    if (g < g_ctx0) energy_role<TM>(p, g - G_EN0, sm);
    else ctx_role<TM>(p, g - g_ctx0, sm);
  } else {
    if (g < g_ctx0) energy_long_role<NC>(p, g - G_EN0, sm);
    else ctx_long_role<NC>(p, g - g_ctx0, sm);
  }
}

constexpr size_t LDS_ATT = (size_t)(LDS_W + PD + QD + NC_MAX * ED + 64 + 8 + 8 * NT) * sizeof(float);
static_assert(LDS_ATT <= 160 * 1024, "an attention-LSTM workgroup's LDS");
constexpr size_t LDS_DEC = (size_t)(LDS_W + DD + QD + (N_EN + 1) * NC_MAX * 256 + 16 + 32 + 8) * sizeof(float);
static_assert(LDS_DEC <= 160 * 1024, "a decoder-LSTM workgroup's LDS");
constexpr size_t LDS_EN = (size_t)(2 * LOCK * LOCF + 32 * LOCF + 32 + 2 * (TMAX + 32) + TMAX * (LOCF + 1) + QD + 32 +
                                   4 * TMAX + N_EN * TMAX + 8 + 8) * sizeof(float);
constexpr size_t TEAM_LDS = LDS_ATT > LDS_DEC ? (LDS_ATT > LDS_EN ? LDS_ATT : LDS_EN) : (LDS_DEC > LDS_EN ? LDS_DEC : LDS_EN);

// ---- packing
struct PackP {
  T2TeamWeights w;
  float *img_att, *lds_att, *img_dec, *lds_dec, *img_en, *conv_t, *loc_dense, *ctx_cols;
};

__global__ void pack_lstm_kernel(PackP p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int KA = PD + ED;  // attention_rnn.weight_ih columns: [prenet 256; context 512]
  const int KD = QD + ED;  // decoder_rnn.weight_ih columns: [h_a 1024; context 512]
  if (i < (size_t)N_ATT * NS_ATT * NT) {
    const int tid = (int)(i % NT), s = (int)((i / NT) % NS_ATT), a = (int)(i / ((size_t)NT * NS_ATT));
    const int lane = tid & 63, wave = tid >> 6;
    float v;
    if (s < A_HA) {
      const int ri = s / 4, j = s % 4, row = (ri & 3) * QD + 16 * a + 2 * wave + (ri >> 2);
      v = p.w.attn_w_ih[(size_t)row * KA + lane + 64 * j];
    } else if (s < A_B) {
      const int ri = (s - A_HA) / 16, j = (s - A_HA) % 16, row = (ri & 3) * QD + 16 * a + 2 * wave + (ri >> 2);
      v = p.w.attn_w_hh[(size_t)row * QD + lane + 64 * j];
    } else {
      const int ri = s - A_B, row = (ri & 3) * QD + 16 * a + 2 * wave + (ri >> 2);
      v = p.w.attn_b_ih[row] + p.w.attn_b_hh[row];
    }
    p.img_att[i] = v;
  }
  if (i < (size_t)N_ATT * LDS_W) {
    // [wave][ri][j][lane]: columns 256 + lane + 64 j of the wave's row ri
    const int lane = (int)(i & 63), j = (int)((i >> 6) & 7), ri = (int)((i >> 9) & 7), wave = (int)((i >> 12) & 7);
    const int a = (int)(i / LDS_W), row = (ri & 3) * QD + 16 * a + 2 * wave + (ri >> 2);
    p.lds_att[i] = p.w.attn_w_ih[(size_t)row * KA + PD + lane + 64 * j];
  }
  if (i < (size_t)N_DEC * NS_DEC * NT) {
    const int tid = (int)(i % NT), s = (int)((i / NT) % NS_DEC), d = (int)(i / ((size_t)NT * NS_DEC));
    const int lane = tid & 63, wave = tid >> 6, u = 8 * d + wave;
    float v;
    if (s < D_B) {
      const int g = s / 16, j = s % 16;
      v = p.w.dec_w_ih[(size_t)(g * DD + u) * KD + lane + 64 * j];
    } else {
      const int g = s - D_B;
      v = p.w.dec_b_ih[g * DD + u] + p.w.dec_b_hh[g * DD + u];
    }
    p.img_dec[i] = v;
  }
  if (i < (size_t)N_DEC * LDS_W) {
    // [wave][g][j][lane]: column lane + 64 j of weight_hh row g * 1024 + unit
    const int lane = (int)(i & 63), j = (int)((i >> 6) & 15), g = (int)((i >> 10) & 3), wave = (int)((i >> 12) & 7);
    const int d = (int)(i / LDS_W), u = 8 * d + wave;
    p.lds_dec[i] = p.w.dec_w_hh[(size_t)(g * DD + u) * DD + lane + 64 * j];
  }
  if (i < (size_t)4 * DD * ED) {
    // [unit][gate][k]: decoder_rnn.weight_ih's columns over the context
    const int k = (int)(i % ED), g = (int)((i / ED) & 3), u = (int)(i / (4 * ED));
    p.ctx_cols[i] = p.w.dec_w_ih[(size_t)(g * DD + u) * KD + QD + k];
  }
  if (i < (size_t)N_EN * NS_EN * NT) {
    const int tid = (int)(i % NT), s = (int)((i / NT) % NS_EN), e = (int)(i / ((size_t)NT * NS_EN));
    const int lane = tid & 63, wave = tid >> 6, row = 32 * e + 4 * wave + s / 16, j = s % 16;
    p.img_en[i] = p.w.query_w[(size_t)row * QD + lane + 64 * j];
  }
  if (i < (size_t)LOCF * 2 * LOCK)  // [f][c][j] -> [(c, j)][f]
    p.conv_t[(i % (2 * LOCK)) * LOCF + i / (2 * LOCK)] = p.w.loc_conv_w ? p.w.loc_conv_w[i] : 0.f;
  if (i < (size_t)AD * LOCF) p.loc_dense[i] = p.w.loc_dense_w && p.w.loc_conv_w ? p.w.loc_dense_w[i] : 0.f;
}

__global__ void pack_proj_kernel(float* __restrict__ img, const float* __restrict__ rows_w, const float* __restrict__ rows_b,
                                 const float* __restrict__ a2f, const float* __restrict__ b2f, int r) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N_PROJ * NS_PROJ * NT) return;
  const int tid = (int)(i % NT), s = (int)((i / NT) % NS_PROJ), q = (int)(i / ((size_t)NT * NS_PROJ));
  const int lane = tid & 63, wave = tid >> 6, n_mel = MEL * r;
  auto src_row = [&](int rs) { return proj_src_row(rs, q, wave, n_mel); };
  float v = 0.f;
  if (s < P_B) {
    const int row = src_row(s / 16), j = s % 16;  // (the columns over h_d; those over the context: ctx_proj)
    if (row >= 0) v = rows_w[(size_t)row * PROJ_K + lane + 64 * j];
  } else if (s < P_W2) {
    const int row = src_row(s - P_B);
    if (row >= 0) v = rows_b[row];
  } else if (s < P_B2) {
    const int i2 = (s - P_W2) / 4, j = (s - P_W2) % 4;
    v = a2f[(size_t)(16 * q + 2 * wave + i2) * PD + lane + 64 * j];
  } else {
    v = b2f[16 * q + 2 * wave + (s - P_B2)];
  }
  img[i] = v;
}

}  // namespace

struct t2_team {
  float *img_att, *lds_att, *img_dec, *lds_dec, *img_proj, *img_en, *conv_t, *loc_dense;
  float *ctx_cols, *ctx_gates;  // [4096][512] context columns of the decoder LSTM; their products with one utterance
  float *proj_cols, *ctx_proj;  // [80 r + 272][1536] the composed projection rows (own copy); the same products
  int proj_rows;
  float *v_w, *v_b, *b1f, *a2f, *b2f;  // own copies
  int proj_r;
  u64* xch;
  int* abort_word;
};

namespace {
int dup(float** dst, const float* src, size_t n, hipStream_t st) {
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(dst), n * sizeof(float)));
  TTSX_HIP(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  return TTSX_OK;
}
}  // namespace

int t2_team_create(const T2TeamWeights* w, hipStream_t st, t2_team** out) {
  TTSX_ARG(w && out, "t2_team_create: null argument");
  *out = nullptr;
  if (!(w->query_w && w->v_w && w->v_b)) return TTSX_OK;  // (a decoder with another attention type)
  {
    // every workgroup of a team needs a CU of its own at the same time: no team on a device (or partition) with
    // fewer CUs than the largest team
    int dev = 0, cus = 0;
    TTSX_HIP(hipGetDevice(&dev));
    TTSX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus < G_EN0 + 2 * N_EN * NC_MAX) return TTSX_OK;
  }
  t2_team* t = new t2_team();
  memset(t, 0, sizeof(*t));
  ScopeExit on_fail([&] { t2_team_destroy(t); });
  auto alloc = [&](float** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(float)); };
  TTSX_HIP(alloc(&t->img_att, (size_t)N_ATT * NS_ATT * NT));
  TTSX_HIP(alloc(&t->lds_att, (size_t)N_ATT * LDS_W));
  TTSX_HIP(alloc(&t->img_dec, (size_t)N_DEC * NS_DEC * NT));
  TTSX_HIP(alloc(&t->lds_dec, (size_t)N_DEC * LDS_W));
  TTSX_HIP(alloc(&t->img_proj, (size_t)N_PROJ * NS_PROJ * NT));
  TTSX_HIP(alloc(&t->img_en, (size_t)N_EN * NS_EN * NT));
  TTSX_HIP(alloc(&t->conv_t, (size_t)LOCF * 2 * LOCK));
  TTSX_HIP(alloc(&t->loc_dense, (size_t)AD * LOCF));
  TTSX_HIP(alloc(&t->ctx_cols, (size_t)4 * DD * ED));
  TTSX_HIP(alloc(&t->ctx_gates, (size_t)4 * DD * NC_MAX * 256));
  TTSX_HIP(alloc(&t->proj_cols, (size_t)PROJ_ROWS_MAX * PROJ_K));
  TTSX_HIP(alloc(&t->ctx_proj, (size_t)PROJ_ROWS_MAX * NC_MAX * 256));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&t->xch), sizeof(u64) * 2 * X_PAR));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&t->abort_word), sizeof(int)));
  int rc;
  if ((rc = dup(&t->v_w, w->v_w, AD, st)) || (rc = dup(&t->v_b, w->v_b, 1, st)) || (rc = dup(&t->b1f, w->b1f, PD, st)) ||
      (rc = dup(&t->a2f, w->a2f, (size_t)PD * PD, st)) || (rc = dup(&t->b2f, w->b2f, PD, st)))
    return rc;
  PackP pp{*w, t->img_att, t->lds_att, t->img_dec, t->lds_dec, t->img_en, t->conv_t, t->loc_dense, t->ctx_cols};
  const size_t n = std::max({(size_t)N_ATT * NS_ATT * NT, (size_t)N_DEC * NS_DEC * NT, (size_t)N_DEC * LDS_W,
                             (size_t)N_ATT * LDS_W, (size_t)4 * DD * ED, (size_t)N_EN * NS_EN * NT});  // every index range
  pack_lstm_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(pp);
  TTSX_LAUNCH_CHECK();
  for (const void* k : {reinterpret_cast<const void*>(t2_team_kernel<128, 1>), reinterpret_cast<const void*>(t2_team_kernel<256, 1>),
                        reinterpret_cast<const void*>(t2_team_kernel<256, NC_MAX>)})
    TTSX_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TEAM_LDS));
  TTSX_HIP(hipStreamSynchronize(st));
  on_fail.dismiss();
  *out = t;
  return TTSX_OK;
}

void t2_team_destroy(t2_team* t) {
  if (!t) return;
  float* bufs[] = {t->img_att, t->lds_att, t->img_dec, t->lds_dec, t->img_proj, t->img_en, t->conv_t,
                   t->loc_dense, t->ctx_cols, t->ctx_gates, t->proj_cols, t->ctx_proj, t->v_w, t->v_b, t->b1f, t->a2f, t->b2f};
  for (float* b : bufs)
    if (b) (void)hipFree(b);
  if (t->xch) (void)hipFree(t->xch);
  if (t->abort_word) (void)hipFree(t->abort_word);
  delete t;
}

int t2_team_set_proj(t2_team* t, const float* rows_w, const float* rows_b, int r, hipStream_t st) {
  TTSX_ARG(t && rows_w && rows_b, "t2_team_set_proj: null argument");
  t->proj_r = 0;
  if (r > 2) return TTSX_OK;  // more mel rows than a projection wave holds: the launch chain runs such models
  const size_t n = (size_t)N_PROJ * NS_PROJ * NT;
  pack_proj_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(t->img_proj, rows_w, rows_b, t->a2f, t->b2f, r);
  TTSX_LAUNCH_CHECK();
  t->proj_rows = MEL * r + 16 + PD;  // frames, the stop row and padding, prenet layer 1
  static_assert((MEL + 16 + PD) % 16 == 0 && (2 * MEL + 16 + PD) % 16 == 0 && 2 * MEL + 16 + PD <= PROJ_ROWS_MAX,
                "whole 16-row blocks of the composed projection");
  TTSX_HIP(hipMemcpyAsync(t->proj_cols, rows_w, sizeof(float) * (size_t)t->proj_rows * PROJ_K, hipMemcpyDeviceToDevice, st));
  t->proj_r = r;
  return TTSX_OK;
}

#ifdef TTSX_TEAM_TIMING
extern "C" int ttsx_debug_t2_team_timing(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(t2_team_marks), sizeof(long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

bool t2_team_supports(const t2_team* t, int B, int T, int r) {
  return t && B == 1 && T >= 1 && T <= NC_MAX * 256 && r >= 1 && r <= 2 && t->proj_r == r;
}

int t2_team_run(t2_team* t, const float* enc, const float* P, const int32_t* len, int T, int r, int max_steps,
                float stop_threshold, int fixed_steps, int norm, float* mel, float* align, float* stop,
                int32_t* n_steps, const T2TeamState* state_out, int* steps_run_host, hipStream_t st) {
  TTSX_ARG(t && enc && P && len && mel && align && stop && n_steps, "t2_team_run: null argument");
  TTSX_ARG(t2_team_supports(t, 1, T, r), "t2_team_run: T=%d r=%d outside what the team kernel covers", T, r);
  TTSX_ARG(max_steps >= 1, "t2_team_run: max_steps=%d", max_steps);
  std::lock_guard<std::mutex> one_team(ttsx_team_mutex());  // until the synchronisation below
  // tags start at 1: all-zero words match no step
  TTSX_HIP(hipMemsetAsync(t->xch, 0, sizeof(u64) * 2 * X_PAR, st));
  TTSX_HIP(hipMemsetAsync(t->abort_word, ttsx_take_wait_failure(TTSX_WAIT_T2_TEAM) ? 1 : 0, sizeof(int), st));
  T2P p{};
  p.img_att = t->img_att, p.lds_att = t->lds_att, p.img_dec = t->img_dec, p.lds_dec = t->lds_dec;
  p.img_proj = t->img_proj, p.img_en = t->img_en, p.conv_t = t->conv_t, p.loc_dense = t->loc_dense;
  p.v_w = t->v_w, p.v_b = t->v_b, p.b1f = t->b1f, p.enc = enc, p.P = P, p.len = len, p.xch = t->xch;
  p.ctx_gates = t->ctx_gates, p.ctx_proj = t->ctx_proj;
  p.abort_word = t->abort_word, p.mel = mel, p.align = align, p.stop = stop, p.n_steps = n_steps;
  if (state_out) p.st = *state_out;
  p.T = T, p.r = r, p.S = max_steps, p.free_run = fixed_steps ? 0 : 1, p.norm = norm, p.thr = stop_threshold;
  // (attention roles for 128 and for 256 positions, a thread owning one position either way; beyond that, four
  // chunks of 256 positions with four energy and four context workgroups each)
  // A plain launch: every workgroup takes a whole CU (512 threads x 256 registers), the team is smaller than the
  // chip, and every wait is bounded.  (A cooperative launch would refuse a team that cannot be resident, but
  // rocprofv3's kernel trace crashes at exit in a process that has made one.)
  const int grid = G_EN0 + 2 * N_EN * (T <= 256 ? 1 : NC_MAX);
  const int TS = T <= 128 ? 128 : T <= 256 ? 256 : NC_MAX * 256;
  ctx_products_kernel<ED><<<dim3(4 * DD / 64, TS / 64, 1), 256, 0, st>>>(t->ctx_cols, 4 * DD, ED, enc, len, 0, T, TS, t->ctx_gates);
  TTSX_LAUNCH_CHECK();
  ctx_products_kernel<ED><<<dim3((t->proj_rows + 63) / 64, TS / 64, 1), 256, 0, st>>>(t->proj_cols + DD, t->proj_rows, PROJ_K, enc,
                                                                                      len, 0, T, TS, t->ctx_proj);
  TTSX_LAUNCH_CHECK();
  if (T <= 128) t2_team_kernel<128, 1><<<grid, NT, TEAM_LDS, st>>>(p);
  else if (T <= 256) t2_team_kernel<256, 1><<<grid, NT, TEAM_LDS, st>>>(p);
  else t2_team_kernel<256, NC_MAX><<<grid, NT, TEAM_LDS, st>>>(p);
  TTSX_LAUNCH_CHECK();
  int32_t host[2] = {0, 0};
  TTSX_HIP(hipMemcpyAsync(&host[0], n_steps, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipMemcpyAsync(&host[1], t->abort_word, sizeof(int), hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipStreamSynchronize(st));
  if (host[1]) {
    ttsx_set_error("t2_team_run: a workgroup of the decoder team waited too long for its inputs");
    return TTSX_ERR_STATE;
  }
  int run = max_steps;
  if (host[0] == 0) {
    // the stop token never rose above the threshold: every step ran
    TTSX_HIP(hipMemcpyAsync(n_steps, &run, sizeof(int32_t), hipMemcpyHostToDevice, st));
    TTSX_HIP(hipStreamSynchronize(st));
  } else if (!fixed_steps) {
    run = host[0];
  }
  if (steps_run_host) *steps_run_host = run;
  return TTSX_OK;
}
