// Tacotron (v1) greedy decoder for batch 1-2: one persistent launch per run, every decoder weight resident on
// chip for the whole utterance.
//
// What a step computes follows Decoder.forward in eval mode (Tacotron/TTS/models.py:586-640) with Prenet
// (:250-253), AttentionRNNCell.forward (:124-156), LocationSensitiveAttention.forward (:71-88), the two residual
// decoder GRUs (:612-620), proj_to_mel + sigmoid (:622-623), StopNet (:663-667) and the stop rule (:635-640).
//
// The decoder is 1.66 M parameters (6.7 MB).  As a chain of launches a step costs seven launch boundaries
// (~4.4 us each) for a microsecond of arithmetic; here a TEAM of 36 workgroups per utterance runs the whole loop:
//   * 32 "row" workgroups of 8 waves.  Workgroup g owns hidden units 8g .. 8g+7 of each of the three GRUs (one
//     unit per wave), rows 8g .. 8g+7 of prenet layer 1 and of project_to_decoder_in, rows 5g .. 5g+4 of
//     proj_to_mel and its 8 columns of the query layer.  Those weights (112 floats per thread) stay in REGISTERS
//     from the first step to the last; prenet layer 2 (128 x 256), which every workgroup evaluates in full, sits
//     in LDS.  Nothing but activations crosses the chip after the first microseconds.
//   * 4 attention workgroups, 32 encoder positions each (T <= 128): the positions' annotations in registers, the
//     location-layer weights and both alignment rows in LDS; per step each adds up the 32 query shares, evaluates
//     the energies of its positions and publishes their sigmoids.
// The context never travels.  It is a weighted sum of encoder outputs, ctx = sum_p a[p] enc[p], and all that the
// decoder does with it is multiply it by weight rows (attention GRU, GRU 1 composed with project_to_decoder_in,
// project_to_decoder_in itself), so W ctx = sum_p a[p] (W enc[p]): the products G[row][p] = W[row] . enc[p] are
// formed once per utterance (ctx_products_kernel), a wave keeps those of its seven rows in registers (two per
// row), and every workgroup normalises the published sigmoids itself (same operations, same order: same bits).
// Activations travel through small exchange buffers in global memory as 64-bit words (step tag, value): a
// consumer re-reads its elements until all of them carry the tag of the step it needs, so an exchange costs one
// store-to-load round trip and needs neither counters nor fences.  Coherent polling reads are served by the
// memory side of the chip at about one 128-byte line per 0.5 ns per channel, and 256 waves that each re-read a
// whole vector saturate that (measured: 2.3 us per exchange with one word per lane and poll, 5.4 us with eight).
// So a workgroup polls every word ONCE (thread i polls word i and shares it through LDS), and a producer
// writes a word to eight replicas (lanes 0-7 of the publishing wave, one store instruction) that lie in
// different channels; workgroup g reads replica g mod 8.  Six exchanges per step lie on the serial
// chain (x1 | h_att + query shares | alignment | h1, d1 | d2 | frames).  GRU 1 takes input weights composed with
// project_to_decoder_in (fp64, once), which saves a seventh.  Buffers are double-buffered by step parity: a
// vector that a phase both reads (previous step) and rewrites cannot be overwritten under a slow reader.
// A GRU's state update is ONE instruction stream: the waves leave their gate sums in LDS and the first wave
// updates the workgroup's eight units side by side on its lanes (two waves of a SIMD take turns on the
// transcendental unit; eight streams cost three times one), through the hardware exponential and reciprocal.
// Every workgroup, the attention ones included, evaluates the stop rule of a step itself from the published
// frames (same instructions, same result), so the team leaves the loop together.  Every wait is bounded: a wave
// that sees no progress for 50 ms raises an abort word that ends all other waits, and the host reports it.
#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"
#include "ctx_products.h"
#include "t1_team.h"

namespace {

typedef unsigned long long u64;

constexpr int TG = 32;            // row workgroups per team
constexpr int NAW = 4;            // attention workgroups per team
constexpr int TEAM = TG + NAW;
constexpr int NT = 512;
constexpr int MEM = 400, ODIM = 160, PRE0 = 256, PRE1 = 128, RNN = 256, ENC = 256, ATT = 128;
constexpr int LOCF = 32, LOCK = 31, LPAD = 15, TMAX = 128;
constexpr int PW = TMAX / NAW;    // positions per attention workgroup
// rows whose context columns are held as products with the encoder outputs: attention GRU (r, z, n), GRU 1
// composed (r, z, n), project_to_decoder_in
constexpr int G_ATT = 0, G_1 = 3 * RNN, G_D0 = 6 * RNN, G_ROWS = 7 * RNN;

// ---- per-thread register image: slot s of thread tid of row workgroup g at img[(g * NS + s) * NT + tid]
enum {
  S_P1 = 0,                                                   // 7: prenet-1 row 8g+w, k = lane + 64 j
  S_A_RX = 7, S_A_ZX = 9, S_A_NX = 11,                        // 2 each: attention GRU input rows over x2
  S_A_RH = 13, S_A_ZH = 17, S_A_NH = 21,                      // 4 each: attention GRU recurrent rows
  S_QS = 25,                                                  // 8: W_q[d = tid][8g + j] (threads < 128)
  S_1_RC = 33, S_1_ZC = 37, S_1_NC = 41,                      // 4 each: GRU 1 input rows composed, over h_att
  S_1_RH = 45, S_1_ZH = 49, S_1_NH = 53,                      // 4 each
  S_D0 = 57,                                                  // 4: project_to_decoder_in row 8g+w over h_att
  S_2_RX = 61, S_2_ZX = 65, S_2_NX = 69, S_2_RH = 73, S_2_ZH = 77, S_2_NH = 81,  // 4 each: GRU 2
  S_MEL = 85,                                                 // 4: proj_to_mel row 5g+w (waves 0..4)
  S_STOP = 89,                                                // 7: stopnet weights over [d2; frames]
  S_B_P1 = 96, S_B_A = 97 /* r, z, n_in, n_h */, S_B_1 = 101 /* r, z, n_in, n_h, d0 */, S_B_2 = 106, S_B_MEL = 110,
  S_B_STOP = 111,
  NS = 112
};

// ---- exchange buffers of one team (u64 words); [2] = step parity
constexpr int X_OUT = 0, X_OUT_N = 192, X_X1 = X_OUT + 2 * X_OUT_N, X_HATT = X_X1 + 2 * 256, X_QS = X_HATT + 2 * 256,
              X_QS_N = TG * ATT, X_S = X_QS + 2 * X_QS_N, X_S_N = TMAX /* sigmoids of the energies */,
              X_H1 = X_S + 2 * X_S_N, X_D1 = X_H1 + 2 * 256, X_H2 = X_D1 + 2 * 256, X_D2 = X_H2 + 2 * 256,
              X_TOTAL = X_D2 + 2 * 256;
constexpr int NREP = 8;                 // replicas of every exchange buffer but the query shares
constexpr int X_REP = X_TOTAL + 544;    // replica stride in words (not a multiple of 4 KB: replicas in different channels)
constexpr int X_TEAM = NREP * X_REP;    // words per team

struct TeamP {
  const float* img;      // [TG][NS][NT]
  const float* w2t;      // [64][128][4] prenet layer 2: (k / 4, row, k % 4)
  const float* b2;       // [128]
  const float *mem_init, *att_init, *rnn_inits;
  const float *loc_conv_t /* [62][32] */, *loc_lin_w /* [128][32] */, *q_bias /* [128] query + location-linear bias */, *v_w;
  const float *stop_w, *stop_b;  // [416], [1]
  const float* annot;            // [B][T][128]
  const float* ctx_g;            // [B][G_ROWS][TMAX] context columns x encoder outputs
  const int32_t* len;            // [B] or null
  u64* xch;                      // [B][X_TEAM]
  int* abort_word;
  float *mel, *align, *stop;
  int32_t* n_steps;
  int T, max_steps;
};

#ifdef TTSX_TEAM_TIMING
// Tuning build: workgroup 0 / the attention workgroup of utterance 0 leave 100 MHz timestamps of step 50.
__device__ long long team_marks[64 + 4 * 64];
#define TEAM_MARK_G(slot)                                                                      \
  do {                                                                                         \
    if (b == 0 && t == 50 && threadIdx.x == 0) team_marks[64 + (slot) * 64 + g] = wall_clock64(); \
  } while (0)
#define TEAM_MARK(cond, i)                                            \
  do {                                                                \
    if ((cond) && t == 50 && threadIdx.x == 0) team_marks[i] = wall_clock64(); \
  } while (0)
#else
#define TEAM_MARK(cond, i) \
  do {                     \
  } while (0)
#define TEAM_MARK_G(slot) \
  do {                    \
  } while (0)
#endif

__device__ __forceinline__ void publish(u64* X, int idx, float v, unsigned tag) {
  __hip_atomic_store(X + idx, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// n <= 8 consecutive words (values in LDS) starting at word `first`, to all replicas, by ONE store instruction of
// the calling wave: lane 8 r + j writes word j of replica r, 8 n contiguous bytes per replica.  Eight waves that
// each wrote their own word into a line they share with another workgroup measured 2.0 us per exchange against
// 1.15 for this form (tools/exchange_probe.hip): partial writes of different waves to one line serialise.
__device__ __forceinline__ void publish_row(u64* X, int first, const float* vals, int n, unsigned tag, int lane) {
  const int j = lane & 7;
  if (j < n) publish(X + (size_t)(lane >> 3) * X_REP, first + j, vals[j], tag);
}

// One word per thread (none where !active; thread 0 of the workgroup may poll a second one): re-read until all
// words of the wave carry `tag`.  False: aborted (the caller must still reach the barrier that follows).
// (Keeping two or three reads in flight a fraction of a round trip apart, to see a word sooner, measured slower:
// 18.6 and 19.1 against 16.4 us per step - more polling reads delay the publishing stores.)
template <bool SECOND>
__device__ __forceinline__ bool poll(const u64* a, bool active, const u64* a2, unsigned tag, float* val, float* val2,
                                     int* abort_word) {
  long long t_wait = 0;
  for (long spin = 0;; ++spin) {
    u64 v = (u64)tag << 32, v2 = (u64)tag << 32;
    if (active) v = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (SECOND && threadIdx.x == 0) v2 = __hip_atomic_load(a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = (unsigned)(v >> 32) == tag && (unsigned)(v2 >> 32) == tag;
    *val = __uint_as_float((unsigned)v);
    if (SECOND) *val2 = __uint_as_float((unsigned)v2);
    if (__all(ok)) return true;
    if ((spin & 127) == 127 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    if (team_wait_expired(spin, t_wait)) {
      __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// tanh through one hardware exponential, as the energies of the launch-chain decoder compute it
__device__ __forceinline__ float fast_tanhf(float x) {
  const float t = __expf(2.f * x);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}

// Sum over the 64 lanes on the DPP network (a __shfl_xor butterfly goes through the LDS crossbar: ~100 cycles per
// step, and a step of the decoder holds sixteen such sums).  Every lane gets the total; the order of the
// additions is fixed, so every workgroup that reduces the same values gets the same bits.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wsum(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror: every lane of a row holds the row's sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int N>
__device__ __forceinline__ void load_init(const float* src, int n, float (&x)[N], int lane) {
#pragma unroll
  for (int j = 0; j < N; ++j) x[j] = lane + 64 * j < n ? src[lane + 64 * j] : 0.f;
}

// element `u` (wave-uniform) of a vector held as x[j] = v[lane + 64 j]
template <int N>
__device__ __forceinline__ float pick(const float (&x)[N], int u) {
  float v = x[0];
#pragma unroll
  for (int j = 1; j < N; ++j) v = (u >> 6) == j ? x[j] : v;
  return __shfl(v, u & 63, 64);
}

// StopNet on [d2; frames] (Tacotron/TTS/models.py:663-667) and the stop rule (:635-640) of the step with 1-based
// index tn.  Every workgroup of the team runs exactly this, so all of them take the same decision.
__device__ __forceinline__ bool stop_rule(const float (&ws)[7], float bs, const float (&d2)[4], const float (&out)[3],
                                          float a_last, int tn, int len, int max_steps, float* token) {
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) acc += ws[j] * d2[j];
#pragma unroll
  for (int j = 0; j < 3; ++j) acc += ws[4 + j] * out[j];
  const float s = sigmoidf_(wsum(acc) + bs);
  *token = s;
  return ((float)tn > (float)len / 4.f && (s > 0.6f || a_last > 0.6f)) || tn > max_steps;
}

// The logistic function through the hardware exponential and reciprocal (absolute error below 2e-7; four
// instructions on the step's serial chain instead of some twenty-five)
__device__ __forceinline__ float fast_sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

__device__ __forceinline__ float gru_new(float sr, float sz, float sni, float snh, float h_old) {
  const float r = fast_sigmoidf(sr), z = fast_sigmoidf(sz);
  const float n = fast_tanhf(sni + r * snh);
  return (1.f - z) * n + z * h_old;
}

// Alignment weights of positions lane and lane + 64 from the published sigmoids s[0..128) (zero beyond the
// utterance), and the weight of position `last`: every workgroup of the team runs exactly this.
__device__ __forceinline__ void alignment_of(const float* s, int last, float* a0, float* a1, float* a_last) {
  const int lane = threadIdx.x & 63;
  const float s0 = s[lane], s1 = s[64 + lane];
  const float sum = wsum(s0) + wsum(s1);
  *a0 = s0 / sum, *a1 = s1 / sum, *a_last = s[last] / sum;
}

// ------------------------------------------------------------------ a row workgroup
// LDS of a row workgroup (floats)
constexpr int RL_W2 = 0, RL_B2 = RL_W2 + PRE0 * PRE1, RL_PART = RL_B2 + PRE1, RL_X1 = RL_PART + 4 * PRE1,
              RL_G = RL_X1 + PRE0, RL_PUB = RL_G + 64, RL_FLAG = RL_PUB + 48, RL_RING = RL_FLAG + 8, RL_OUT = RL_RING + MEM,
              RL_HC = RL_OUT + 192, RL_D1 = RL_HC + 576, RL_D2 = RL_D1 + 512, RL_TOTAL = RL_D2 + 512;

__device__ void row_workgroup(const TeamP& p, int b, int g, float* sm) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u = 8 * g + wave;  // this wave's hidden unit / prenet row / project_to_decoder_in row
  float* s_w2 = sm + RL_W2;      // [64][128][4]: (k / 4, row, k % 4)
  float* s_b2 = sm + RL_B2;      // [128]
  float* s_part = sm + RL_PART;  // [4][128]
  float* s_x1 = sm + RL_X1;      // [256]
  float* s_g = sm + RL_G;        // [8 waves][8]: the gate sums of the wave's unit (r, z, n_in, n_h, d0)
  float* s_pub = sm + RL_PUB;    // [8] x1 | [8] frames before the sigmoid
  int* s_flag = reinterpret_cast<int*>(sm + RL_FLAG);
  float* ring = sm + RL_RING;    // [400] memory queue
  float* s_out = sm + RL_OUT;    // [160] frames of the previous step
  float* s_hc = sm + RL_HC;      // [256] h_att, [128] sigmoids of the energies
  float* s_d1 = sm + RL_D1;      // [256] d1, [256] h1
  float* s_d2 = sm + RL_D2;      // [256] d2, [256] h2
  u64* X = p.xch + (size_t)b * X_TEAM;                  // publishing base (replica 0)
  const u64* Xr = X + (size_t)(g % NREP) * X_REP;        // the replica this workgroup reads
  u64* Xl = X + (size_t)(lane >> 3) * X_REP;             // the replica lane 8 r + j publishes word j of a row to
  int* ab = p.abort_word;
  const int len = p.len ? min(max(p.len[b], 1), p.T) : p.T;
  const int ul = lane & 7;  // the unit (of the workgroup's eight) whose state the first waves update on this lane

  float w[NS], wg[7][2];
  {
    const float* img = p.img + ((size_t)g * NS) * NT + tid;
#pragma unroll
    for (int s = 0; s < NS; ++s) w[s] = img[(size_t)s * NT];
#pragma unroll
    for (int s = S_B_P1; s < NS; ++s) w[s] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w[s])));
    const float* G = p.ctx_g + (size_t)b * G_ROWS * TMAX + lane;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int row = k < 3 ? G_ATT + k * RNN + u : k < 6 ? G_1 + (k - 3) * RNN + u : G_D0 + u;
      wg[k][0] = G[(size_t)row * TMAX], wg[k][1] = G[(size_t)row * TMAX + 64];
    }
  }
  for (int i = tid; i < PRE0 * PRE1; i += NT) s_w2[i] = p.w2t[i];
  if (tid < PRE1) s_b2[tid] = p.b2[tid];
  for (int i = tid; i < MEM; i += NT) ring[i] = p.mem_init[i];
  if (tid == 0) *s_flag = 0;
  int head = 0;
  __syncthreads();

  // Recurrent state as every wave needs it (element lane + 64 j of each vector).  A vector published in step t is
  // fetched once, in step t, together with the vector whose arrival the workgroup waits for anyway, and kept.
  // The context's share of the attention GRU's rows is kept as per-lane partial sums (ca).
  float hatt[4], h1[4], h2[4], xd2[4] = {0.f, 0.f, 0.f, 0.f}, ca[3] = {0.f, 0.f, 0.f};
  float a_last = 0.f;
  load_init<4>(p.att_init, RNN, hatt, lane);
  load_init<4>(p.rnn_inits, RNN, h1, lane);
  load_init<4>(p.rnn_inits + RNN, RNN, h2, lane);
  // the states of unit 8g + ul as the updating waves hold them
  float own_att = p.att_init[8 * g + ul], own_1 = p.rnn_inits[8 * g + ul], own_2 = p.rnn_inits[RNN + 8 * g + ul], own_d1 = 0.f;
  for (int t = 0;; ++t) {
    const unsigned tag = (unsigned)t + 1u;  // what this step publishes; the previous step's data carries tag - 1
    const int cur = t & 1, prv = cur ^ 1;
    TEAM_MARK(b == 0 && g == 0, 0);
    if (t > 0) {
      // ---- the previous step's frames: into the memory queue
      {
        float v, v2;
        const bool ok = poll<false>(Xr + X_OUT + prv * X_OUT_N + min(tid, ODIM - 1), tid < ODIM, nullptr, tag - 1, &v, &v2, ab);
        if (tid < ODIM) {
          int ph = head + tid;
          ph = ph >= MEM ? ph - MEM : ph;
          ring[ph] = v;
          s_out[tid] = v;
        }
        if (!ok) *s_flag = 1;
        __syncthreads();
        if (*s_flag) return;
      }
      head = head + ODIM >= MEM ? head + ODIM - MEM : head + ODIM;
    }
    TEAM_MARK(b == 0 && g == 0, 1);
    // ---- prenet layer 1: row u over the memory queue
    {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int k = lane + 64 * j;
        int ph = head + k;
        ph = ph >= MEM ? ph - MEM : ph;
        const float xm = k < MEM ? ring[ph] : 0.f;
        acc += w[S_P1 + j] * xm;
      }
      const float x1 = fmaxf(wsum(acc) + w[S_B_P1], 0.f);
      if (lane == 0) s_pub[wave] = x1;
      __syncthreads();
      if (wave == 0) publish_row(X + X_X1 + cur * 256, 8 * g, s_pub, 8, tag, lane);
    }
    if (t > 0) {
      // ---- the stop rule of the previous step (behind the publication: a team that stops here has published
      // eight words of a step that does not exist, which nobody reads)
      float xo[3], ws[7];
#pragma unroll
      for (int j = 0; j < 3; ++j) xo[j] = lane + 64 * j < ODIM ? s_out[lane + 64 * j] : 0.f;
#pragma unroll
      for (int j = 0; j < 7; ++j) ws[j] = w[S_STOP + j];
      float token;
      const bool done = stop_rule(ws, w[S_B_STOP], xd2, xo, a_last, t, len, p.max_steps, &token);
      if (g == 0 && wave == 0 && lane == 0) {
        p.stop[(size_t)b * (p.max_steps + 1) + (t - 1)] = token;
        if (done) p.n_steps[b] = t;
      }
      if (done) return;
    }
    TEAM_MARK(b == 0 && g == 0, 2);
    // ---- prenet layer 2, whole, by every workgroup: thread (row, quarter of K)
    float x2[2];
    {
      {
        float v, v2;
        const bool ok = poll<false>(Xr + X_X1 + cur * 256 + (tid & (PRE0 - 1)), tid < PRE0, nullptr, tag, &v, &v2, ab);
        if (tid < PRE0) s_x1[tid] = v;
        if (!ok) *s_flag = 1;
        __syncthreads();
        if (*s_flag) return;
      }
      TEAM_MARK(b == 0 && g == 0, 3);
      const int kq = wave >> 1, row = (wave & 1) * 64 + lane;
      const float4* wq = reinterpret_cast<const float4*>(s_w2) + (size_t)(16 * kq) * PRE1 + row;
      const float4* xq = reinterpret_cast<const float4*>(s_x1 + 64 * kq);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 wv = wq[i * PRE1], xv = xq[i];
        a0 += wv.x * xv.x, a1 += wv.y * xv.y, a2 += wv.z * xv.z, a3 += wv.w * xv.w;
      }
      const float acc = (a0 + a1) + (a2 + a3);
      s_part[kq * PRE1 + row] = acc;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = lane + 64 * j;
        x2[j] = fmaxf(((s_part[r] + s_part[PRE1 + r]) + (s_part[2 * PRE1 + r] + s_part[3 * PRE1 + r])) + s_b2[r], 0.f);
      }
    }
    TEAM_MARK(b == 0 && g == 0, 4);
    // ---- attention GRU, unit u, on [x2; ctx(t-1)] and h_att(t-1); the workgroup's share of the query
    {
      float sr = ca[0], sz = ca[1], sni = ca[2], snh = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) sr += w[S_A_RX + j] * x2[j], sz += w[S_A_ZX + j] * x2[j], sni += w[S_A_NX + j] * x2[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float h = hatt[j];
        sr += w[S_A_RH + j] * h, sz += w[S_A_ZH + j] * h, snh += w[S_A_NH + j] * h;
      }
      sr = wsum(sr) + w[S_B_A], sz = wsum(sz) + w[S_B_A + 1];
      sni = wsum(sni) + w[S_B_A + 2], snh = wsum(snh) + w[S_B_A + 3];
      if (lane == 0) reinterpret_cast<float4*>(s_g)[2 * wave] = make_float4(sr, sz, sni, snh);
      __syncthreads();
      if (wave < 2) {  // (both waves that hold query columns update the eight units: the same stream twice)
        const float4 gv = reinterpret_cast<const float4*>(s_g)[2 * ul];
        own_att = gru_new(gv.x, gv.y, gv.z, gv.w, own_att);
        if (wave == 0) publish(Xl + X_HATT + cur * 256, 8 * g + ul, own_att, tag);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) q += w[S_QS + j] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(own_att), j));
        publish(X + X_QS + cur * X_QS_N, g * ATT + tid, q, tag);
      }
    }
    TEAM_MARK(b == 0 && g == 0, 5);
    // ---- GRU 1 (input weights composed with project_to_decoder_in) on [h_att; ctx] and h1(t-1); d0
    {
      float c1[4];
      {
        float v, v2;
        const u64* a = tid < RNN ? Xr + X_HATT + cur * 256 + tid : Xr + X_S + cur * X_S_N + min(tid - RNN, TMAX - 1);
        const bool ok = poll<false>(a, tid < RNN + TMAX, nullptr, tag, &v, &v2, ab);
        if (tid < RNN + TMAX) s_hc[tid] = v;
        if (!ok) *s_flag = 1;
        __syncthreads();
        if (*s_flag) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) hatt[j] = s_hc[lane + 64 * j];
        float a0, a1;
        alignment_of(s_hc + RNN, len - 1, &a0, &a1, &a_last);
#pragma unroll
        for (int k = 0; k < 3; ++k) ca[k] = wg[k][0] * a0 + wg[k][1] * a1;  // for the next step's attention GRU
#pragma unroll
        for (int k = 0; k < 4; ++k) c1[k] = wg[3 + k][0] * a0 + wg[3 + k][1] * a1;
      }
      TEAM_MARK(b == 0 && g == 0, 6);
      float sr = c1[0], sz = c1[1], sni = c1[2], snh = 0.f, sd = c1[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sr += w[S_1_RC + j] * hatt[j], sz += w[S_1_ZC + j] * hatt[j];
        sni += w[S_1_NC + j] * hatt[j], sd += w[S_D0 + j] * hatt[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) sr += w[S_1_RH + j] * h1[j], sz += w[S_1_ZH + j] * h1[j], snh += w[S_1_NH + j] * h1[j];
      sr = wsum(sr) + w[S_B_1], sz = wsum(sz) + w[S_B_1 + 1];
      sni = wsum(sni) + w[S_B_1 + 2], snh = wsum(snh) + w[S_B_1 + 3];
      const float d0 = wsum(sd) + w[S_B_1 + 4];
      if (lane == 0) {
        reinterpret_cast<float4*>(s_g)[2 * wave] = make_float4(sr, sz, sni, snh);
        s_g[8 * wave + 4] = d0;
      }
      __syncthreads();
      if (wave == 0) {
        const float4 gv = reinterpret_cast<const float4*>(s_g)[2 * ul];
        own_1 = gru_new(gv.x, gv.y, gv.z, gv.w, own_1);
        own_d1 = own_1 + s_g[8 * ul + 4];
        publish(Xl + X_D1 + cur * 256, 8 * g + ul, own_d1, tag);
        publish(Xl + X_H1 + cur * 256, 8 * g + ul, own_1, tag);
      }
    }
    TEAM_MARK(b == 0 && g == 0, 7);
    TEAM_MARK_G(0);
    // ---- GRU 2 on d1 and h2(t-1); d2 = h2 + d1
    {
      float x[4];
      {
        float v, v2;
        const u64* a = tid < RNN ? Xr + X_D1 + cur * 256 + tid : Xr + X_H1 + cur * 256 + (tid - RNN);
        const bool ok = poll<false>(a, true, nullptr, tag, &v, &v2, ab);
        s_d1[tid] = v;
        TEAM_MARK_G(1);
        if (!ok) *s_flag = 1;
        __syncthreads();
        if (*s_flag) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = s_d1[lane + 64 * j], h1[j] = s_d1[RNN + lane + 64 * j];
      }
      TEAM_MARK(b == 0 && g == 0, 8);
      TEAM_MARK_G(2);
      float sr = 0.f, sz = 0.f, sni = 0.f, snh = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sr += w[S_2_RX + j] * x[j] + w[S_2_RH + j] * h2[j];
        sz += w[S_2_ZX + j] * x[j] + w[S_2_ZH + j] * h2[j];
        sni += w[S_2_NX + j] * x[j], snh += w[S_2_NH + j] * h2[j];
      }
      sr = wsum(sr) + w[S_B_2], sz = wsum(sz) + w[S_B_2 + 1];
      sni = wsum(sni) + w[S_B_2 + 2], snh = wsum(snh) + w[S_B_2 + 3];
      if (lane == 0) reinterpret_cast<float4*>(s_g)[2 * wave] = make_float4(sr, sz, sni, snh);
      __syncthreads();
      if (wave == 0) {
        const float4 gv = reinterpret_cast<const float4*>(s_g)[2 * ul];
        own_2 = gru_new(gv.x, gv.y, gv.z, gv.w, own_2);
        publish(Xl + X_D2 + cur * 256, 8 * g + ul, own_2 + own_d1, tag);
        publish(Xl + X_H2 + cur * 256, 8 * g + ul, own_2, tag);
      }
    }
    TEAM_MARK(b == 0 && g == 0, 9);
    // ---- frames: row 5g + wave of proj_to_mel (waves 0..4), sigmoid
    {
      float v, v2;
      const u64* a = tid < RNN ? Xr + X_D2 + cur * 256 + tid : Xr + X_H2 + cur * 256 + (tid - RNN);
      const bool ok = poll<false>(a, true, nullptr, tag, &v, &v2, ab);
      s_d2[tid] = v;
      if (!ok) *s_flag = 1;
      __syncthreads();
      if (*s_flag) return;
#pragma unroll
      for (int j = 0; j < 4; ++j) xd2[j] = s_d2[lane + 64 * j], h2[j] = s_d2[RNN + lane + 64 * j];
    }
    TEAM_MARK(b == 0 && g == 0, 10);
    {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += w[S_MEL + j] * xd2[j];
      const float pre = wsum(acc) + w[S_B_MEL];
      if (wave < 5 && lane == 0) s_pub[8 + wave] = pre;
      __syncthreads();
      if (wave == 0 && ul < 5) {
        const float o = sigmoidf_(s_pub[8 + ul]);
        publish(Xl + X_OUT + cur * X_OUT_N, 5 * g + ul, o, tag);
        if (lane < 5) p.mel[((size_t)b * (p.max_steps + 1) + t) * ODIM + 5 * g + lane] = o;
      }
    }
    TEAM_MARK(b == 0 && g == 0, 11);
  }
}

// ------------------------------------------------------------------ an attention workgroup: positions 32 a .. 32 a + 31
__device__ void attention_workgroup(const TeamP& p, int b, int a, float* sm) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.T, len = p.len ? min(max(p.len[b], 1), T) : T;
  float* s_wl = sm;                                  // [62][32] location conv, filter fastest
  float* s_wld = s_wl + 2 * LOCK * LOCF;             // [128][32] location linear layer
  float* s_v = s_wld + ATT * LOCF;                   // [128]
  float* s_a = s_v + 2 * ATT;                        // [2][TMAX + 32] previous / cumulative alignment of ALL positions, 15 zeros in front
  float* s_cf = s_a + 2 * (TMAX + 32);               // [PW][33] conv features of the own positions
  float* s_q = s_cf + PW * (LOCF + 1);               // [4][128] partial sums of the query (+ biases in the first)
  float* s_ep = s_q + 4 * ATT;                       // [16 dim groups][PW]
  float* s_s = s_ep + 16 * PW;                       // [TMAX] sigmoids of all positions
  float* s_st = s_s + TMAX;                          // [256] d2, [160] frames: inputs of the stop rule
  int* s_flag = reinterpret_cast<int*>(s_st + 512);
  u64* X = p.xch + (size_t)b * X_TEAM;               // replica 0: this workgroup's reads; base of what it publishes
  int* ab = p.abort_word;
  const float* annot = p.annot + (size_t)b * T * ATT;

  for (int i = tid; i < 2 * LOCK * LOCF; i += NT) s_wl[i] = p.loc_conv_t[i];
  for (int i = tid; i < ATT * LOCF; i += NT) s_wld[i] = p.loc_lin_w[i];
  if (tid < ATT) s_v[tid] = p.v_w[tid];
  const float qb_d = p.q_bias[tid & (ATT - 1)];  // query bias + location-linear bias
  for (int i = tid; i < 2 * (TMAX + 32); i += NT) s_a[i] = 0.f;
  if (tid == 0) *s_flag = 0;
  // energy role: position 32 a + pos_l, attention dims 8 dg .. 8 dg + 7 (a wave has two dg: the weights it reads
  // from LDS are two broadcasts)
  const int pos_l = tid & (PW - 1), dg = tid / PW, pos = PW * a + pos_l;
  float pa[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pa[i] = pos < len ? annot[(size_t)pos * ATT + 8 * dg + i] : 0.f;
  float ws[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) ws[j] = lane + 64 * j < RNN + ODIM ? p.stop_w[lane + 64 * j] : 0.f;
  const float bs = p.stop_b[0];
  __syncthreads();

  // Location conv of the own positions, thread (filter f, 2 positions), then the location-linear layer:
  // pl[i] = annot[pos][d] + (W_ld . conv features of pos)[d].  Both depend on earlier alignments only, so the
  // workgroup evaluates them for the coming step right after it has published, while the row workgroups run
  // their GRUs.
  float pl[8];
  auto location_terms = [&]() {
    {
      const int f = tid & 31, tp = tid >> 5;
      float acc[2] = {0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const float* al = s_a + ch * (TMAX + 32) + PW * a + 2 * tp;
        float win[LOCK + 1];
#pragma unroll
        for (int j = 0; j < LOCK + 1; ++j) win[j] = al[j];
#pragma unroll
        for (int j = 0; j < LOCK; ++j) {
          const float wv = s_wl[(ch * LOCK + j) * LOCF + f];
          acc[0] += wv * win[j], acc[1] += wv * win[j + 1];
        }
      }
      s_cf[(2 * tp) * (LOCF + 1) + f] = acc[0], s_cf[(2 * tp + 1) * (LOCF + 1) + f] = acc[1];
    }
    __syncthreads();
    float cf[LOCF];
#pragma unroll
    for (int f = 0; f < LOCF; ++f) cf[f] = s_cf[pos_l * (LOCF + 1) + f];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4* wl = reinterpret_cast<const float4*>(s_wld + (8 * dg + i) * LOCF);
      float loc = 0.f;
#pragma unroll
      for (int f4 = 0; f4 < LOCF / 4; ++f4) {
        const float4 w4 = wl[f4];
        loc += w4.x * cf[4 * f4] + w4.y * cf[4 * f4 + 1] + w4.z * cf[4 * f4 + 2] + w4.w * cf[4 * f4 + 3];
      }
      pl[i] = pa[i] + loc;
      // the value is needed HERE, before the next wait (the compiler would otherwise sink the arithmetic behind
      // the wait and hold the weight reads across it)
      asm volatile("" : "+v"(pl[i])::"memory");
    }
  };
  location_terms();
  for (int t = 0;; ++t) {
    const unsigned tag = (unsigned)t + 1u;
    const int cur = t & 1;
    TEAM_MARK(b == 0 && a == 0, 17);
    // ---- query = sum of the 32 shares (+ biases): thread (d, tg) reads 8 words, all of them in every round (a
    // read inside a branch would wait for its own round trip); the four partial sums meet in the energies
    {
      const int d = tid & (ATT - 1), tg = tid >> 7;
      const u64* Q = X + X_QS + cur * X_QS_N + (size_t)(8 * tg) * ATT + d;
      float val[8];
      bool fail = false;
      long long t_wait = 0;
      for (long spin = 0;; ++spin) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const u64 v = __hip_atomic_load(Q + (size_t)j * ATT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && (unsigned)(v >> 32) == tag;
          val[j] = __uint_as_float((unsigned)v);
        }
        if (__all(ok)) break;
        if ((spin & 127) == 127 && __hip_atomic_load(ab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) fail = true;
        if (team_wait_expired(spin, t_wait)) {
          __hip_atomic_store(ab, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          fail = true;
        }
        if (fail) break;
        __builtin_amdgcn_s_sleep(1);
      }
      s_q[tg * ATT + d] = (((val[0] + val[1]) + (val[2] + val[3])) + ((val[4] + val[5]) + (val[6] + val[7]))) +
                          (tg == 0 ? qb_d : 0.f);
      if (fail) *s_flag = 1;
    }
    __syncthreads();
    if (*s_flag) return;
    TEAM_MARK(b == 0 && a == 0, 18);
    // ---- energies of the own positions: e[pos] = sum_d v[d] tanh(q[d] + annot[pos][d] + (W_ld cf[pos])[d])
    {
      const float4* q4 = reinterpret_cast<const float4*>(s_q + 8 * dg);
      const float4* v4 = reinterpret_cast<const float4*>(s_v + 8 * dg);
      float e0 = 0.f, e1 = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 qa = q4[i], qb = q4[ATT / 4 + i], qc = q4[2 * ATT / 4 + i], qd = q4[3 * ATT / 4 + i], v = v4[i];
        const float4 q = make_float4((qa.x + qb.x) + (qc.x + qd.x), (qa.y + qb.y) + (qc.y + qd.y),
                                     (qa.z + qb.z) + (qc.z + qd.z), (qa.w + qb.w) + (qc.w + qd.w));
        e0 += v.x * fast_tanhf(q.x + pl[4 * i]) + v.z * fast_tanhf(q.z + pl[4 * i + 2]);
        e1 += v.y * fast_tanhf(q.y + pl[4 * i + 1]) + v.w * fast_tanhf(q.w + pl[4 * i + 3]);
      }
      s_ep[dg * PW + pos_l] = e0 + e1;
    }
    __syncthreads();
    TEAM_MARK(b == 0 && a == 0, 19);
    // ---- their sigmoids (zero beyond the utterance), to every replica: lane 8 r + j of wave i publishes position
    // 8 i + j to replica r
    if (wave < PW / 8) {
      const int pl_ = 8 * wave + (lane & 7);
      float e = s_ep[pl_];
#pragma unroll
      for (int k = 1; k < 16; ++k) e += s_ep[k * PW + pl_];
      const float sg = PW * a + pl_ < len ? sigmoidf_(e) : 0.f;
      publish(X + (size_t)(lane >> 3) * X_REP + X_S + cur * X_S_N, PW * a + pl_, sg, tag);
    }
    TEAM_MARK(b == 0 && a == 0, 21);
    // ---- off the serial chain: the alignment of all positions (state of the location layer; output row)
    float a_last;
    {
      float v, v2;
      const bool ok = poll<false>(X + X_S + cur * X_S_N + min(tid, TMAX - 1), tid < TMAX, nullptr, tag, &v, &v2, ab);
      if (tid < TMAX) s_s[tid] = v;
      if (!ok) *s_flag = 1;
      __syncthreads();
      if (*s_flag) return;
      float a0, a1;
      alignment_of(s_s, len - 1, &a0, &a1, &a_last);
      if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ps = lane + 64 * j;
          const float al = j ? a1 : a0;
          s_a[LPAD + ps] = al;
          s_a[(TMAX + 32) + LPAD + ps] += al;
          if (ps / PW == a && ps < T) p.align[((size_t)b * (p.max_steps + 1) + t) * T + ps] = al;
        }
      }
    }
    __syncthreads();
    TEAM_MARK(b == 0 && a == 0, 20);
    location_terms();  // for step t + 1 (unused if this was the last step)
    TEAM_MARK(b == 0 && a == 0, 16);
    // ---- the stop rule of this step, as every row workgroup evaluates it
    {
      float v, v2;
      const u64* ad = tid < RNN ? X + X_D2 + cur * 256 + tid : X + X_OUT + cur * X_OUT_N + min(tid - RNN, ODIM - 1);
      const bool ok = poll<false>(ad, tid < RNN + ODIM, nullptr, tag, &v, &v2, ab);
      s_st[tid] = v;
      if (!ok) *s_flag = 1;
      __syncthreads();
      if (*s_flag) return;
      TEAM_MARK(b == 0 && a == 0, 22);
      float xd2[4], xo[3];
#pragma unroll
      for (int j = 0; j < 4; ++j) xd2[j] = s_st[lane + 64 * j];
#pragma unroll
      for (int j = 0; j < 3; ++j) xo[j] = lane + 64 * j < ODIM ? s_st[RNN + lane + 64 * j] : 0.f;
      float token;
      if (stop_rule(ws, bs, xd2, xo, a_last, t + 1, len, p.max_steps, &token)) return;
    }
  }
}

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void t1_team_kernel(TeamP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x / TEAM, g = blockIdx.x % TEAM;
  if (g >= TG) attention_workgroup(p, b, g - TG, sm);
  else row_workgroup(p, b, g, sm);
}

constexpr size_t ROW_LDS = (size_t)RL_TOTAL * sizeof(float);
constexpr size_t ATT_LDS = (size_t)(2 * LOCK * LOCF + ATT * LOCF + 2 * ATT + 2 * (TMAX + 32) + PW * (LOCF + 1) + 4 * ATT +
                                    16 * PW + TMAX + 512 + 8) * sizeof(float);
constexpr size_t TEAM_LDS = ROW_LDS > ATT_LDS ? ROW_LDS : ATT_LDS;

// ---- packing
// GRU input weights composed with a preceding linear layer x = Wd y + bd:
// w_c[row][k] = sum_j w_ih[row][j] wd[j][k], b_c[row] = b_ih[row] + sum_j w_ih[row][j] bd[j]  (fp64, one block per row)
__global__ void compose_kernel(float* __restrict__ w_c, float* __restrict__ b_c, const float* __restrict__ w_ih,
                               const float* __restrict__ b_ih, const float* __restrict__ wd, const float* __restrict__ bd,
                               int kin, int ky) {
  const int row = blockIdx.x;
  for (int k = threadIdx.x; k < ky; k += blockDim.x) {
    double acc = 0.0;
    for (int j = 0; j < kin; ++j) acc += (double)w_ih[(size_t)row * kin + j] * (double)wd[(size_t)j * ky + k];
    w_c[(size_t)row * ky + k] = (float)acc;
  }
  if (threadIdx.x == 0) {
    double acc = b_ih[row];
    for (int j = 0; j < kin; ++j) acc += (double)w_ih[(size_t)row * kin + j] * (double)bd[j];
    b_c[row] = (float)acc;
  }
}

struct PackP {
  T1StepWeights w;
  const float *wc, *bc;  // composed GRU-1 input weights [768][512], [768]
  float *img, *w2t, *q_bias, *conv_t, *ctx_cols;
};

// a row of a [rows][K] matrix at column lane + 64 j (zero beyond K)
__device__ __forceinline__ float at(const float* W, int row, int K, int lane, int j) {
  const int k = lane + 64 * j;
  return k < K ? W[(size_t)row * K + k] : 0.f;
}

__global__ void pack_team_kernel(PackP p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)PRE0 * PRE1) {
    const int kk = (int)(i & 3), r = (int)((i >> 2) % PRE1), k4 = (int)(i / (4 * PRE1));
    p.w2t[i] = p.w.pre_w1[(size_t)r * PRE0 + 4 * k4 + kk];
  }
  if (i < ATT) p.q_bias[i] = p.w.query_b[i] + p.w.loc_lin_b[i];
  if (i < LOCF * 2 * LOCK) p.conv_t[(i % (2 * LOCK)) * LOCF + i / (2 * LOCK)] = p.w.loc_conv_w[i];
  if (i < (size_t)G_ROWS * ENC) {
    // the columns over the context of: attention GRU weight_ih, GRU 1 weight_ih composed, project_to_decoder_in
    const int k = (int)(i % ENC), row = (int)(i / ENC);
    p.ctx_cols[i] = row < G_1    ? p.w.att_w_ih[(size_t)row * (PRE1 + ENC) + PRE1 + k]
                    : row < G_D0 ? p.wc[(size_t)(row - G_1) * 2 * RNN + RNN + k]
                                 : p.w.pd_w[(size_t)(row - G_D0) * 2 * RNN + RNN + k];
  }
  if (i >= (size_t)TG * NS * NT) return;
  static_assert(ENC == RNN && PRE1 + ENC == 384, "the context is the second part of every input it is part of");
  const int tid = (int)(i % NT), s = (int)((i / NT) % NS), g = (int)(i / ((size_t)NT * NS));
  const int lane = tid & 63, wave = tid >> 6, u = 8 * g + wave;
  const int XA = PRE1 + ENC;  // attention GRU input width
  float v = 0.f;
  auto in = [&](int lo, int n) { return s >= lo && s < lo + n; };
  if (in(S_P1, 7)) v = at(p.w.pre_w0, u, MEM, lane, s - S_P1);
  else if (in(S_A_RX, 2)) v = at(p.w.att_w_ih, u, XA, lane, s - S_A_RX);  // (columns over x2; the context's: ctx_cols)
  else if (in(S_A_ZX, 2)) v = at(p.w.att_w_ih, RNN + u, XA, lane, s - S_A_ZX);
  else if (in(S_A_NX, 2)) v = at(p.w.att_w_ih, 2 * RNN + u, XA, lane, s - S_A_NX);
  else if (in(S_A_RH, 4)) v = at(p.w.att_w_hh, u, RNN, lane, s - S_A_RH);
  else if (in(S_A_ZH, 4)) v = at(p.w.att_w_hh, RNN + u, RNN, lane, s - S_A_ZH);
  else if (in(S_A_NH, 4)) v = at(p.w.att_w_hh, 2 * RNN + u, RNN, lane, s - S_A_NH);
  else if (in(S_QS, 8)) v = tid < ATT ? p.w.query_w[(size_t)tid * RNN + 8 * g + (s - S_QS)] : 0.f;
  else if (in(S_1_RC, 4)) v = at(p.wc, u, 2 * RNN, lane, s - S_1_RC);     // (columns over h_att)
  else if (in(S_1_ZC, 4)) v = at(p.wc, RNN + u, 2 * RNN, lane, s - S_1_ZC);
  else if (in(S_1_NC, 4)) v = at(p.wc, 2 * RNN + u, 2 * RNN, lane, s - S_1_NC);
  else if (in(S_1_RH, 4)) v = at(p.w.rnn_w_hh[0], u, RNN, lane, s - S_1_RH);
  else if (in(S_1_ZH, 4)) v = at(p.w.rnn_w_hh[0], RNN + u, RNN, lane, s - S_1_ZH);
  else if (in(S_1_NH, 4)) v = at(p.w.rnn_w_hh[0], 2 * RNN + u, RNN, lane, s - S_1_NH);
  else if (in(S_D0, 4)) v = at(p.w.pd_w, u, 2 * RNN, lane, s - S_D0);
  else if (in(S_2_RX, 4)) v = at(p.w.rnn_w_ih[1], u, RNN, lane, s - S_2_RX);
  else if (in(S_2_ZX, 4)) v = at(p.w.rnn_w_ih[1], RNN + u, RNN, lane, s - S_2_ZX);
  else if (in(S_2_NX, 4)) v = at(p.w.rnn_w_ih[1], 2 * RNN + u, RNN, lane, s - S_2_NX);
  else if (in(S_2_RH, 4)) v = at(p.w.rnn_w_hh[1], u, RNN, lane, s - S_2_RH);
  else if (in(S_2_ZH, 4)) v = at(p.w.rnn_w_hh[1], RNN + u, RNN, lane, s - S_2_ZH);
  else if (in(S_2_NH, 4)) v = at(p.w.rnn_w_hh[1], 2 * RNN + u, RNN, lane, s - S_2_NH);
  else if (in(S_MEL, 4)) v = wave < 5 ? at(p.w.mel_w, 5 * g + wave, RNN, lane, s - S_MEL) : 0.f;
  else if (in(S_STOP, 7)) v = at(p.w.stop_w, 0, RNN + ODIM, lane, s - S_STOP);
  else if (s == S_B_P1) v = p.w.pre_b0[u];
  else if (s == S_B_A) v = p.w.att_b_ih[u] + p.w.att_b_hh[u];
  else if (s == S_B_A + 1) v = p.w.att_b_ih[RNN + u] + p.w.att_b_hh[RNN + u];
  else if (s == S_B_A + 2) v = p.w.att_b_ih[2 * RNN + u];
  else if (s == S_B_A + 3) v = p.w.att_b_hh[2 * RNN + u];
  else if (s == S_B_1) v = p.bc[u] + p.w.rnn_b_hh[0][u];
  else if (s == S_B_1 + 1) v = p.bc[RNN + u] + p.w.rnn_b_hh[0][RNN + u];
  else if (s == S_B_1 + 2) v = p.bc[2 * RNN + u];
  else if (s == S_B_1 + 3) v = p.w.rnn_b_hh[0][2 * RNN + u];
  else if (s == S_B_1 + 4) v = p.w.pd_b[u];
  else if (s == S_B_2) v = p.w.rnn_b_ih[1][u] + p.w.rnn_b_hh[1][u];
  else if (s == S_B_2 + 1) v = p.w.rnn_b_ih[1][RNN + u] + p.w.rnn_b_hh[1][RNN + u];
  else if (s == S_B_2 + 2) v = p.w.rnn_b_ih[1][2 * RNN + u];
  else if (s == S_B_2 + 3) v = p.w.rnn_b_hh[1][2 * RNN + u];
  else if (s == S_B_MEL) v = wave < 5 ? p.w.mel_b[5 * g + wave] : 0.f;
  else if (s == S_B_STOP) v = p.w.stop_b[0];
  p.img[i] = v;
}

}  // namespace

struct t1_team {
  float *img, *w2t, *q_bias, *conv_t;
  float *ctx_cols, *ctx_g;  // [G_ROWS][256] context columns; [2][G_ROWS][TMAX] their products with the utterances' encoder outputs
  const float *b2, *mem_init, *att_init, *rnn_inits, *loc_lin_w, *v_w, *stop_w, *stop_b;  // borrowed from the owner
  u64* xch;
  int cap_B;
  int* abort_word;       // device
};

int t1_team_create(const T1StepWeights* w, hipStream_t st, t1_team** out) {
  TTSX_ARG(w && out, "t1_team_create: null argument");
  *out = nullptr;
  if (w->mem_dim != MEM || w->odim != ODIM) return TTSX_OK;
  {
    // every workgroup of a team needs a CU of its own at the same time (two utterances: two teams)
    int dev = 0, cus = 0;
    TTSX_HIP(hipGetDevice(&dev));
    TTSX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus < 2 * TEAM) return TTSX_OK;
  }
  t1_team* t = new t1_team();
  memset(t, 0, sizeof(*t));
  float *wc = nullptr, *bc = nullptr;
  ScopeExit on_fail([&] { t1_team_destroy(t); });
  ScopeExit free_tmp([&] { (void)hipFree(wc), (void)hipFree(bc); });
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&wc), sizeof(float) * 3 * RNN * 2 * RNN));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&bc), sizeof(float) * 3 * RNN));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&t->img), sizeof(float) * (size_t)TG * NS * NT));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&t->w2t), sizeof(float) * PRE0 * PRE1));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&t->q_bias), sizeof(float) * ATT));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&t->conv_t), sizeof(float) * LOCF * 2 * LOCK));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&t->ctx_cols), sizeof(float) * (size_t)G_ROWS * ENC));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&t->ctx_g), sizeof(float) * (size_t)2 * G_ROWS * TMAX));
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&t->abort_word), sizeof(int)));
  compose_kernel<<<3 * RNN, 256, 0, st>>>(wc, bc, w->rnn_w_ih[0], w->rnn_b_ih[0], w->pd_w, w->pd_b, RNN, 2 * RNN);
  TTSX_LAUNCH_CHECK();
  PackP pp{*w, wc, bc, t->img, t->w2t, t->q_bias, t->conv_t, t->ctx_cols};
  const size_t n = std::max((size_t)TG * NS * NT, (size_t)G_ROWS * ENC);
  pack_team_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(pp);
  TTSX_LAUNCH_CHECK();
  t->b2 = w->pre_b1, t->mem_init = w->mem_init, t->att_init = w->att_init, t->rnn_inits = w->rnn_inits;
  t->loc_lin_w = w->loc_lin_w, t->v_w = w->v_w, t->stop_w = w->stop_w, t->stop_b = w->stop_b;
  TTSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(t1_team_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)TEAM_LDS));
  TTSX_HIP(hipStreamSynchronize(st));
  on_fail.dismiss();
  *out = t;
  return TTSX_OK;
}

void t1_team_destroy(t1_team* t) {
  if (!t) return;
  float* bufs[] = {t->img, t->w2t, t->q_bias, t->conv_t, t->ctx_cols, t->ctx_g};
  for (float* b : bufs)
    if (b) (void)hipFree(b);
  if (t->xch) (void)hipFree(t->xch);
  if (t->abort_word) (void)hipFree(t->abort_word);
  delete t;
}

#ifdef TTSX_TEAM_TIMING
extern "C" int ttsx_debug_team_timing(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(team_marks), sizeof(long long) * (64 + 4 * 64)) == hipSuccess ? 0 : -1;
}
#endif

bool t1_team_supports(const t1_team* t, int B, int T) { return t && B >= 1 && B <= 2 && T >= 1 && T <= TMAX; }

int t1_team_run(t1_team* t, const float* enc, const float* annot, const int32_t* lengths, int B, int T,
                int max_decoder_steps, float* mel, float* align, float* stop, int32_t* n_steps, int* steps_run_host,
                hipStream_t st) {
  TTSX_ARG(t && enc && annot && mel && align && stop && n_steps, "t1_team_run: null argument");
  TTSX_ARG(t1_team_supports(t, B, T), "t1_team_run: B=%d T=%d outside what the team kernel covers", B, T);
  if (t->cap_B < B) {
    if (t->xch) (void)hipFree(t->xch);
    t->xch = nullptr, t->cap_B = 0;
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&t->xch), sizeof(u64) * (size_t)B * X_TEAM));
    t->cap_B = B;
  }
  std::lock_guard<std::mutex> one_team(ttsx_team_mutex());  // until the synchronisation below
  // tags start at 1: all-zero words match no step
  TTSX_HIP(hipMemsetAsync(t->xch, 0, sizeof(u64) * (size_t)B * X_TEAM, st));
  TTSX_HIP(hipMemsetAsync(t->abort_word, ttsx_take_wait_failure(TTSX_WAIT_T1_TEAM) ? 1 : 0, sizeof(int), st));
  TTSX_HIP(hipMemsetAsync(n_steps, 0, sizeof(int32_t) * B, st));
  TeamP p{};
  p.img = t->img, p.w2t = t->w2t, p.b2 = t->b2, p.mem_init = t->mem_init, p.att_init = t->att_init;
  p.rnn_inits = t->rnn_inits, p.loc_conv_t = t->conv_t, p.loc_lin_w = t->loc_lin_w, p.q_bias = t->q_bias, p.v_w = t->v_w;
  p.stop_w = t->stop_w, p.stop_b = t->stop_b, p.annot = annot, p.ctx_g = t->ctx_g, p.len = lengths, p.xch = t->xch;
  p.abort_word = t->abort_word, p.mel = mel, p.align = align, p.stop = stop, p.n_steps = n_steps, p.T = T;
  p.max_steps = max_decoder_steps;
  static_assert(G_ROWS % 64 == 0 && TMAX % 64 == 0, "whole blocks of the products kernel");
  ctx_products_kernel<ENC><<<dim3(G_ROWS / 64, TMAX / 64, B), 256, 0, st>>>(t->ctx_cols, G_ROWS, ENC, enc, lengths, 1, T, TMAX,
                                                                            t->ctx_g);
  TTSX_LAUNCH_CHECK();
  t1_team_kernel<<<B * TEAM, NT, TEAM_LDS, st>>>(p);
  TTSX_LAUNCH_CHECK();
  std::vector<int32_t> host(B + 1);
  TTSX_HIP(hipMemcpyAsync(host.data(), n_steps, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipMemcpyAsync(host.data() + B, t->abort_word, sizeof(int), hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipStreamSynchronize(st));
  if (host[B]) {
    ttsx_set_error("t1_team_run: a workgroup of the decoder team waited too long for its inputs (team not co-resident?)");
    return TTSX_ERR_STATE;
  }
  int run = 0;
  for (int b = 0; b < B; ++b) run = std::max(run, (int)host[b]);
  if (steps_run_host) *steps_run_host = run;
  return TTSX_OK;
}
