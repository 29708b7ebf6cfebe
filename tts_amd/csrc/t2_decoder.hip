// Tacotron2 decoder for gfx950: row-block fp32-MFMA GEMMs (LSTM gates, query, projection),
// location-sensitive attention, prenet + stopnet, and the greedy loop around them.
//
// What it computes per step follows Decoder.decode (Tacotron2/TTS/models.py:578-617):
//   x2  = prenet(memory)                        K1
//   h_a = LSTMCell([x2; ctx], h_a, c_a)         K2
//   q   = W_q h_a                               K3
//   a   = norm(v . tanh(q + loc(a_prev, a_cum) + P))   K4-K6
//   ctx = a . enc ; a_cum += a                  K7
//   h_d = LSTMCell([h_a; ctx], h_d, c_d)        K8
//   o   = W_o [h_d; ctx] + b_o                  K9
//   stop = sigmoid(w_s [h_d; o] + b_s)          K10
//
// Data layout.  Weights are packed once: 16 output rows form a row block; inside a row
// block the K axis is cut in groups of 16 and each group is stored as 64 lanes x 4 floats
// in exactly the order mfma_f32_16x16x4f32 wants its A operand over four consecutive
// issues (lane l: row l&15, k = 4*(l>>4) + s for issue s).  A wave therefore streams its
// weights with one fully coalesced 1 KiB load per 16 k.  LSTM gate rows are interleaved so
// that row 4u+g of a block is gate g (i,f,g,o) of hidden unit u: after the MFMA the four
// accumulator registers of a lane are the four gates of one (unit, utterance) pair and the
// cell update is lane-local.  Activations are quad-interleaved (see qidx in common.h).
#include <algorithm>
#include <vector>

#include "common.h"

namespace {

struct Ctrl {
  int base;      // decoder steps completed before the current chunk (state parity, t == 0 test)
  int out_base;  // output row of the chunk's first step
  int all_done;  // set when every utterance has emitted its stop frame (free-running mode)
  int n_done;
};

constexpr int ENC_DIM = 512, QUERY_DIM = 1024, DEC_DIM = 1024, PRENET_DIM = 256, ATT_DIM = 128;
constexpr int MEL_DIM = 80, LOC_F = 32, LOC_K = 31;
constexpr int GEMM_THREADS = 512, GEMM_WAVES = GEMM_THREADS / 64;

// ------------------------------------------------------------------ weight packing
// dst[((rb*KG + kg)*64 + lane)*4 + s] = src_row(rb*16 + (lane&15))[kg*16 + 4*(lane>>4) + s]
// where the source row is the concatenation [w1 row | w2 row], optionally scaled per row.
// lstm_hidden > 0 selects gate interleaving: packed row p of block rb <- source row
// (p&3)*hidden + 4*rb + (p>>2).
__global__ void pack_rows_kernel(float* __restrict__ dst, const float* __restrict__ w1, int k1,
                                 const float* __restrict__ w2, int k2, int rows, int KG, int lstm_hidden,
                                 const float* __restrict__ row_scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)ceil_div(rows, 16) * KG * 256;
  if (i >= total) return;
  int s = i & 3, lane = (i >> 2) & 63;
  size_t g = i >> 8;
  int kg = g % KG, rb = g / KG;
  int p = lane & 15;
  int k = kg * 16 + 4 * (lane >> 4) + s;
  int row = lstm_hidden > 0 ? (p & 3) * lstm_hidden + 4 * rb + (p >> 2) : rb * 16 + p;
  float v = 0.f;
  if (row < rows) {
    if (k < k1) v = w1[(size_t)row * k1 + k];
    else if (k - k1 < k2) v = w2[(size_t)row * k2 + (k - k1)];
    if (row_scale) v *= row_scale[row];
  }
  dst[i] = v;
}

__global__ void pack_bias_kernel(float* __restrict__ dst, const float* __restrict__ b1,
                                 const float* __restrict__ b2, int rows, int lstm_hidden) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int padded = round_up(rows, 16);
  if (i >= padded) return;
  int rb = i >> 4, p = i & 15;
  int row = lstm_hidden > 0 ? (p & 3) * lstm_hidden + 4 * rb + (p >> 2) : i;
  float v = 0.f;
  if (row < rows) {
    if (b1) v += b1[row];
    if (b2) v += b2[row];
  }
  dst[i] = v;
}

// Prenet layer with eval-mode BatchNorm folded in: wt[k][n] = w[n][k]*g[n]/sqrt(var[n]+eps),
// bias[n] = beta[n] - mean[n]*g[n]/sqrt(var[n]+eps); stored transposed so that thread n reads
// consecutive addresses.  Without BatchNorm ("original" prenet) scale = 1, bias = 0.
__global__ void pack_prenet_kernel(float* __restrict__ wt, float* __restrict__ bias, const float* __restrict__ w,
                                   const float* __restrict__ g, const float* __restrict__ beta,
                                   const float* __restrict__ mean, const float* __restrict__ var, int n_out,
                                   int n_in) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out * n_in) return;
  int n = i / n_in, k = i % n_in;
  float sc = 1.f;
  if (g) sc = g[n] / sqrtf(var[n] + 1e-5f);
  wt[(size_t)k * n_out + n] = w[i] * sc;
  if (k == 0) bias[n] = g ? beta[n] - mean[n] * sc : 0.f;
}

// ------------------------------------------------------------------ row-block GEMM
struct Seg {
  const float* x[2];  // quad-interleaved activation buffer(s)
  int sel;            // 0: x[0]; 1: x[step&1] (state before this step); 2: x[(step+1)&1] (new state)
  int kgroups;        // K of the segment / 16
};

enum { MODE_LINEAR = 0, MODE_LSTM = 1, MODE_PROJ = 2 };

struct GemmP {
  const float* Wp;
  const float* bias;
  Seg seg[3];
  int nseg, KG, Bpad, B;
  float* out[2];  // LINEAR/PROJ: out[0] quad layout; LSTM: h ping-pong pair, written at (step+1)&1
  float* c;       // LSTM cell state, quad layout
  float* mel;     // PROJ: [B, mel_rows, 80]
  int r, rows, mel_rows;
  const Ctrl* ctrl;
  int step_in_chunk;
};

template <int NT, int MODE>
__global__ __launch_bounds__(GEMM_THREADS) void rowblock_gemm_kernel(GemmP p) {
  if (p.ctrl->all_done) return;
  const int step = p.ctrl->base + p.step_in_chunk;
  const int rb = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b0 = blockIdx.y * (NT * 16);
  __shared__ f32x4 red[GEMM_WAVES][NT][64];

  f32x4 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float4* W = reinterpret_cast<const float4*>(p.Wp) + (size_t)rb * p.KG * 64 + lane;
  const int col = b0 + (lane & 15);
  const int kq = lane >> 4;
  int kg_base = 0;
  for (int s = 0; s < p.nseg; ++s) {
    const Seg sg = p.seg[s];
    const float* xb = sg.sel == 0 ? sg.x[0] : sg.x[(step + (sg.sel == 2)) & 1];
    const float4* X = reinterpret_cast<const float4*>(xb) + (size_t)kq * p.Bpad + col;
#pragma unroll 4
    for (int kg = wave; kg < sg.kgroups; kg += GEMM_WAVES) {
      const float4 wv = W[(size_t)(kg_base + kg) * 64];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float4 xv = X[(size_t)kg * 4 * p.Bpad + n * 16];
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, acc[n], 0, 0, 0);
      }
    }
    kg_base += sg.kgroups;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) red[wave][n][lane] = acc[n];
  __syncthreads();

  // wave w finishes n-tiles w, w+8, ...: lane l holds rows 4*(l>>4)+{0..3} of column l&15
  for (int n = wave; n < NT; n += GEMM_WAVES) {
    f32x4 v = red[0][n][lane];
#pragma unroll
    for (int w = 1; w < GEMM_WAVES; ++w) v += red[w][n][lane];
    const int b = b0 + n * 16 + (lane & 15);
    const int quad = rb * 4 + (lane >> 4);  // output rows 4*quad .. 4*quad+3
    const float4 bias = reinterpret_cast<const float4*>(p.bias)[quad];
    v[0] += bias.x, v[1] += bias.y, v[2] += bias.z, v[3] += bias.w;
    if (MODE == MODE_LSTM) {
      // rows of this lane = gates i,f,g,o of hidden unit `quad`
      const size_t ci = qidx(quad, b, p.Bpad);
      const float c_old = p.c[ci];
      const float c_new = sigmoidf_(v[1]) * c_old + sigmoidf_(v[0]) * tanhf(v[2]);
      const float h_new = sigmoidf_(v[3]) * tanhf(c_new);
      p.c[ci] = c_new;
      p.out[(step + 1) & 1][ci] = h_new;
    } else {
      reinterpret_cast<float4*>(p.out[0])[(size_t)quad * p.Bpad + b] = make_float4(v[0], v[1], v[2], v[3]);
      if (MODE == MODE_PROJ && b < p.B && quad * 4 < p.r * MEL_DIM) {
        // frames of this step, frame-major [B, mel_rows, 80]: o[j] -> frame j/80, bin j%80
        const int row0 = (p.ctrl->out_base + p.step_in_chunk) * p.r;
        float* dst = p.mel + ((size_t)b * p.mel_rows + row0) * MEL_DIM + quad * 4;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

// ------------------------------------------------------------------ prenet + stopnet
struct PrenetP {
  const float* w1t;  // [80][256]
  const float* b1;
  const float* w2t;  // [256][256]
  const float* b2;
  const float* stop_w;  // [1024 + 80*r_init]
  const float* stop_b;
  const float* h_d[2];  // quad
  const float* o;       // quad [80*r_init]
  const float* mem_override;  // [B,80] row-major or null
  float* x2;            // quad [256]
  float* stop_out;      // [B, stop_rows]
  int32_t* n_steps;     // [B]
  Ctrl* ctrl;
  int step_in_chunk, Bpad, B, r, r_init, stop_rows;
  int do_stop;    // evaluate the stop token of step-1 (needs step > 0)
  int do_prenet;  // compute x2 for `step`
  int free_running;
  float threshold;
};

__global__ __launch_bounds__(256) void prenet_stop_kernel(PrenetP p) {
  if (p.ctrl->all_done) return;
  const int step = p.ctrl->base + p.step_in_chunk;
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ float sm[MEL_DIM];
  __shared__ float sx1[PRENET_DIM];
  __shared__ float sred[4];
  const int odim = MEL_DIM * p.r_init;

  if (p.do_stop && step > 0) {
    // stop token of the step that just finished: sigmoid(w_s . [h_d; o] + b_s)
    const float* hd = p.h_d[step & 1];
    float acc = 0.f;
    for (int k = tid; k < DEC_DIM; k += 256) acc += p.stop_w[k] * hd[qidx(k, b, p.Bpad)];
    for (int k = tid; k < odim; k += 256) acc += p.stop_w[DEC_DIM + k] * p.o[qidx(k, b, p.Bpad)];
    acc = wave_sum(acc);
    if ((tid & 63) == 0) sred[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      const float s = sigmoidf_(sred[0] + sred[1] + sred[2] + sred[3] + p.stop_b[0]);
      const int row = p.ctrl->out_base + p.step_in_chunk - 1;
      p.stop_out[(size_t)b * p.stop_rows + row] = s;
      if (s > p.threshold && p.n_steps[b] == 0) {
        p.n_steps[b] = step;  // steps run so far, the stop frame included
        const int done = atomicAdd(&p.ctrl->n_done, 1) + 1;
        if (p.free_running && done == p.B) p.ctrl->all_done = 1;
      }
    }
  }
  if (!p.do_prenet) return;

  // memory = last of the r frames emitted by the previous step (zeros before the first)
  if (tid < MEL_DIM) {
    float m = 0.f;
    if (p.mem_override) m = p.mem_override[b * MEL_DIM + tid];
    else if (step > 0) m = p.o[qidx(MEL_DIM * (p.r - 1) + tid, b, p.Bpad)];
    sm[tid] = m;
  }
  __syncthreads();
  float a1 = p.b1[tid];
#pragma unroll 8
  for (int k = 0; k < MEL_DIM; ++k) a1 += p.w1t[k * PRENET_DIM + tid] * sm[k];
  sx1[tid] = fmaxf(a1, 0.f);
  __syncthreads();
  float a2 = p.b2[tid];
#pragma unroll 8
  for (int k = 0; k < PRENET_DIM; ++k) a2 += p.w2t[k * PRENET_DIM + tid] * sx1[k];
  p.x2[qidx(tid, b, p.Bpad)] = fmaxf(a2, 0.f);
}

// ------------------------------------------------------------------ attention
struct AttnP {
  const float* q;        // quad [128]
  const float* P;        // [B][128][T]  processed_inputs, transposed
  const float* enc;      // [B][T][512]
  const int32_t* enc_len;
  const float* loc_conv_w;   // [32][2][31]
  const float* loc_dense_w;  // [128][32]
  const float* v_w;          // [128]
  const float* v_b;          // [1]
  float* a_prev;         // [B][T]
  float* a_cum;          // [B][T]
  float* ctx;            // quad [512]
  float* align_out;      // [B][align_rows][T]
  const Ctrl* ctrl;
  int step_in_chunk, Bpad, T, align_rows, norm;
};

constexpr int ATT_THREADS = 512, ATT_TT = 128, ATT_PAD = (LOC_K - 1) / 2;

__device__ inline float block_reduce_sum(float v, float* sred, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) sred[tid >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < ATT_THREADS / 64; ++w) t += sred[w];
  return t;
}
__device__ inline float block_reduce_max(float v, float* sred, int tid) {
  v = wave_max(v);
  __syncthreads();
  if ((tid & 63) == 0) sred[tid >> 6] = v;
  __syncthreads();
  float t = sred[0];
#pragma unroll
  for (int w = 1; w < ATT_THREADS / 64; ++w) t = fmaxf(t, sred[w]);
  return t;
}

__global__ __launch_bounds__(ATT_THREADS) void attention_kernel(AttnP p) {
  if (p.ctrl->all_done) return;
  const int b = blockIdx.x, tid = threadIdx.x, T = p.T;
  const int len = p.enc_len ? p.enc_len[b] : T;
  extern __shared__ float smem[];
  float* s_prev = smem;                      // [T + 30], zero padded by 15 each side
  float* s_cum = s_prev + (T + 2 * ATT_PAD);  // [T + 30]
  float* s_e = s_cum + (T + 2 * ATT_PAD);     // [T]
  float* s_cf = s_e + T;                      // [128][33]
  float* s_part = s_cf + ATT_TT * (LOC_F + 1);  // [4][128]
  float* s_q = s_part + 4 * ATT_TT;             // [128]
  float* s_red = s_q + ATT_DIM;                 // [8]

  for (int i = tid; i < T + 2 * ATT_PAD; i += ATT_THREADS) {
    const int t = i - ATT_PAD;
    const bool in = t >= 0 && t < T;
    s_prev[i] = in ? p.a_prev[(size_t)b * T + t] : 0.f;
    s_cum[i] = in ? p.a_cum[(size_t)b * T + t] : 0.f;
  }
  if (tid < ATT_DIM) s_q[tid] = p.q[qidx(tid, b, p.Bpad)];
  __syncthreads();

  const int tl = tid & (ATT_TT - 1);  // position inside the tile
  const int grp = tid >> 7;           // 0..3, wave-uniform
  const float vb = p.v_b[0];
  for (int t0 = 0; t0 < T; t0 += ATT_TT) {
    const int t = t0 + tl;
    // location conv: filters grp*8 .. grp*8+7 at position t
    {
      float win[2][LOC_K];
#pragma unroll
      for (int j = 0; j < LOC_K; ++j) {
        const int i = min(t + j, T + 2 * ATT_PAD - 1);
        win[0][j] = s_prev[i];
        win[1][j] = s_cum[i];
      }
#pragma unroll
      for (int ff = 0; ff < 8; ++ff) {
        const int f = grp * 8 + ff;
        const float* w = p.loc_conv_w + f * 2 * LOC_K;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < LOC_K; ++j) acc += w[j] * win[0][j];
#pragma unroll
        for (int j = 0; j < LOC_K; ++j) acc += w[LOC_K + j] * win[1][j];
        s_cf[tl * (LOC_F + 1) + f] = acc;
      }
    }
    __syncthreads();
    // energies: attention dims grp*32 .. grp*32+31 at position t
    {
      float cf[LOC_F];
#pragma unroll
      for (int f = 0; f < LOC_F; ++f) cf[f] = s_cf[tl * (LOC_F + 1) + f];
      float part = 0.f;
      const bool valid = t < T;
      for (int dd = 0; dd < 32; ++dd) {
        const int d = grp * 32 + dd;
        const float* w = p.loc_dense_w + d * LOC_F;
        float loc = 0.f;
#pragma unroll
        for (int f = 0; f < LOC_F; ++f) loc += w[f] * cf[f];
        const float pin = valid ? p.P[((size_t)b * ATT_DIM + d) * T + t] : 0.f;
        part += p.v_w[d] * tanhf(s_q[d] + loc + pin);
      }
      s_part[grp * ATT_TT + tl] = part;
    }
    __syncthreads();
    if (tid < ATT_TT && t0 + tid < T) {
      const float e = s_part[tid] + s_part[ATT_TT + tid] + s_part[2 * ATT_TT + tid] + s_part[3 * ATT_TT + tid] + vb;
      s_e[t0 + tid] = (t0 + tid < len) ? e : -INFINITY;
    }
    __syncthreads();
  }

  // normalise over the encoder axis
  if (p.norm == TTSX_NORM_SOFTMAX) {
    float m = -INFINITY;
    for (int t = tid; t < T; t += ATT_THREADS) m = fmaxf(m, s_e[t]);
    m = block_reduce_max(m, s_red, tid);
    float sum = 0.f;
    for (int t = tid; t < T; t += ATT_THREADS) {
      const float e = expf(s_e[t] - m);
      s_e[t] = e;
      sum += e;
    }
    sum = block_reduce_sum(sum, s_red, tid);
    for (int t = tid; t < T; t += ATT_THREADS) s_e[t] = s_e[t] / sum;
  } else {
    float sum = 0.f;
    for (int t = tid; t < T; t += ATT_THREADS) {
      const float s = sigmoidf_(s_e[t]);
      s_e[t] = s;
      sum += s;
    }
    sum = block_reduce_sum(sum, s_red, tid);
    for (int t = tid; t < T; t += ATT_THREADS) s_e[t] = s_e[t] / sum;
  }
  __syncthreads();

  const int row = p.ctrl->out_base + p.step_in_chunk;
  for (int t = tid; t < T; t += ATT_THREADS) {
    const float a = s_e[t];
    p.a_prev[(size_t)b * T + t] = a;
    p.a_cum[(size_t)b * T + t] = s_cum[t + ATT_PAD] + a;
    p.align_out[((size_t)b * p.align_rows + row) * T + t] = a;
  }
  // context = a . enc : thread c owns column c
  {
    const float* e = p.enc + (size_t)b * T * ENC_DIM + tid;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int t = 0;
    for (; t + 4 <= len; t += 4) {
      acc0 += s_e[t] * e[(size_t)t * ENC_DIM];
      acc1 += s_e[t + 1] * e[(size_t)(t + 1) * ENC_DIM];
      acc2 += s_e[t + 2] * e[(size_t)(t + 2) * ENC_DIM];
      acc3 += s_e[t + 3] * e[(size_t)(t + 3) * ENC_DIM];
    }
    for (; t < len; ++t) acc0 += s_e[t] * e[(size_t)t * ENC_DIM];
    p.ctx[qidx(tid, b, p.Bpad)] = (acc0 + acc1) + (acc2 + acc3);
  }
}

// processed_inputs, transposed: P[b][d][t] = sum_c enc[b][t][c] * w_in[d][c]
// (OriginalAttention.preprocess_inputs, once per utterance).
__global__ __launch_bounds__(256) void inputs_proj_kernel(float* __restrict__ P, const float* __restrict__ enc,
                                                          const float* __restrict__ w_in, int T) {
  // block: 16 positions of one utterance; thread (tl = tid&15, dg = tid>>4) -> dims dg*8..+7
  const int b = blockIdx.y, t0 = blockIdx.x * 16, tid = threadIdx.x;
  __shared__ float s_enc[16][ENC_DIM + 1];
  for (int i = tid; i < 16 * ENC_DIM; i += 256) {
    const int tt = i / ENC_DIM, c = i % ENC_DIM;
    s_enc[tt][c] = (t0 + tt < T) ? enc[((size_t)b * T + t0 + tt) * ENC_DIM + c] : 0.f;
  }
  __syncthreads();
  const int tl = tid & 15, dg = tid >> 4;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int c = 0; c < ENC_DIM; ++c) {
    const float x = s_enc[tl][c];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += w_in[(dg * 8 + i) * ENC_DIM + c] * x;
  }
  if (t0 + tl < T) {
#pragma unroll
    for (int i = 0; i < 8; ++i) P[((size_t)b * ATT_DIM + dg * 8 + i) * T + t0 + tl] = acc[i];
  }
}

// ------------------------------------------------------------------ small utility kernels
__global__ void ctrl_set_kernel(Ctrl* c, int base, int out_base, int reset_done) {
  c->base = base;
  c->out_base = out_base;
  if (reset_done) c->all_done = 0, c->n_done = 0;
}
__global__ void ctrl_advance_kernel(Ctrl* c, int n) {
  if (c->all_done) return;
  c->base += n;
  c->out_base += n;
}
__global__ void finalize_steps_kernel(int32_t* n_steps, int B, const Ctrl* c, int steps_launched) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B && n_steps[b] == 0) n_steps[b] = steps_launched;
}
// row-major [B][K] <-> quad-interleaved
__global__ void to_quad_kernel(float* __restrict__ dst, const float* __restrict__ src, int B, int K, int Bpad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * K) dst[qidx(i % K, i / K, Bpad)] = src[i];
}
__global__ void from_quad_kernel(float* __restrict__ dst, const float* __restrict__ src, int B, int K, int Bpad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * K) dst[i] = src[qidx(i % K, i / K, Bpad)];
}

}  // namespace

// ------------------------------------------------------------------ host side
struct ttsx_t2_decoder {
  int r_init, norm, odim;
  // packed weights
  float *w_lstm1, *b_lstm1, *w_lstm2, *b_lstm2, *w_query, *b_query, *w_proj, *b_proj;
  float *pre_w1t, *pre_b1, *pre_w2t, *pre_b2;
  float *stop_w, *stop_b, *inputs_w, *v_w, *v_b, *loc_conv_w, *loc_dense_w;
  // per-run workspace
  int B, T, Bpad, cap_B, cap_T;
  const float* enc;
  const int32_t* enc_len;
  float *x2, *ctx, *h_a[2], *c_a, *h_d[2], *c_d, *q, *o, *a_prev, *a_cum, *P, *mem_tmp;
  int32_t* n_steps_tmp;
  float* stop_tmp;
  Ctrl* ctrl;
  int* host_flag;  // pinned
  int steps_done;  // host mirror of ctrl.base
  bool begun;
};

namespace {

template <typename T>
int dev_alloc(T** p, size_t n) {
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  return TTSX_OK;
}

int copy_dev(float** dst, const float* src, size_t n, hipStream_t st) {
  int rc = dev_alloc(dst, n);
  if (rc) return rc;
  TTSX_HIP(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  return TTSX_OK;
}

int pack_layer(float** wp, float** bp, const float* w1, int k1, const float* w2, int k2, const float* b1,
               const float* b2, int rows, int lstm_hidden, hipStream_t st) {
  const int KG = (k1 + k2) / 16, nrb = ceil_div(rows, 16);
  const size_t n = (size_t)nrb * KG * 256;
  int rc = dev_alloc(wp, n);
  if (rc) return rc;
  rc = dev_alloc(bp, (size_t)nrb * 16);
  if (rc) return rc;
  pack_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(*wp, w1, k1, w2, k2, rows, KG, lstm_hidden, nullptr);
  TTSX_LAUNCH_CHECK();
  pack_bias_kernel<<<ceil_div(nrb * 16, 256), 256, 0, st>>>(*bp, b1, b2, rows, lstm_hidden);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

void free_workspace(ttsx_t2_decoder* d) {
  float** bufs[] = {&d->x2, &d->ctx, &d->h_a[0], &d->h_a[1], &d->c_a, &d->h_d[0], &d->h_d[1], &d->c_d,
                    &d->q,  &d->o,   &d->a_prev, &d->a_cum,  &d->P,   &d->mem_tmp, &d->stop_tmp};
  for (float** b : bufs) {
    if (*b) (void)hipFree(*b);
    *b = nullptr;
  }
  if (d->n_steps_tmp) (void)hipFree(d->n_steps_tmp);
  d->n_steps_tmp = nullptr;
  d->cap_B = d->cap_T = 0;
}

int ensure_workspace(ttsx_t2_decoder* d, int B, int T) {
  const int Bpad = B <= 64 ? round_up(B, 16) : round_up(B, 64);
  if (d->cap_B >= Bpad && d->cap_T >= T) {
    d->Bpad = Bpad;
    return TTSX_OK;
  }
  free_workspace(d);
  const int cb = Bpad, ct = T;
  int rc = 0;
  rc |= dev_alloc(&d->x2, (size_t)PRENET_DIM * cb);
  rc |= dev_alloc(&d->ctx, (size_t)ENC_DIM * cb);
  for (int i = 0; i < 2; ++i) {
    rc |= dev_alloc(&d->h_a[i], (size_t)QUERY_DIM * cb);
    rc |= dev_alloc(&d->h_d[i], (size_t)DEC_DIM * cb);
  }
  rc |= dev_alloc(&d->c_a, (size_t)QUERY_DIM * cb);
  rc |= dev_alloc(&d->c_d, (size_t)DEC_DIM * cb);
  rc |= dev_alloc(&d->q, (size_t)ATT_DIM * cb);
  rc |= dev_alloc(&d->o, (size_t)round_up(d->odim, 16) * cb);
  rc |= dev_alloc(&d->a_prev, (size_t)cb * ct);
  rc |= dev_alloc(&d->a_cum, (size_t)cb * ct);
  rc |= dev_alloc(&d->P, (size_t)cb * ATT_DIM * ct);
  rc |= dev_alloc(&d->mem_tmp, (size_t)cb * MEL_DIM);
  rc |= dev_alloc(&d->stop_tmp, (size_t)cb);
  rc |= dev_alloc(&d->n_steps_tmp, (size_t)cb);
  if (rc) return TTSX_ERR_HIP;
  d->cap_B = cb, d->cap_T = ct, d->Bpad = Bpad;
  return TTSX_OK;
}

size_t attn_smem_bytes(int T) {
  return sizeof(float) * (size_t)(2 * (T + 2 * ATT_PAD) + T + ATT_TT * (LOC_F + 1) + 4 * ATT_TT + ATT_DIM + 8);
}

template <int MODE>
int launch_gemm(const GemmP& p, int nrb, hipStream_t st) {
  const int nt_total = p.Bpad / 16;
  if (nt_total <= 4) {
    dim3 grid(nrb, 1);
    switch (nt_total) {
      case 1: rowblock_gemm_kernel<1, MODE><<<grid, GEMM_THREADS, 0, st>>>(p); break;
      case 2: rowblock_gemm_kernel<2, MODE><<<grid, GEMM_THREADS, 0, st>>>(p); break;
      case 3: rowblock_gemm_kernel<3, MODE><<<grid, GEMM_THREADS, 0, st>>>(p); break;
      default: rowblock_gemm_kernel<4, MODE><<<grid, GEMM_THREADS, 0, st>>>(p); break;
    }
  } else {
    dim3 grid(nrb, p.Bpad / 64);
    rowblock_gemm_kernel<4, MODE><<<grid, GEMM_THREADS, 0, st>>>(p);
  }
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

struct StepOut {
  float* mel;
  int mel_rows;
  float* align;
  int align_rows;
  float* stop;
  int stop_rows;
  int32_t* n_steps;
};

// Enqueue one decoder step.  `i` is the step's index inside the current chunk.
int enqueue_step(ttsx_t2_decoder* d, int i, int r, const StepOut& o, const float* mem_override, bool stop_prev,
                 bool free_running, float thr, hipStream_t st) {
  const int Bpad = d->Bpad, B = d->B;
  PrenetP pp{};
  pp.w1t = d->pre_w1t, pp.b1 = d->pre_b1, pp.w2t = d->pre_w2t, pp.b2 = d->pre_b2;
  pp.stop_w = d->stop_w, pp.stop_b = d->stop_b;
  pp.h_d[0] = d->h_d[0], pp.h_d[1] = d->h_d[1], pp.o = d->o, pp.mem_override = mem_override;
  pp.x2 = d->x2, pp.stop_out = o.stop, pp.n_steps = o.n_steps, pp.ctrl = d->ctrl;
  pp.step_in_chunk = i, pp.Bpad = Bpad, pp.B = B, pp.r = r, pp.r_init = d->r_init, pp.stop_rows = o.stop_rows;
  pp.do_stop = stop_prev, pp.do_prenet = 1, pp.free_running = free_running, pp.threshold = thr;
  prenet_stop_kernel<<<B, 256, 0, st>>>(pp);
  TTSX_LAUNCH_CHECK();

  GemmP g{};
  g.Bpad = Bpad, g.B = B, g.ctrl = d->ctrl, g.step_in_chunk = i, g.r = r;
  // attention LSTM: [x2; ctx; h_a]
  g.Wp = d->w_lstm1, g.bias = d->b_lstm1, g.nseg = 3, g.KG = (PRENET_DIM + ENC_DIM + QUERY_DIM) / 16;
  g.seg[0] = Seg{{d->x2, nullptr}, 0, PRENET_DIM / 16};
  g.seg[1] = Seg{{d->ctx, nullptr}, 0, ENC_DIM / 16};
  g.seg[2] = Seg{{d->h_a[0], d->h_a[1]}, 1, QUERY_DIM / 16};
  g.out[0] = d->h_a[0], g.out[1] = d->h_a[1], g.c = d->c_a, g.rows = 4 * QUERY_DIM;
  int rc = launch_gemm<MODE_LSTM>(g, QUERY_DIM / 4, st);
  if (rc) return rc;
  // query projection
  g.Wp = d->w_query, g.bias = d->b_query, g.nseg = 1, g.KG = QUERY_DIM / 16;
  g.seg[0] = Seg{{d->h_a[0], d->h_a[1]}, 2, QUERY_DIM / 16};
  g.out[0] = d->q, g.out[1] = nullptr, g.c = nullptr, g.rows = ATT_DIM;
  rc = launch_gemm<MODE_LINEAR>(g, ATT_DIM / 16, st);
  if (rc) return rc;
  // attention
  AttnP ap{};
  ap.q = d->q, ap.P = d->P, ap.enc = d->enc, ap.enc_len = d->enc_len;
  ap.loc_conv_w = d->loc_conv_w, ap.loc_dense_w = d->loc_dense_w, ap.v_w = d->v_w, ap.v_b = d->v_b;
  ap.a_prev = d->a_prev, ap.a_cum = d->a_cum, ap.ctx = d->ctx, ap.align_out = o.align;
  ap.ctrl = d->ctrl, ap.step_in_chunk = i, ap.Bpad = Bpad, ap.T = d->T, ap.align_rows = o.align_rows;
  ap.norm = d->norm;
  attention_kernel<<<B, ATT_THREADS, attn_smem_bytes(d->T), st>>>(ap);
  TTSX_LAUNCH_CHECK();
  // decoder LSTM: [h_a(new); ctx; h_d]
  g.Wp = d->w_lstm2, g.bias = d->b_lstm2, g.nseg = 3, g.KG = (QUERY_DIM + ENC_DIM + DEC_DIM) / 16;
  g.seg[0] = Seg{{d->h_a[0], d->h_a[1]}, 2, QUERY_DIM / 16};
  g.seg[1] = Seg{{d->ctx, nullptr}, 0, ENC_DIM / 16};
  g.seg[2] = Seg{{d->h_d[0], d->h_d[1]}, 1, DEC_DIM / 16};
  g.out[0] = d->h_d[0], g.out[1] = d->h_d[1], g.c = d->c_d, g.rows = 4 * DEC_DIM;
  rc = launch_gemm<MODE_LSTM>(g, DEC_DIM / 4, st);
  if (rc) return rc;
  // mel projection: [h_d(new); ctx]
  g.Wp = d->w_proj, g.bias = d->b_proj, g.nseg = 2, g.KG = (DEC_DIM + ENC_DIM) / 16;
  g.seg[0] = Seg{{d->h_d[0], d->h_d[1]}, 2, DEC_DIM / 16};
  g.seg[1] = Seg{{d->ctx, nullptr}, 0, ENC_DIM / 16};
  g.out[0] = d->o, g.out[1] = nullptr, g.c = nullptr, g.rows = d->odim;
  g.mel = o.mel, g.mel_rows = o.mel_rows;
  rc = launch_gemm<MODE_PROJ>(g, ceil_div(d->odim, 16), st);
  return rc;
}

// stop token of the chunk's last step (index i inside the chunk)
int enqueue_stop_only(ttsx_t2_decoder* d, int i, int r, const StepOut& o, bool free_running, float thr,
                      hipStream_t st) {
  PrenetP pp{};
  pp.stop_w = d->stop_w, pp.stop_b = d->stop_b;
  pp.h_d[0] = d->h_d[0], pp.h_d[1] = d->h_d[1], pp.o = d->o;
  pp.stop_out = o.stop, pp.n_steps = o.n_steps, pp.ctrl = d->ctrl;
  pp.step_in_chunk = i + 1, pp.Bpad = d->Bpad, pp.B = d->B, pp.r = r, pp.r_init = d->r_init;
  pp.stop_rows = o.stop_rows, pp.do_stop = 1, pp.do_prenet = 0, pp.free_running = free_running, pp.threshold = thr;
  prenet_stop_kernel<<<d->B, 256, 0, st>>>(pp);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

}  // namespace

extern "C" int ttsx_t2_decoder_create(const ttsx_t2_decoder_weights* w, int r_init, int attn_norm, void* stream,
                                      ttsx_t2_decoder** out) {
  TTSX_ARG(w && out, "ttsx_t2_decoder_create: null argument");
  TTSX_ARG(r_init >= 1 && r_init <= 16, "ttsx_t2_decoder_create: r_init %d out of range", r_init);
  TTSX_ARG(attn_norm == TTSX_NORM_SOFTMAX || attn_norm == TTSX_NORM_SIGMOID,
           "Unknown value for attention norm type");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ttsx_t2_decoder* d = new ttsx_t2_decoder();
  memset(d, 0, sizeof(*d));
  d->r_init = r_init, d->norm = attn_norm, d->odim = MEL_DIM * r_init;
  int rc = 0;
  rc |= pack_layer(&d->w_lstm1, &d->b_lstm1, w->attn_rnn_w_ih, PRENET_DIM + ENC_DIM, w->attn_rnn_w_hh, QUERY_DIM,
                   w->attn_rnn_b_ih, w->attn_rnn_b_hh, 4 * QUERY_DIM, QUERY_DIM, st);
  rc |= pack_layer(&d->w_lstm2, &d->b_lstm2, w->dec_rnn_w_ih, QUERY_DIM + ENC_DIM, w->dec_rnn_w_hh, DEC_DIM,
                   w->dec_rnn_b_ih, w->dec_rnn_b_hh, 4 * DEC_DIM, DEC_DIM, st);
  rc |= pack_layer(&d->w_query, &d->b_query, w->query_w, QUERY_DIM, nullptr, 0, nullptr, nullptr, ATT_DIM, 0, st);
  rc |= pack_layer(&d->w_proj, &d->b_proj, w->proj_w, DEC_DIM + ENC_DIM, nullptr, 0, w->proj_b, nullptr, d->odim, 0,
                   st);
  if (rc) return TTSX_ERR_HIP;
  const int pin[2] = {MEL_DIM, PRENET_DIM};
  float** wt[2] = {&d->pre_w1t, &d->pre_w2t};
  float** bt[2] = {&d->pre_b1, &d->pre_b2};
  for (int i = 0; i < 2; ++i) {
    if (dev_alloc(wt[i], (size_t)pin[i] * PRENET_DIM) || dev_alloc(bt[i], PRENET_DIM)) return TTSX_ERR_HIP;
    pack_prenet_kernel<<<ceil_div(pin[i] * PRENET_DIM, 256), 256, 0, st>>>(
        *wt[i], *bt[i], w->prenet_w[i], w->prenet_bn_weight[i], w->prenet_bn_bias[i], w->prenet_bn_mean[i],
        w->prenet_bn_var[i], PRENET_DIM, pin[i]);
    TTSX_LAUNCH_CHECK();
  }
  rc |= copy_dev(&d->stop_w, w->stop_w, DEC_DIM + d->odim, st);
  rc |= copy_dev(&d->stop_b, w->stop_b, 1, st);
  rc |= copy_dev(&d->inputs_w, w->inputs_w, ATT_DIM * ENC_DIM, st);
  rc |= copy_dev(&d->v_w, w->v_w, ATT_DIM, st);
  rc |= copy_dev(&d->v_b, w->v_b, 1, st);
  rc |= copy_dev(&d->loc_conv_w, w->loc_conv_w, LOC_F * 2 * LOC_K, st);
  rc |= copy_dev(&d->loc_dense_w, w->loc_dense_w, ATT_DIM * LOC_F, st);
  if (rc) return TTSX_ERR_HIP;
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&d->ctrl), sizeof(Ctrl)));
  TTSX_HIP(hipMemsetAsync(d->ctrl, 0, sizeof(Ctrl), st));
  TTSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&d->host_flag), 4 * sizeof(int), hipHostMallocDefault));
  TTSX_HIP(hipStreamSynchronize(st));
  *out = d;
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_destroy(ttsx_t2_decoder* d) {
  if (!d) return TTSX_OK;
  free_workspace(d);
  float* bufs[] = {d->w_lstm1, d->b_lstm1, d->w_lstm2, d->b_lstm2, d->w_query, d->b_query, d->w_proj,
                   d->b_proj,  d->pre_w1t, d->pre_b1,  d->pre_w2t, d->pre_b2,  d->stop_w,  d->stop_b,
                   d->inputs_w, d->v_w,    d->v_b,     d->loc_conv_w, d->loc_dense_w};
  for (float* b : bufs)
    if (b) (void)hipFree(b);
  if (d->ctrl) (void)hipFree(d->ctrl);
  if (d->host_flag) (void)hipHostFree(d->host_flag);
  delete d;
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_begin(ttsx_t2_decoder* d, const float* enc, const int32_t* enc_len, int B, int T,
                                     void* stream) {
  TTSX_ARG(d && enc, "ttsx_t2_decoder_begin: null argument");
  TTSX_ARG(B >= 1 && B <= 4096 && T >= 1, "ttsx_t2_decoder_begin: bad shape B=%d T=%d", B, T);
  TTSX_ARG(attn_smem_bytes(T) <= 160 * 1024, "ttsx_t2_decoder_begin: encoder length %d exceeds the LDS budget", T);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = ensure_workspace(d, B, T);
  if (rc) return rc;
  d->B = B, d->T = T, d->enc = enc, d->enc_len = enc_len;
  const int cb = d->Bpad;
  TTSX_HIP(hipMemsetAsync(d->x2, 0, sizeof(float) * PRENET_DIM * cb, st));
  TTSX_HIP(hipMemsetAsync(d->ctx, 0, sizeof(float) * ENC_DIM * cb, st));
  for (int i = 0; i < 2; ++i) {
    TTSX_HIP(hipMemsetAsync(d->h_a[i], 0, sizeof(float) * QUERY_DIM * cb, st));
    TTSX_HIP(hipMemsetAsync(d->h_d[i], 0, sizeof(float) * DEC_DIM * cb, st));
  }
  TTSX_HIP(hipMemsetAsync(d->c_a, 0, sizeof(float) * QUERY_DIM * cb, st));
  TTSX_HIP(hipMemsetAsync(d->c_d, 0, sizeof(float) * DEC_DIM * cb, st));
  TTSX_HIP(hipMemsetAsync(d->q, 0, sizeof(float) * ATT_DIM * cb, st));
  TTSX_HIP(hipMemsetAsync(d->o, 0, sizeof(float) * round_up(d->odim, 16) * cb, st));
  TTSX_HIP(hipMemsetAsync(d->a_prev, 0, sizeof(float) * (size_t)B * T, st));
  TTSX_HIP(hipMemsetAsync(d->a_cum, 0, sizeof(float) * (size_t)B * T, st));
  if (attn_smem_bytes(T) > 64 * 1024)
    TTSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_smem_bytes(T)));
  inputs_proj_kernel<<<dim3(ceil_div(T, 16), B), 256, 0, st>>>(d->P, enc, d->inputs_w, T);
  TTSX_LAUNCH_CHECK();
  ctrl_set_kernel<<<1, 1, 0, st>>>(d->ctrl, 0, 0, 1);
  TTSX_LAUNCH_CHECK();
  d->steps_done = 0;
  d->begun = true;
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_load_state(ttsx_t2_decoder* d, const float* h_a, const float* c_a, const float* h_d,
                                          const float* c_d, const float* ctx, const float* a_prev,
                                          const float* a_cum, void* stream) {
  TTSX_ARG(d, "ttsx_t2_decoder_load_state: null handle");
  if (!d->begun) {
    ttsx_set_error("ttsx_t2_decoder_load_state called before ttsx_t2_decoder_begin");
    return TTSX_ERR_STATE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int B = d->B, Bpad = d->Bpad, cur = d->steps_done & 1;
  auto put = [&](float* dst, const float* src, int K) {
    if (src) to_quad_kernel<<<ceil_div(B * K, 256), 256, 0, st>>>(dst, src, B, K, Bpad);
  };
  put(d->h_a[cur], h_a, QUERY_DIM);
  put(d->c_a, c_a, QUERY_DIM);
  put(d->h_d[cur], h_d, DEC_DIM);
  put(d->c_d, c_d, DEC_DIM);
  put(d->ctx, ctx, ENC_DIM);
  TTSX_LAUNCH_CHECK();
  if (a_prev) TTSX_HIP(hipMemcpyAsync(d->a_prev, a_prev, sizeof(float) * (size_t)B * d->T, hipMemcpyDeviceToDevice, st));
  if (a_cum) TTSX_HIP(hipMemcpyAsync(d->a_cum, a_cum, sizeof(float) * (size_t)B * d->T, hipMemcpyDeviceToDevice, st));
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_read_state(ttsx_t2_decoder* d, float* h_a, float* c_a, float* h_d, float* c_d,
                                          float* ctx, float* a_prev, float* a_cum, void* stream) {
  TTSX_ARG(d, "ttsx_t2_decoder_read_state: null handle");
  if (!d->begun) {
    ttsx_set_error("ttsx_t2_decoder_read_state called before ttsx_t2_decoder_begin");
    return TTSX_ERR_STATE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int B = d->B, Bpad = d->Bpad, cur = d->steps_done & 1;
  auto get = [&](float* dst, const float* src, int K) {
    if (dst) from_quad_kernel<<<ceil_div(B * K, 256), 256, 0, st>>>(dst, src, B, K, Bpad);
  };
  get(h_a, d->h_a[cur], QUERY_DIM);
  get(c_a, d->c_a, QUERY_DIM);
  get(h_d, d->h_d[cur], DEC_DIM);
  get(c_d, d->c_d, DEC_DIM);
  get(ctx, d->ctx, ENC_DIM);
  TTSX_LAUNCH_CHECK();
  if (a_prev) TTSX_HIP(hipMemcpyAsync(a_prev, d->a_prev, sizeof(float) * (size_t)B * d->T, hipMemcpyDeviceToDevice, st));
  if (a_cum) TTSX_HIP(hipMemcpyAsync(a_cum, d->a_cum, sizeof(float) * (size_t)B * d->T, hipMemcpyDeviceToDevice, st));
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_step(ttsx_t2_decoder* d, const float* memory, int r, float* frames, float* align,
                                    float* stop, void* stream) {
  TTSX_ARG(d && memory && frames && align && stop, "ttsx_t2_decoder_step: null argument");
  TTSX_ARG(r >= 1 && r <= d->r_init, "ttsx_t2_decoder_step: r=%d outside 1..%d", r, d->r_init);
  if (!d->begun) {
    ttsx_set_error("ttsx_t2_decoder_step called before ttsx_t2_decoder_begin");
    return TTSX_ERR_STATE;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  ctrl_set_kernel<<<1, 1, 0, st>>>(d->ctrl, d->steps_done, 0, 1);
  TTSX_LAUNCH_CHECK();
  TTSX_HIP(hipMemsetAsync(d->n_steps_tmp, 0, sizeof(int32_t) * d->B, st));
  StepOut o{frames, r, align, 1, stop, 1, d->n_steps_tmp};
  int rc = enqueue_step(d, 0, r, o, memory, false, false, 2.f, st);
  if (rc) return rc;
  rc = enqueue_stop_only(d, 0, r, o, false, 2.f, st);
  if (rc) return rc;
  d->steps_done += 1;
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_run(ttsx_t2_decoder* d, const float* enc, const int32_t* enc_len, int B, int T, int r,
                                   int max_steps, float stop_threshold, int fixed_steps, float* mel, float* align,
                                   float* stop, int32_t* n_steps, int* steps_run_host, void* stream) {
  TTSX_ARG(d && enc && mel && align && stop && n_steps, "ttsx_t2_decoder_run: null argument");
  TTSX_ARG(r >= 1 && r <= d->r_init, "ttsx_t2_decoder_run: r=%d outside 1..%d", r, d->r_init);
  TTSX_ARG(max_steps >= 1, "ttsx_t2_decoder_run: max_steps=%d", max_steps);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = ttsx_t2_decoder_begin(d, enc, enc_len, B, T, stream);
  if (rc) return rc;
  TTSX_HIP(hipMemsetAsync(n_steps, 0, sizeof(int32_t) * B, st));
  StepOut o{mel, max_steps * r, align, max_steps, stop, max_steps, n_steps};
  const bool free_running = !fixed_steps;
  const int CHUNK = 32;
  int launched = 0;
  d->host_flag[0] = 0;
  while (launched < max_steps) {
    const int n = std::min(CHUNK, max_steps - launched);
    for (int i = 0; i < n; ++i) {
      rc = enqueue_step(d, i, r, o, nullptr, i > 0, free_running, stop_threshold, st);
      if (rc) return rc;
    }
    rc = enqueue_stop_only(d, n - 1, r, o, free_running, stop_threshold, st);
    if (rc) return rc;
    ctrl_advance_kernel<<<1, 1, 0, st>>>(d->ctrl, n);
    TTSX_LAUNCH_CHECK();
    launched += n;
    if (free_running) {
      TTSX_HIP(hipMemcpyAsync(d->host_flag, &d->ctrl->all_done, sizeof(int), hipMemcpyDeviceToHost, st));
      TTSX_HIP(hipStreamSynchronize(st));
      if (d->host_flag[0]) break;
    }
  }
  finalize_steps_kernel<<<ceil_div(B, 256), 256, 0, st>>>(n_steps, B, d->ctrl, launched);
  TTSX_LAUNCH_CHECK();
  // steps actually executed = max over utterances of n_steps
  std::vector<int32_t> host(B);
  TTSX_HIP(hipMemcpyAsync(host.data(), n_steps, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipStreamSynchronize(st));
  int run = 0;
  for (int b = 0; b < B; ++b) run = std::max(run, (int)host[b]);
  if (fixed_steps) run = launched;
  if (steps_run_host) *steps_run_host = run;
  d->steps_done = run;
  d->begun = false;  // the state belongs to a finished utterance batch
  return TTSX_OK;
}
