// Tacotron2 decoder for gfx950.
//
// What it computes per step follows Decoder.decode (Tacotron2/TTS/models.py:578-617) with
// Prenet (:76-82), OriginalAttention.forward (:314-361), LocationLayer (:84-104) and the
// greedy loop of Decoder.inference (:643-673):
//   x2  = prenet(memory)
//   h_a = LSTMCell([x2; ctx], h_a, c_a)
//   q   = W_q h_a
//   a   = norm(v . tanh(q + W_ld conv1d([a_prev; a_cum]) + P) + b_v) ; a_cum += a
//   ctx = a . enc
//   h_d = LSTMCell([h_a; ctx], h_d, c_d)
//   o   = W_o [h_d; ctx] + b_o ;  stop = sigmoid(w_s [h_d; o] + b_s)
//
// Launches per step (captured into a hipGraph per chunk of steps):
//   prenet layer 2 | attention LSTM | query | energies | normalise+context | decoder LSTM | projection
// or, for texts short enough for one workgroup per utterance (FusedAttnOp),
//   prenet layer 2 | attention LSTM (+ its shares of the query projection) | fused attention | decoder LSTM |
//   projection.
// Each LSTM can be cut in two along K (always at batch 1-2; on the MFMA engine where it measured faster, see
// ttsx_t2_decoder_begin).  The columns whose inputs are known early (attention LSTM:
// previous context and hidden state; decoder LSTM: new attention hidden state and own previous
// hidden state) are multiplied by "partial" work that leaves raw gate sums in a buffer; the launch
// on the serial path only adds the late columns (prenet output, 256; new context, 512) and applies
// the cell update.  The partial work is bandwidth/MFMA-bound while the small launches on the serial
// path (prenet, query, energies, context, projection) are latency-bound, so the partial row blocks
// are dealt out in three pieces that ride along in the same launches as those small kernels
// (a launch = workgroups of the small op first, then workgroups of a partial piece): the heavy
// streaming runs in the shadow of the serial chain without any cross-stream synchronisation.
// (On the MFMA engine the pieces are 32 x 32 tiles over two halves of the early K range, with one gate-sum
// buffer per half.)
// The projection launch carries, besides the 80*r_init mel rows, two sets of composed rows so
// that nothing else sits on the serial path between two steps:
//   * one row  u = [w_s,h + W_o,h^T w_s,o ; W_o,c^T w_s,o]  whose product with [h_d; ctx] is the
//     stopnet logit (w_s [h_d; o] + b_s with o substituted), and
//   * 256 rows (A1 W_o[last frame rows]) whose product is the pre-activation of prenet layer 1
//     for the NEXT step (prenet layer 1 is linear in the fed-back frame; BatchNorm folded).
// Both are exact algebraic rewrites; rounding differs from the reference at the 1e-6 level.
//
// Layouts.  Weights are packed once: 16 output rows form a row block; inside a row block the
// K axis is cut in groups of 16 and each group is stored as 64 lanes x 4 floats in the order
// v_mfma_f32_16x16x4_f32 wants its A operand over four consecutive issues (lane l: row l&15,
// k = 4*(l>>4) + s for issue s), so a wave streams weights with fully coalesced 1 KiB loads.
// LSTM gate rows are interleaved (row 4u+g of a block = gate g of hidden unit u): the four
// accumulator registers of a lane are the four gates of one (unit, utterance) pair and the
// cell update is lane-local.  Activations are quad-interleaved (qidx in common.h).
//
// Two engines share the packed weights: B <= 2 utterances run a GEMV engine (weights straight
// to VGPRs with every load of a wave in flight at once, activations staged in LDS; this regime
// is HBM-bandwidth bound), larger batches run the fp32 MFMA engine.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "enc_scan.h"
#include "t2_team.h"

namespace {

struct RunCtx {
  int base;      // decoder steps completed before the current chunk (state parity)
  int out_base;  // output row of the chunk's first step
  int all_done;  // every utterance has emitted its stop frame (free-running mode): what the host polls
  int done_at;   // index of the step during which that happened (INT_MAX before); see step_done()
  int n_done;
  int B, T, r, free_running;
  int max_steps;  // Tacotron (v1) stop rule: max_decoder_steps
  int mask_finished;  // output rows of an utterance whose n_steps is already set are left unwritten
  float threshold;
  int mel_rows, align_rows, stop_rows;
  float* mel;        // [B, mel_rows, 80]
  float* align;      // [B, align_rows, T]
  float* stop;       // [B, stop_rows]
  int32_t* n_steps;  // [B]
  // words in pinned host memory that the device writes and the host polls while it keeps enqueueing chunks (no
  // stream synchronisation, no copy): [0] every utterance has ended, [1] steps completed (written by the last
  // node of every chunk).  Null: not reported.
  int* host_words;
  int sync_failed;  // a workgroup that waits for other workgroups of its own launch gave up (see Prenet2AfterProjOp)
};

// A step's launches do their arithmetic but keep nothing once every utterance has ended.  The test compares the
// step's own index with the index of the step that raised the flag, so it cannot change under a launch: the
// workgroup that detects the last stop shares its launch with the workgroups that write that step's frames, and
// those must not see the flag (a plain flag would rest on the order in which workgroups are dispatched).
constexpr int NEVER_DONE = 0x7fffffff;
__device__ __forceinline__ int step_done(const RunCtx* c, int step_in_chunk) {
  return c->base + step_in_chunk > c->done_at;
}

constexpr int ENC_DIM = 512, QUERY_DIM = 1024, DEC_DIM = 1024, PRENET_DIM = 256, ATT_DIM = 128;
constexpr int MEL_DIM = 80, LOC_F = 32, LOC_K = 31;
constexpr int GEMM_THREADS = 512, GEMM_WAVES = GEMM_THREADS / 64;
constexpr int GEMV_MAX_B = 2;

// ------------------------------------------------------------------ weight packing
// dst[((rb*KG + kg)*64 + lane)*4 + s] = src(row)[kg*16 + 4*(lane>>4) + s], row = rb*16 + (lane&15)
// (or, with lstm_hidden > 0, packed row p of block rb <- source row (p&3)*hidden + 4*rb + (p>>2)).
// The source row is the concatenation [w1 row | w2 row].
__global__ void pack_rows_kernel(float* __restrict__ dst, const float* __restrict__ w1, int k1,
                                 const float* __restrict__ w2, int k2, int rows, int KG, int lstm_hidden) {
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
    else if (w2 && k - k1 < k2) v = w2[(size_t)row * k2 + (k - k1)];  // (null w2: zero padding of K)
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

// dst row = [src columns k1 .. k1+k2-1 | src columns 0 .. k1-1]; one block per row
__global__ void swap_cols_kernel(float* __restrict__ dst, const float* __restrict__ src, int k1, int k2) {
  const size_t row = blockIdx.x;
  const int K = k1 + k2;
  for (int k = threadIdx.x; k < K; k += blockDim.x)
    dst[row * K + k] = src[row * K + (k < k2 ? k1 + k : k - k2)];
}

// Prenet layer with eval-mode BatchNorm folded in (row-major [n_out][n_in]):
// wf[n][k] = w[n][k]*g[n]/sqrt(var[n]+eps), bf[n] = beta[n] - mean[n]*g[n]/sqrt(var[n]+eps).
// Without BatchNorm ("original" prenet, bias-free Linear) the scale is 1 and the bias 0.
__global__ void fold_prenet_kernel(float* __restrict__ wf, float* __restrict__ bf, const float* __restrict__ w,
                                   const float* __restrict__ g, const float* __restrict__ beta,
                                   const float* __restrict__ mean, const float* __restrict__ var, int n_out,
                                   int n_in) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out * n_in) return;
  int n = i / n_in, k = i % n_in;
  float sc = 1.f;
  if (g) sc = g[n] / sqrtf(var[n] + 1e-5f);
  wf[i] = w[i] * sc;
  if (k == 0) bf[n] = g ? beta[n] - mean[n] * sc : 0.f;
}

// Rows of the projection launch beyond the mel rows (fp64 accumulation, done once per r):
//   ext_w[0]        = [w_s,h + W_o,h^T w_s,o ; W_o,c^T w_s,o],  ext_b[0] = w_s,o . b_o + b_s
//   ext_w[1..15]    = 0
//   ext_w[16 + n]   = sum_j A1f[n][j] W_o[80(r-1)+j][:],        ext_b[16+n] = sum_j A1f[n][j] b_o[80(r-1)+j] + b1f[n]
__global__ void compose_proj_ext_kernel(float* __restrict__ ext_w, float* __restrict__ ext_b,
                                        const float* __restrict__ proj_w, const float* __restrict__ proj_b,
                                        const float* __restrict__ stop_w, const float* __restrict__ stop_b,
                                        const float* __restrict__ a1f, const float* __restrict__ b1f, int odim,
                                        int r) {
  const int K = DEC_DIM + ENC_DIM;
  const int row = blockIdx.x;  // 0 .. 16+255
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    double acc = 0.0;
    if (row == 0) {
      for (int j = 0; j < odim; ++j) acc += (double)stop_w[DEC_DIM + j] * (double)proj_w[(size_t)j * K + k];
      if (k < DEC_DIM) acc += (double)stop_w[k];
    } else if (row >= 16) {
      const int n = row - 16;
      for (int j = 0; j < MEL_DIM; ++j)
        acc += (double)a1f[n * MEL_DIM + j] * (double)proj_w[(size_t)(MEL_DIM * (r - 1) + j) * K + k];
    }
    ext_w[(size_t)row * K + k] = (float)acc;
  }
  if (threadIdx.x == 0) {
    double acc = 0.0;
    if (row == 0) {
      for (int j = 0; j < odim; ++j) acc += (double)stop_w[DEC_DIM + j] * (double)proj_b[j];
      acc += (double)stop_b[0];
    } else if (row >= 16) {
      const int n = row - 16;
      for (int j = 0; j < MEL_DIM; ++j) acc += (double)a1f[n * MEL_DIM + j] * (double)proj_b[MEL_DIM * (r - 1) + j];
      acc += (double)b1f[n];
    }
    ext_b[row] = (float)acc;
  }
}

#ifdef TTSX_FA_TIMING
// phase timestamps of workgroup 0 (100 MHz wall clock), for tuning builds only
__device__ long long g_fa_t[16];
#define FA_MARK(i) do { if (wg == 0 && threadIdx.x == 0) g_fa_t[i] = wall_clock64(); } while (0)
// start, end and placement of every workgroup of the launches whose grid size equals g_wg_grid (each such launch
// overwrites the one before): [start, end, HW_ID register, XCC_ID register]
__device__ long long g_wg_t[2048][4];
__device__ int g_wg_grid, g_wg_first;
__device__ __forceinline__ void wg_mark(long long t0, int n_first) {
  if (threadIdx.x == 0 && (int)gridDim.x == g_wg_grid && n_first == g_wg_first && gridDim.x <= 2048) {
    long long* r = g_wg_t[blockIdx.x];
    r[0] = t0, r[1] = wall_clock64();
    r[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4), r[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
}
#define WG_T0() const long long wg_t0 = wall_clock64()
#define WG_MARK(n) wg_mark(wg_t0, n)
#else
#define FA_MARK(i) do { } while (0)
#define WG_T0() do { } while (0)
#define WG_MARK(n) do { } while (0)
#endif

// ------------------------------------------------------------------ GEMM / GEMV engines
struct Seg {
  const float* x[2];  // quad-interleaved activation buffer(s)
  int sel;            // 0: x[0]; 1: x[parity] (state before this step); 2: x[parity^1] (new state)
  int kgroups;        // K of the segment / 16
};

enum { MODE_LINEAR = 0, MODE_LSTM = 1, MODE_PROJ = 2, MODE_PARTIAL = 3, MODE_GRU = 4, MODE_MEL = 5 };

// Everything a launch needs before its arithmetic is a kernel argument (no pointer is fetched
// from device memory first), so that all global loads of a workgroup - weights, activations and
// the operands of the epilogue - leave in one burst and the launch costs one memory round trip.
struct GemmP {
  const float* Wp;
  const float* bias;
  Seg seg[3];
  int nseg, KG, Bpad;
  int kg0, kg_stride;  // first k-group of this launch inside a row block, k-groups per row block
  int rb0, nrb;        // row blocks rb0 .. rb0+nrb-1 are this launch's; workgroup id = tile * nrb + (rb - rb0)
  int parity;          // (decoder steps completed before this one) & 1: selects the h ping-pong buffers
  float* gates;        // PARTIAL: raw sums out (quad layout); LSTM: sums of the partial launch, added in
  int gates_pitch;     // columns per quad row of `gates` (0: Bpad) and first column of this launch in it
  int gates_col0;
  // K cut into parts (MFMA-engine pieces): a PARTIAL launch covers parts kpart0, kpart0+1, ... (one per block of
  // workgroups, KGW*8 k-groups each) and writes part k to gates + k * gates_part_stride (float4 units); an LSTM
  // launch adds up `kparts` such buffers.  0 or 1: a single part.
  int kparts, kpart0, gates_part_stride;
  // LSTM over a sequence (encoder): h also goes to seq_out[b * seq_pitch + unit]; columns whose
  // utterance does not reach time index t_index (t_index >= len[b]) hold zero state and output
  float* seq_out;
  size_t seq_pitch;
  const int32_t* len;
  int t_index, n_valid;  // n_valid: utterances (columns below it are real)
  float* out[2];  // LINEAR/PROJ: out[0] quad layout; LSTM: h ping-pong pair, written at parity^1
  float* c;       // LSTM cell state, quad layout
  float* x1;      // PROJ: prenet layer-1 activations for the next step, quad layout
  // PROJ with prenet layer 2 of the next step in the same launch (Prenet2AfterProjOp): x1 leaves through
  // cache-bypassing stores and the row block then raises x1_flags[column tile * 16 + block] to the step's tag
  unsigned* x1_flags;
  // The late decoder-LSTM columns inside the attention launch (GemmOp<..., LATE>): step tags raised by the producers
  // of what they read - [2 b + w]: waves w = 0, 1 of utterance b's attention workgroup have stored their halves of
  // the context; [2 Bpad + (part * nrb / 8 + row group) * 8 + w]: wave w of a staged piece has stored its partial
  // sums.  Both kinds of data leave through coh_store16 and are read through coh_load16.  Null: off.
  unsigned* late_flags;
  // GRU (Tacotron v1; rows of a hidden unit = r, z, n_input, n_hidden over K = [x; h]): h ping-pong in
  // `out` like the LSTM; with `resid` the launch also writes sum_out = h_new + resid (residual GRU stack)
  const float* resid;
  const float* resid2;  // optional second residual term
  float* sum_out;
  // MEL (Tacotron v1 proj_to_mel): sigmoid frames to out[0], to the run's mel rows and to the tail of the
  // next memory queue (quads mem_keep .. mem_keep + o_quads - 1 of mem_new)
  float* mem_new;
  int mem_keep;
  int relu;       // LINEAR: apply ReLU
  int o_quads;    // PROJ: number of quads that are mel rows (80*r_init/4)
  // LSTM (attention LSTM only): the workgroup also leaves its share of the query projection,
  //   qpart[(g * Bpad + b) * 128 + d] = sum over the workgroup's hidden units u of wq_t[u][d] * h[u][b],
  // g = index of the workgroup's unit group; the attention kernel adds the groups up
  const float* wq_t;  // [1024][128] query_layer weight, transposed
  float* qpart;
  RunCtx* ctx;
  int step_in_chunk;
};

// 16-byte accesses past the (per-XCD, not mutually coherent) L2, for data that workgroups of ONE launch hand to
// each other: buffer instructions with the system-coherent cache policy.  Unlike volatile accesses (a wait after
// each) or inline assembly (loads the compiler does not count) they are scheduled and waited for like any other
// load.  `base` is wave-uniform; the offset is in bytes (buffers here are far below 2 GB).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
constexpr int COHERENT_POLICY = 17;  // sc0 | sc1
__device__ __forceinline__ f32x4 coh_load16(const void* base, unsigned byte_off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, COHERENT_POLICY));
}
__device__ __forceinline__ void coh_store16(void* base, unsigned byte_off, f32x4 v) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, (int)byte_off, 0, COHERENT_POLICY);
}

// K parts of the decoder LSTM's early columns when its pieces stage their activations in LDS (StagedPieceOp)
constexpr int STAGED_PARTS = 6;

// Operands of the epilogue of one (quad, utterance) pair, fetched while the weights stream.
struct Pre {
  float4 bias, gates;
  float c;
};
template <int MODE, bool COH = false>
__device__ __forceinline__ Pre prefetch_quad(const GemmP& p, int quad, int b) {
  Pre r;
  r.bias = make_float4(0.f, 0.f, 0.f, 0.f), r.gates = r.bias, r.c = 0.f;
  if (MODE != MODE_PARTIAL) r.bias = reinterpret_cast<const float4*>(p.bias)[quad];
  if (MODE == MODE_LSTM) {
    r.c = p.c[qidx(quad, b, p.Bpad)];
    const int gp = p.gates_pitch ? p.gates_pitch : p.Bpad;
    if (COH) {
      // the partial sums of staged pieces of this same launch, added in part order
      const unsigned off = (unsigned)(((size_t)quad * gp + p.gates_col0 + b) * sizeof(float4));
      f32x4 gk[STAGED_PARTS];
#pragma unroll
      for (int k = 0; k < STAGED_PARTS; ++k) gk[k] = coh_load16(p.gates, off + (unsigned)(k * p.gates_part_stride * sizeof(float4)));
      f32x4 sum = gk[0];
#pragma unroll
      for (int k = 1; k < STAGED_PARTS; ++k) sum += gk[k];
      r.gates = make_float4(sum[0], sum[1], sum[2], sum[3]);
    } else if (p.gates) {
      const float4* G = reinterpret_cast<const float4*>(p.gates) + (size_t)quad * gp + p.gates_col0 + b;
      r.gates = G[0];
      if (p.kparts == STAGED_PARTS) {
        // (all requested before the first sum; added in part order)
        float4 gk[STAGED_PARTS - 1];
#pragma unroll
        for (int k = 1; k < STAGED_PARTS; ++k) gk[k - 1] = G[(size_t)k * p.gates_part_stride];
#pragma unroll
        for (int k = 0; k < STAGED_PARTS - 1; ++k)
          r.gates.x += gk[k].x, r.gates.y += gk[k].y, r.gates.z += gk[k].z, r.gates.w += gk[k].w;
      } else if (p.kparts > 1) {
        const float4 g2 = G[p.gates_part_stride];
        r.gates.x += g2.x, r.gates.y += g2.y, r.gates.z += g2.z, r.gates.w += g2.w;
      }
    }
  }
  if (MODE == MODE_GRU) {
    const float* h_in = p.parity ? p.out[1] : p.out[0];  // the state before this step
    r.c = h_in[qidx(quad, b, p.Bpad)];
    if (p.resid) r.gates.x = p.resid[qidx(quad, b, p.Bpad)];
    if (p.resid2) r.gates.y = p.resid2[qidx(quad, b, p.Bpad)];
  }
  return r;
}

// Finish one quad (4 consecutive output rows) of utterance column b.
// (LSTM: returns the new hidden state of the pair, otherwise 0)
template <int MODE>
__device__ __forceinline__ float epilogue_quad(const GemmP& p, int quad, int b, f32x4 v, const Pre& pre,
                                               int kpart = 0) {
  const int Bpad = p.Bpad;
  if (MODE == MODE_PARTIAL) {
    const int gp = p.gates_pitch ? p.gates_pitch : Bpad;
    reinterpret_cast<float4*>(p.gates)[(size_t)kpart * p.gates_part_stride + (size_t)quad * gp + p.gates_col0 + b] =
        make_float4(v[0], v[1], v[2], v[3]);
    return 0.f;
  }
  v[0] += pre.bias.x, v[1] += pre.bias.y, v[2] += pre.bias.z, v[3] += pre.bias.w;
  if (MODE == MODE_LSTM) {
    // the four rows are gates i,f,g,o of hidden unit `quad`
    const size_t idx = qidx(quad, b, Bpad);
    v[0] += pre.gates.x, v[1] += pre.gates.y, v[2] += pre.gates.z, v[3] += pre.gates.w;
    float c_new = sigmoidf_(v[1]) * pre.c + sigmoidf_(v[0]) * tanhf(v[2]);
    float h_new = sigmoidf_(v[3]) * tanhf(c_new);
    if (p.seq_out) {
      const bool live = b < p.n_valid && (!p.len || p.t_index < p.len[b]);
      if (!live) c_new = 0.f, h_new = 0.f;
      if (b < p.n_valid) p.seq_out[(size_t)b * p.seq_pitch + quad] = h_new;
    }
    p.c[idx] = c_new;
    float* h_out = p.parity ? p.out[0] : p.out[1];
    h_out[idx] = h_new;
    return h_new;
  } else if (MODE == MODE_GRU) {
    // torch.nn.GRUCell: r = s(W_ir x + W_hr h + b), z likewise, n = tanh(W_in x + b_in + r (W_hn h + b_hn))
    const size_t idx = qidx(quad, b, Bpad);
    const float rg = sigmoidf_(v[0]), zg = sigmoidf_(v[1]);
    const float ng = tanhf(v[2] + rg * v[3]);
    const float h_new = (1.f - zg) * ng + zg * pre.c;
    float* h_out = p.parity ? p.out[0] : p.out[1];
    h_out[idx] = h_new;
    if (p.sum_out) p.sum_out[idx] = (h_new + pre.gates.x) + pre.gates.y;
    return h_new;
  } else if (MODE == MODE_MEL) {
    RunCtx* cx = p.ctx;
    if (quad < p.o_quads) {
      const float4 o = make_float4(sigmoidf_(v[0]), sigmoidf_(v[1]), sigmoidf_(v[2]), sigmoidf_(v[3]));
      reinterpret_cast<float4*>(p.out[0])[(size_t)quad * Bpad + b] = o;
      reinterpret_cast<float4*>(p.mem_new)[(size_t)(p.mem_keep + quad) * Bpad + b] = o;
      if (b < cx->B && !(cx->mask_finished && cx->n_steps[b] != 0)) {
        const int row0 = (cx->out_base + p.step_in_chunk) * cx->r;
        *reinterpret_cast<float4*>(cx->mel + ((size_t)b * cx->mel_rows + row0) * MEL_DIM + quad * 4) = o;
      }
    }
  } else if (MODE == MODE_LINEAR) {
    if (p.relu) v = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    reinterpret_cast<float4*>(p.out[0])[(size_t)quad * Bpad + b] = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    RunCtx* cx = p.ctx;
    if (quad < p.o_quads) {
      reinterpret_cast<float4*>(p.out[0])[(size_t)quad * Bpad + b] = make_float4(v[0], v[1], v[2], v[3]);
      if (b < cx->B && quad * 4 < cx->r * MEL_DIM) {
        // frames of this step, frame-major [B, mel_rows, 80]: o[j] -> frame j/80, bin j%80
        const int row0 = (cx->out_base + p.step_in_chunk) * cx->r;
        float* dst = cx->mel + ((size_t)b * cx->mel_rows + row0) * MEL_DIM + quad * 4;
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else if (quad < p.o_quads + 4) {
      if (quad == p.o_quads && b < cx->B) {
        const float s = sigmoidf_(v[0]);
        cx->stop[(size_t)b * cx->stop_rows + cx->out_base + p.step_in_chunk] = s;
        if (s > cx->threshold && cx->n_steps[b] == 0) {
          cx->n_steps[b] = cx->base + p.step_in_chunk + 1;  // steps run so far, the stop frame included
          const int done = atomicAdd(&cx->n_done, 1) + 1;
          if (cx->free_running && done == cx->B) {
            cx->done_at = cx->base + p.step_in_chunk, cx->all_done = 1;
            if (cx->host_words) __hip_atomic_store(cx->host_words, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
    } else {
      const int xq = quad - p.o_quads - 4;
      const float4 x = make_float4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
      if (p.x1_flags) {
        // read by other workgroups of this launch: past the (per-XCD, not mutually coherent) L2
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.x1) + ((size_t)xq * Bpad + b) * 2;
        __hip_atomic_store(dst, ((unsigned long long)__float_as_uint(x.y) << 32) | __float_as_uint(x.x), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, ((unsigned long long)__float_as_uint(x.w) << 32) | __float_as_uint(x.z),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        reinterpret_cast<float4*>(p.x1)[(size_t)xq * Bpad + b] = x;
      }
    }
  }
  return 0.f;
}

// (selects, not an index into x[]: a dynamically indexed copy of the argument struct would be
// spilled to a per-lane array)
__device__ __forceinline__ const float* seg_ptr(const Seg& sg, int parity) {
  const float *a = sg.x[0], *b = sg.x[1];
  const bool second = sg.sel != 0 && ((parity != 0) != (sg.sel == 2));
  return second ? b : a;
}

// fp32 MFMA engine: R row blocks x NT column tiles of 16 utterances per workgroup, the K axis
// split over the 8 waves (wave w takes k-groups w, w+8, ...), partial tiles combined through LDS.
// KGW = k-groups per wave.  The k-loop is fully unrolled and software-pipelined: the loads of
// k-group i+D are issued before the MFMAs of k-group i, so D groups (weights from HBM, activations
// from L2) are always in flight per wave.  (Issuing the whole weight stream first is slower here:
// vector loads return in order, so the first activation tile would queue behind all the weights
// and no MFMA could overlap the stream.)  The activation tiles are what bounds the big launches
// (every workgroup re-reads its 16*NT columns of the whole K range through the vector cache; the
// weight stream costs next to nothing beside it), hence R = 2 for the LSTMs: one activation load
// feeds two row blocks.
// DYN: the combine buffer lives in the launch's dynamic LDS instead of a static array (for pieces that share a
// launch with the fused attention kernel: its ~55 KB of dynamic LDS are unused by the piece workgroups, and a
// second 32 KB static array would push the launch past two workgroups per CU).
// LATE (the decoder LSTM's columns over the context, 32 x 32 tiles, inside the launch that forms the context): the
// workgroup requests its weights, then every wave waits for the step tags of the producers of what it reads - the
// attention workgroups of its 32 utterances and the staged pieces of its rows (p.late_flags) - and reads the context
// tiles and the six partial sums past the L2.  Its block indices come after the producers', so they are running
// or done when it starts; the wait is bounded all the same (RunCtx::sync_failed).
template <int R, int NT, int KGW, int MODE, bool DYN = false, bool LATE = false>
struct GemmOp {
  typedef GemmP P;
  static __device__ __forceinline__ void run(const GemmP& p, int wg_in) {
    int wg = wg_in;
    static_assert(!LATE || (R == 2 && NT == 2 && MODE == MODE_LSTM && KGW <= 4), "the late op: 32 x 32 tiles, every k-group up front");
    if (LATE && step_done(p.ctx, p.step_in_chunk)) return;  // (the producers raise no flags for such a step)
    if (KGW == 20) FA_MARK(8);
    // (the projection is on the step's serial path and shares its CUs with piece workgroups: its waves go first)
    if (MODE == MODE_PROJ) __builtin_amdgcn_s_setprio(3);
    const int done = step_done(p.ctx, p.step_in_chunk);
    const int ngrp = p.nrb / R;  // row-block groups of this launch
    int kpart = 0;
    if (MODE == MODE_PARTIAL && p.kparts > 1) {
      // one block of workgroups per part of K
      const int per = ngrp * (p.Bpad / (NT * 16));
      kpart = wg / per, wg -= kpart * per, kpart += p.kpart0;
    }
    const int kbase = kpart * (KGW * GEMM_WAVES);
    const int rb = p.rb0 + (wg % ngrp) * R, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b0 = (wg / ngrp) * (NT * 16);
    // k-groups in flight per wave (deeper prefetch on the big tiles measured slower: 54.0 / 56.4 / 57.2 us per
    // step at 6 / 8 / 10 against 53.4 at 4)
    // (64 x 64 tiles: three, or the 128 operand registers of four groups push the kernel past 256 and it spills)
    constexpr int DMAX = (R * NT >= 16) ? 3 : (R * NT >= 4) ? 4 : (NT == 1 && R == 1 ? 8 : 6);
    constexpr int D = KGW < DMAX ? KGW : DMAX;
    constexpr int TILES = R * NT, TPW = (TILES + GEMM_WAVES - 1) / GEMM_WAVES;
    f32x4(*red)[TILES][64];
    if constexpr (DYN) {
      extern __shared__ float4 gemm_dyn_lds[];
      red = reinterpret_cast<f32x4(*)[TILES][64]>(gemm_dyn_lds);
    } else {
      __shared__ f32x4 red_s[GEMM_WAVES][TILES][64];
      red = red_s;
    }

    const float4* W =
        reinterpret_cast<const float4*>(p.Wp) + ((size_t)rb * p.kg_stride + p.kg0 + kbase + wave) * 64 + lane;
    const size_t wrs = (size_t)p.kg_stride * 64;  // float4 stride between row blocks
    // activation pointers of the (up to three) segments, positioned at this lane's quad row / column
    const size_t xoff = (size_t)(lane >> 4) * p.Bpad + b0 + (lane & 15);
    const float4* X0 = reinterpret_cast<const float4*>(seg_ptr(p.seg[0], p.parity)) + xoff;
    const float4* X1 = p.nseg > 1 ? reinterpret_cast<const float4*>(seg_ptr(p.seg[1], p.parity)) + xoff : X0;
    const float4* X2 = p.nseg > 2 ? reinterpret_cast<const float4*>(seg_ptr(p.seg[2], p.parity)) + xoff : X0;
    const int e0 = p.seg[0].kgroups, e1 = e0 + (p.nseg > 1 ? p.seg[1].kgroups : 0);
    const size_t xs = (size_t)4 * p.Bpad;  // float4 stride between k-groups
    auto xptr = [&](int gl) -> const float4* {
      const int g = gl + kbase;
      return g < e0 ? X0 + (size_t)g * xs : (g < e1 ? X1 + (size_t)(g - e0) * xs : X2 + (size_t)(g - e1) * xs);
    };
    float4 wv[D][R], xv[D][NT];
    // (LATE) the step tags this wave waits for.  lane l: wave l & 1 of the attention workgroup of utterance
    // b0 + l / 2 (padding columns have none); lanes 0-11 also: the staged piece of part l / 2 that holds row block
    // rb + (l & 1).  Most workgroups start when their producers are through: one look at the tags BEFORE anything
    // else is requested (vector loads return in order - behind the weights the answer would come 2 us later), and
    // if they are all up, weights, context tiles and partial sums leave in one burst.
    unsigned late_tag = 0;
    const unsigned *late_fa = nullptr, *late_fp = nullptr;
    bool late_ready = false;
    auto late_look = [&]() {
      const unsigned a = b0 + (lane >> 1) < p.ctx->B ? __hip_atomic_load(late_fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : late_tag;
      const unsigned b = lane < 2 * STAGED_PARTS ? __hip_atomic_load(late_fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : late_tag;
      return __all(a == late_tag && b == late_tag) != 0;
    };
    if constexpr (LATE) {
      late_tag = (unsigned)(p.ctx->base + p.step_in_chunk + 1);
      late_fa = p.late_flags + 2 * min(b0 + (lane >> 1), p.Bpad - 1) + (lane & 1);
      late_fp = p.late_flags + 2 * p.Bpad +
                ((min(lane >> 1, STAGED_PARTS - 1) * (p.nrb / GEMM_WAVES) + rb / GEMM_WAVES) * GEMM_WAVES + (rb % GEMM_WAVES) + (lane & 1));
      late_ready = late_look();
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
      for (int r = 0; r < R; ++r) wv[i][r] = W[(size_t)i * GEMM_WAVES * 64 + r * wrs];
      if constexpr (!LATE) {
        const float4* xp = xptr(wave + i * GEMM_WAVES);
#pragma unroll
        for (int n = 0; n < NT; ++n) xv[i][n] = xp[n * 16];
      }
    }
    if constexpr (LATE) {
      for (long spin = 0; !late_ready && spin < 4000000; ++spin) {
        __builtin_amdgcn_s_sleep(1);
        late_ready = late_look();
      }
      if (!late_ready) {
        if (threadIdx.x == 0) p.ctx->sync_failed = 1;
        return;  // (no barrier has been passed: the waves of the workgroup leave one by one)
      }
      asm volatile("" ::: "memory");  // (the compiler keeps the loads below behind the looks at the tags)
      const float* xb = seg_ptr(p.seg[0], p.parity);
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const f32x4 v = coh_load16(xb, (unsigned)((xoff + (size_t)(wave + i * GEMM_WAVES + kbase) * xs + n * 16) * sizeof(float4)));
          xv[i][n] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    // query-projection share: thread (d = tid & 127, cg = tid >> 7) covers columns cg*4*NT .. +4*NT-1
    constexpr int QU = 4 * R, QC = 4 * NT;
    float wq[QU];
    const bool qshare = (MODE == MODE_LSTM || MODE == MODE_GRU) && p.qpart != nullptr;
    if (qshare) {
#pragma unroll
      for (int j = 0; j < QU; ++j) wq[j] = p.wq_t[(size_t)(rb * 4 + j) * ATT_DIM + (threadIdx.x & (ATT_DIM - 1))];
    }
    __shared__ float s_h[QU][NT * 16];
    // epilogue operands of the tile(s) this wave finishes: tile t = r * NT + n
    Pre pre[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int t = wave + k * GEMM_WAVES;
      if (t < TILES)
        pre[k] = prefetch_quad<MODE, LATE>(p, (rb + t / NT) * 4 + (lane >> 4), b0 + (t % NT) * 16 + (lane & 15));
    }

    f32x4 acc[R][NT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KGW; ++i) {
      const int sl = i % D;
      // issue order: the four k-slices outermost, so that consecutive MFMAs write different accumulators
      // (back-to-back MFMAs on one accumulator wait for each other's result: 60 % of the wave cycles of
      // this launch were issue stalls with the slices innermost)
#define TTSX_SLICE(C)                                                                                        \
  _Pragma("unroll") for (int r = 0; r < R; ++r) _Pragma("unroll") for (int n = 0; n < NT; ++n) acc[r][n] =  \
      __builtin_amdgcn_mfma_f32_16x16x4f32(wv[sl][r].C, xv[sl][n].C, acc[r][n], 0, 0, 0);
      TTSX_SLICE(x)
      TTSX_SLICE(y)
      TTSX_SLICE(z)
      TTSX_SLICE(w)
#undef TTSX_SLICE
      if (!LATE && i + D < KGW) {
#pragma unroll
        for (int r = 0; r < R; ++r) wv[sl][r] = W[(size_t)(i + D) * GEMM_WAVES * 64 + r * wrs];
        const float4* xp = xptr(wave + (i + D) * GEMM_WAVES);
#pragma unroll
        for (int n = 0; n < NT; ++n) xv[sl][n] = xp[n * 16];
      }
    }
    if (KGW == 20) FA_MARK(9);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int n = 0; n < NT; ++n) red[wave][r * NT + n][lane] = acc[r][n];
    __syncthreads();
    if (KGW == 20) FA_MARK(10);
    if (done) return;

    // wave w finishes tile w: lane l holds rows 4*(l>>4)+{0..3} of column l&15
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int t = wave + k * GEMM_WAVES;
      if (t >= TILES) break;
      f32x4 v = red[0][t][lane];
#pragma unroll
      for (int w = 1; w < GEMM_WAVES; ++w) v += red[w][t][lane];
      const float h =
          epilogue_quad<MODE>(p, (rb + t / NT) * 4 + (lane >> 4), b0 + (t % NT) * 16 + (lane & 15), v, pre[k], kpart);
      if (MODE == MODE_LSTM || MODE == MODE_GRU) s_h[(t / NT) * 4 + (lane >> 4)][(t % NT) * 16 + (lane & 15)] = h;
    }
    if (KGW == 20) FA_MARK(11);
    if (MODE == MODE_PROJ && R == 1 && NT <= GEMM_WAVES && p.x1_flags) {
      // (one row block: wave t alone has written column tile t) the stores have reached memory, then the flag
      const int xb = rb - (p.o_quads / 4 + 1);
      if (xb >= 0 && wave < NT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
          __hip_atomic_store(p.x1_flags + (b0 / 16 + wave) * (PRENET_DIM / 16) + xb,
                             (unsigned)(p.ctx->base + p.step_in_chunk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (qshare) {
      __syncthreads();
      const int d = threadIdx.x & (ATT_DIM - 1), c0 = (threadIdx.x >> 7) * QC;
      float* dst = p.qpart + ((size_t)(rb / R) * p.Bpad + b0 + c0) * ATT_DIM + d;
#pragma unroll
      for (int c = 0; c < QC; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < QU; ++j) acc += wq[j] * s_h[j][c0 + c];
        dst[(size_t)c * ATT_DIM] = acc;
      }
    }
  }
};

// An LDS-only barrier: __syncthreads() carries a fence that waits for every outstanding global load, which would
// serialise loads that were issued early on purpose.
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Piece of early LSTM columns at padded batch 64 with the activations staged in LDS (MODE_PARTIAL only): a workgroup
// is (8 row blocks = 128 gate rows, one wave each) x (one of PARTS parts of K) x all 64 columns.  The part's
// activation slice (contiguous in the quad layout) is read once per workgroup and shared by its eight waves through
// LDS; every wave streams the weights of its own row block into registers and owns its rows over the whole part, so
// nothing is combined across waves, and once its operands arrive a workgroup keeps the matrix pipes of its CU
// full.  That is also why such a workgroup must not share a CU with a workgroup of the serial path (measured: the
// fused attention workgroup of an utterance takes 21 us instead of 11 next to one - its own matrix instructions
// and its loads queue behind the piece's): the launch asks for more than half a CU's LDS per workgroup (the slice
// of the longest part is 88 KB), so every workgroup gets a CU of its own, and the K range is cut into as many parts
// as leave one workgroup per remaining CU (decoder LSTM: 128 k-groups in 6 parts = 192 workgroups beside the 64
// attention workgroups).  Parts hold an even number of k-groups: KB, or KB + 2 for the first NX parts.
// Request order: every wave's share of the slice, a barrier (issue only, nothing is waited for), then weights, eight
// k-groups ahead of their use; the slice lands in chunks of four k-groups and the matrix instructions of a chunk
// start when it is in LDS.  Part k goes to gates + k * gates_part_stride; the late LSTM launch adds the parts.
template <int KG_TOTAL, int PARTS>
struct StagedPieceOp {
  typedef GemmP P;
  static constexpr int KB = (KG_TOTAL / PARTS) & ~1, NX = (KG_TOTAL - PARTS * KB) / 2, KMAX = NX ? KB + 2 : KB;
  static constexpr int CH = 4, WD = 8, NJ = KMAX / 2;
  static constexpr int LDS_BYTES = KMAX * 256 * (int)sizeof(float4);
  static_assert(KB % CH == 0 && KB >= WD && NX <= PARTS && KB * PARTS + 2 * NX == KG_TOTAL, "parts of K");
  static __device__ __forceinline__ void run(const GemmP& p, int wg) {
    // (native vector types: copies of the float4 struct between memory spaces compile to block copies through a
    // per-lane array, not to registers)
    extern __shared__ f32x4 staged_lds[];
    const int done = step_done(p.ctx, p.step_in_chunk);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nrg = p.nrb / GEMM_WAVES;
    const int part = p.kpart0 + wg / nrg, rb = p.rb0 + (wg % nrg) * GEMM_WAVES + wave;
    const int kg_first = part * KB + 2 * min(part, NX);
    const bool tail = part < NX;  // this part has KB + 2 k-groups
    // the slice: thread tid owns float4 tid of every pair of k-groups (a k-group is 4 quads x 64 columns)
    const f32x4* X0 = reinterpret_cast<const f32x4*>(seg_ptr(p.seg[0], p.parity)) + tid;
    const f32x4* X1 = p.nseg > 1 ? reinterpret_cast<const f32x4*>(seg_ptr(p.seg[1], p.parity)) + tid : X0;
    const int e0 = p.seg[0].kgroups;
    f32x4 xr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      // (segments and parts hold an even number of k-groups; a part without the last pair reads its own last pair
      // again - a load inside a branch would wait for its own round trip)
      const int g = kg_first + 2 * ((j < KB / 2 || tail) ? j : KB / 2 - 1);
      xr[j] = g < e0 ? X0[(size_t)g * 256] : X1[(size_t)(g - e0) * 256];
    }
    asm volatile("s_barrier" ::: "memory");
    const f32x4* W = reinterpret_cast<const f32x4*>(p.Wp) + ((size_t)rb * p.kg_stride + p.kg0 + kg_first) * 64 + lane;
    const int last = tail ? KMAX - 1 : KB - 1;
    f32x4 wv[KMAX];
#pragma unroll
    for (int i = 0; i < WD; ++i) wv[i] = W[(size_t)i * 64];
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4* xl = staged_lds + (lane >> 4) * 64 + (lane & 15);
#define TTSX_SLICE(C) \
  _Pragma("unroll") for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i].C, xv[n].C, acc[n], 0, 0, 0);
#define TTSX_KGROUP(i_)                                                          \
  {                                                                              \
    const int i = (i_);                                                          \
    f32x4 xv[4];                                                                 \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) xv[n] = xl[i * 256 + n * 16];   \
    TTSX_SLICE(x) TTSX_SLICE(y) TTSX_SLICE(z) TTSX_SLICE(w)                      \
  }
#pragma unroll
    for (int c = 0; c < KB / CH; ++c) {
#pragma unroll
      for (int jj = 0; jj < CH / 2; ++jj) staged_lds[tid + (c * (CH / 2) + jj) * GEMM_THREADS] = xr[c * (CH / 2) + jj];
      if (c == KB / CH - 1 && NX) staged_lds[tid + (KB / 2) * GEMM_THREADS] = xr[KB / 2];  // (the last pair rides along)
      lds_only_barrier();
#pragma unroll
      for (int ii = 0; ii < CH; ++ii) TTSX_KGROUP(c * CH + ii)
#pragma unroll
      for (int ii = 0; ii < CH; ++ii)
        if (c * CH + WD + ii < KMAX) wv[c * CH + WD + ii] = W[(size_t)min(c * CH + WD + ii, last) * 64];
    }
    if (NX && tail) {
      TTSX_KGROUP(KB)
      TTSX_KGROUP(KB + 1)
    }
#undef TTSX_KGROUP
#undef TTSX_SLICE
    if (done) return;
    const size_t o0 = (size_t)part * p.gates_part_stride + (size_t)(rb * 4 + (lane >> 4)) * p.Bpad + (lane & 15);
    if (p.late_flags) {
      // read by the late decoder-LSTM workgroups of this launch: past the L2, then this wave's step tag
#pragma unroll
      for (int n = 0; n < 4; ++n) coh_store16(p.gates, (unsigned)((o0 + n * 16) * sizeof(float4)), acc[n]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0)
        __hip_atomic_store(p.late_flags + 2 * p.Bpad + ((part * nrg + wg % nrg) * GEMM_WAVES + wave),
                           (unsigned)(p.ctx->base + p.step_in_chunk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    float4* O = reinterpret_cast<float4*>(p.gates) + o0;
#pragma unroll
    for (int n = 0; n < 4; ++n) O[n * 16] = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
  }
};

// GEMV engine for NB <= 2 utterances (Bpad == NB): every weight load of a wave is issued before
// the first use, activations come from LDS.  KGW = k-groups per wave (KG / 8).
// KP > 1 (PARTIAL pieces): the K range is cut into KP parts of KGW*8 k-groups and the launch runs one block of
// workgroups per part (part k writes gates + k * gates_part_stride), so that a piece of a few row blocks still
// spreads its weight stream over many CUs.
template <int NB, int KGW, int MODE, int KP = 1>
struct GemvOp {
  typedef GemmP P;
  static __device__ __forceinline__ void run(const GemmP& p, int wg) {
    const int done = step_done(p.ctx, p.step_in_chunk);
    int kpart = 0;
    if (KP > 1) kpart = wg / p.nrb, wg -= kpart * p.nrb, kpart += p.kpart0;
    const int kbase = kpart * (KGW * GEMM_WAVES);
    const int rb = p.rb0 + wg, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    constexpr int KG = KGW * GEMM_WAVES * KP;
    __shared__ float4 s_x[KG * 4 * NB];  // quad-interleaved: [quad][b]
    __shared__ float s_red[GEMM_WAVES][NB][16];

    // issue the weight stream first
    const float4* W =
        reinterpret_cast<const float4*>(p.Wp) + ((size_t)rb * p.kg_stride + p.kg0 + kbase) * 64 + lane;
    float4 wv[KGW];
#pragma unroll
    for (int i = 0; i < KGW; ++i) wv[i] = W[(size_t)(wave + i * GEMM_WAVES) * 64];

    // stage the activations (segments are already quad-interleaved with Bpad == NB)
    {
      int q0 = 0;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        if (s < p.nseg) {
          const float4* X = reinterpret_cast<const float4*>(seg_ptr(p.seg[s], p.parity));
          const int n = p.seg[s].kgroups * 4 * NB;
          for (int i = tid; i < n; i += GEMM_THREADS) s_x[q0 + i] = X[i];
          q0 += n;
        }
      }
    }
    const bool qshare = (MODE == MODE_LSTM || MODE == MODE_GRU) && p.qpart != nullptr;
    float wq[4];
    if (qshare && tid < ATT_DIM * NB) {
#pragma unroll
      for (int j = 0; j < 4; ++j) wq[j] = p.wq_t[(size_t)(rb * 4 + j) * ATT_DIM + (tid & (ATT_DIM - 1))];
    }
    __shared__ float s_hq[4][NB];
    // epilogue operands: thread (qi, b) finishes quad qi of the row block for utterance b
    const int qi = tid / NB, eb = tid % NB, quad_e = rb * 4 + qi;
    Pre pre;
    if (tid < 4 * NB) pre = prefetch_quad<MODE>(p, quad_e, eb);
    __syncthreads();

    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    const int kq = lane >> 4;
#pragma unroll
    for (int i = 0; i < KGW; ++i) {
      const int quad = (kbase + wave + i * GEMM_WAVES) * 4 + kq;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 xv = s_x[quad * NB + b];
        acc[b] += wv[i].x * xv.x + wv[i].y * xv.y + wv[i].z * xv.z + wv[i].w * xv.w;
      }
    }
    // lanes l, l^16, l^32, l^48 hold partial sums of the same row
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      acc[b] += __shfl_xor(acc[b], 16, 64);
      acc[b] += __shfl_xor(acc[b], 32, 64);
      if (lane < 16) s_red[wave][b][lane] = acc[b];
    }
    __syncthreads();
    if (done) return;
    if (tid < 4 * NB) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < GEMM_WAVES; ++w) t += s_red[w][eb][qi * 4 + j];
        v[j] = t;
      }
      const float h = epilogue_quad<MODE>(p, quad_e, eb, v, pre, kpart);
      if (MODE == MODE_LSTM || MODE == MODE_GRU) s_hq[qi][eb] = h;
    }
    if (qshare) {
      __syncthreads();
      if (tid < ATT_DIM * NB) {
        const int d = tid & (ATT_DIM - 1), bb = tid >> 7;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += wq[j] * s_hq[j][bb];
        p.qpart[((size_t)rb * NB + bb) * ATT_DIM + d] = acc;
      }
    }
  }
};

// Nothing: the second slot of a launch that carries no partial piece.
struct NoOp {
  typedef int P;
  static __device__ __forceinline__ void run(const int&, int) {}
};

// A launch: workgroups [0, n_first) run op A (the one on the serial path, scheduled first), the
// rest run op B (a piece of partial LSTM work).
template <class A, class B>
__global__ __launch_bounds__(GEMM_THREADS) void dual_kernel(typename A::P pa, typename B::P pb, int n_first) {
  WG_T0();
  if ((int)blockIdx.x < n_first) A::run(pa, blockIdx.x);
  else B::run(pb, blockIdx.x - n_first);
  WG_MARK(n_first);
}

// Prenet layer 2 of the NEXT step inside the projection launch of this one (MFMA engine): one launch boundary
// less per step.  Workgroup (row block rb2 of 16, column tile ct) waits until the 16 projection workgroups that
// produce prenet layer 1 for its 16 utterances have raised their flags to this step's tag, then wave 0 reads the
// x1 tile past the L2 (the producers may sit on other XCDs), multiplies (64 MFMAs) and writes x2 for the next
// launch.  The projection workgroups have the lower block indices, so they are resident before a waiting
// workgroup can occupy a slot; the wait is bounded all the same.
struct Prenet2AfterProjOp {
  typedef GemmP P;  // Wp / bias: prenet layer 2; seg[0].x[0]: x1; out[0]: x2; x1_flags; ctx, step_in_chunk of THIS step
  static __device__ __forceinline__ void run(const GemmP& p, int wg) {
    if (threadIdx.x >= 64) return;
    __builtin_amdgcn_s_setprio(3);
    RunCtx* cx = p.ctx;
    if (step_done(cx, p.step_in_chunk)) return;  // (the producers skip their epilogue under the same test)
    const int lane = threadIdx.x;
    constexpr int NRB = PRENET_DIM / 16, KG = PRENET_DIM / 16;
    const int rb = wg % NRB, ct = wg / NRB, b0 = ct * 16;
    const unsigned tag = (unsigned)(cx->base + p.step_in_chunk + 1);
    const float4* W = reinterpret_cast<const float4*>(p.Wp) + (size_t)rb * KG * 64 + lane;
    float4 wv[KG];
#pragma unroll
    for (int i = 0; i < KG; ++i) wv[i] = W[(size_t)i * 64];
    const float4 bias = reinterpret_cast<const float4*>(p.bias)[rb * 4 + (lane >> 4)];
    bool ok = false;
    for (long spin = 0; spin < 4000000; ++spin) {
      const unsigned f = lane < NRB ? __hip_atomic_load(p.x1_flags + ct * NRB + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
      if (__all(f == tag)) {
        ok = true;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) {
      if (lane == 0) cx->sync_failed = 1;
      return;
    }
    asm volatile("" ::: "memory");  // (the compiler keeps the loads below behind the flag loads)
    // (No acquire fence here and no release store on the producer's side: at agent scope they invalidate / write
    // back the L2 - measured +0.7 us on this launch.  x1 and the flags only move through agent-scope accesses that
    // bypass the caches which such fences would maintain; the producer waits for its x1 stores to be acknowledged
    // (vmcnt) before it raises the flag, and the x1 loads below are issued after the flag loads have returned.)
    // B fragments: lane l holds x1[16 kg + 4 (l >> 4) + s][b0 + (l & 15)], s = 0..3 (one quad of the quad layout)
    const unsigned long long* X = reinterpret_cast<const unsigned long long*>(p.seg[0].x[0]) +
                                  ((size_t)(lane >> 4) * p.Bpad + b0 + (lane & 15)) * 2;
    unsigned long long xr[KG][2];
#pragma unroll
    for (int i = 0; i < KG; ++i) {
      xr[i][0] = __hip_atomic_load(X + (size_t)i * 4 * p.Bpad * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      xr[i][1] = __hip_atomic_load(X + (size_t)i * 4 * p.Bpad * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The sums are formed in the order of the launch this replaces (GemmOp<1, 1, 2, MODE_LINEAR>: wave w adds
    // k-groups w and w + 8, the eight partial tiles are added in wave order, then the bias), so that the two
    // ways to run a step agree bit for bit.
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < GEMM_WAVES; ++w) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KG / GEMM_WAVES; ++i) {
        const int kg = w + GEMM_WAVES * i;
        a = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[kg].x, __uint_as_float((unsigned)xr[kg][0]), a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[kg].y, __uint_as_float((unsigned)(xr[kg][0] >> 32)), a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[kg].z, __uint_as_float((unsigned)xr[kg][1]), a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[kg].w, __uint_as_float((unsigned)(xr[kg][1] >> 32)), a, 0, 0, 0);
      }
      if (w == 0) v = a;
      else v += a;
    }
    v[0] += bias.x, v[1] += bias.y, v[2] += bias.z, v[3] += bias.w;
    const float4 o = make_float4(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
    reinterpret_cast<float4*>(p.out[0])[(size_t)(rb * 4 + (lane >> 4)) * p.Bpad + b0 + (lane & 15)] = o;
  }
};

// Workgroups [0, na) run op A, [na, na + nc) op C, the rest op B.
template <class A, class C, class B>
__global__ __launch_bounds__(GEMM_THREADS) void triple_kernel(typename A::P pa, typename C::P pc, typename B::P pb, int na,
                                                              int nc) {
  const int wg = blockIdx.x;
  if (wg < na) A::run(pa, wg);
  else if (wg < na + nc) C::run(pc, wg - na);
  else B::run(pb, wg - na - nc);
}

// The same with op B's workgroups dispatched before op C's (C: workgroups that start by waiting for op A's - they
// should not hold slots that B's could work in).
template <class A, class C, class B>
__global__ __launch_bounds__(GEMM_THREADS) void triple_kernel_b_first(typename A::P pa, typename C::P pc, typename B::P pb,
                                                                      int na, int nb) {
  WG_T0();
  const int wg = blockIdx.x;
  if (wg < na) A::run(pa, wg);
  else if (wg < na + nb) B::run(pb, wg - na);
  else C::run(pc, wg - na - nb);
  WG_MARK(na);
}

// ------------------------------------------------------------------ attention
struct AttnP {
  const float* q;        // quad [128]
  const float* P;        // [B][T][128]  processed_inputs
  const float* loc_conv_t;   // [2*31][32]   location_conv weight, filter index fastest
  const float* loc_dense_t;  // [8][128][4]  location_dense weight: quad of filters x dim, dim fastest
  const float* v_w;          // [128]
  const float* v_b;          // [1]
  float* a_prev;         // [B][T]
  float* a_cum;          // [B][T]
  float* energies;       // [B][T]
  float* ctx_out;        // quad [512]
  unsigned* late_flags;  // (fused kernel) see GemmP::late_flags; null: off
  const float* enc;      // [B][T][512] encoder outputs (the decoder's own copy: a stable address)
  const int32_t* len;    // [B] valid encoder positions per utterance
  RunCtx* ctx;
  int step_in_chunk, Bpad, norm, T;
  int gx;  // workgroups per utterance: workgroup id = b * gx + x
  // fused kernel (FusedAttnOp) only
  const float* q_bias;  // [128] constant added to the query (Tacotron v1: query + location-linear biases), or null
  const float* qpart;   // [n_qgroups][Bpad][128] shares of the query projection left by the attention LSTM
  int n_qgroups;
  const float* wlc_p;   // [2][16][64]  location_conv weight as MFMA A-operand fragments
  const float* wld_p;   // [8][8][64]   location_dense weight as MFMA A-operand fragments
  // optional attention variants (AttnPostOp / GravesOp only)
  int windowing, forward_attn, forward_attn_mask, trans_agent;
  int32_t* win_idx;     // [B] window centre of apply_windowing, -1 before the first step
  float* alpha;         // [B][T] forward-attention state
  float* u;             // [B] transition probability of forward attention
  const float* ta_w;    // [1536] transition agent: [context 512; query 1024]
  const float* ta_b;    // [1]
  const float* h_new;   // quad [1024]: attention-LSTM hidden state of this step (the "query")
  const float* gbk;     // quad [3K]: output of GravesAttention.N_a for this step
  float* mu;            // [B][K] means of the Graves mixture
  int graves_K;
};

constexpr int ATT_PAD = (LOC_K - 1) / 2;

// tanh through one hardware exponential: 1 - 2/(e^{2x}+1); saturates correctly at +-inf
__device__ __forceinline__ float fast_tanhf(float x) {
  const float t = __expf(2.f * x);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
}

// energies e[b][t] for a tile of EN_TT (32, or 16 for small batches: more workgroups) encoder positions:
//   cf[t][f] = sum_{c,j} W_lc[f][c][j] * a_c[t + j - 15]         (thread: filter f, EN_CP positions)
//   e[t] = b_v + sum_d v[d] * tanh(q[d] + sum_f W_ld[d][f] cf[t][f] + P[t][d])   (thread: dim d, EN_DP positions)
// Every global load of the kernel is issued before the first barrier, so the whole kernel costs
// one memory round trip.
template <int EN_TT>
struct EnergyOp {
  static constexpr int EN_CP = EN_TT / 16, EN_DP = EN_TT / 4;  // positions per thread in the conv / dense role
  typedef AttnP P;
  static __device__ __forceinline__ void run(const AttnP& p, int wg) {
  const int done = step_done(p.ctx, p.step_in_chunk);
  const int T = p.T, b = wg / p.gx, t0 = (wg % p.gx) * EN_TT, tid = threadIdx.x;
  __shared__ float s_a[2][EN_TT + 2 * ATT_PAD + 2];
  __shared__ __attribute__((aligned(16))) float s_cf[EN_TT][LOC_F];
  __shared__ float s_part[2][EN_TT];

  // ---- loads: alignment window (124 threads), conv weights, dense weights, P, q, v
  float a_in = 0.f;
  if (tid < 2 * (EN_TT + 2 * ATT_PAD)) {
    const int c = tid / (EN_TT + 2 * ATT_PAD), i = tid % (EN_TT + 2 * ATT_PAD);
    const int t = t0 + i - ATT_PAD;
    const float* src = c == 0 ? p.a_prev : p.a_cum;
    if (t >= 0 && t < T) a_in = src[(size_t)b * T + t];
  }
  const int f = tid & 31, tp = tid >> 5;  // conv role: filter f, positions EN_CP tp .. EN_CP tp + EN_CP - 1
  float wl[2 * LOC_K];
#pragma unroll
  for (int j = 0; j < 2 * LOC_K; ++j) wl[j] = p.loc_conv_t[j * LOC_F + f];
  const int d = tid & 127, tg = tid >> 7;  // dense role: dim d, positions tg*EN_DP .. tg*EN_DP+EN_DP-1
  float wd[LOC_F];
  {
    const float4* wsrc = reinterpret_cast<const float4*>(p.loc_dense_t) + d;
#pragma unroll
    for (int i = 0; i < LOC_F / 4; ++i) {
      const float4 t4 = wsrc[i * ATT_DIM];
      wd[4 * i] = t4.x, wd[4 * i + 1] = t4.y, wd[4 * i + 2] = t4.z, wd[4 * i + 3] = t4.w;
    }
  }
  float pin[EN_DP];
#pragma unroll
  for (int i = 0; i < EN_DP; ++i) {
    const int t = t0 + tg * EN_DP + i;
    pin[i] = t < T ? p.P[((size_t)b * T + t) * ATT_DIM + d] : 0.f;
  }
  const float vd = p.v_w[d], vb = p.v_b[0];
  // The query either comes from the query launch, or (p.qpart) as the shares the attention-LSTM workgroups left:
  // thread (d, tg) adds the groups tg, tg + 4, ... (every load issued before the first use; clamped and masked,
  // not conditional), the four partial sums meet in LDS in a fixed order.
  __shared__ float s_qs[4][ATT_DIM];
  float qd = 0.f;
  if (p.qpart) {
    constexpr int QU = QUERY_DIM / 4 / 4;  // at most 256 groups
    const float* qp = p.qpart + (size_t)b * ATT_DIM + d;
    const size_t gs = (size_t)p.Bpad * ATT_DIM;
    float qv[QU];
#pragma unroll
    for (int u = 0; u < QU; ++u) qv[u] = qp[(size_t)min(tg + 4 * u, p.n_qgroups - 1) * gs];
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < QU; ++u) acc += tg + 4 * u < p.n_qgroups ? qv[u] : 0.f;
    s_qs[tg][d] = acc;
  } else {
    qd = p.q[qidx(d, b, p.Bpad)];
  }

  if (tid < 2 * (EN_TT + 2 * ATT_PAD)) s_a[tid / (EN_TT + 2 * ATT_PAD)][tid % (EN_TT + 2 * ATT_PAD)] = a_in;
  __syncthreads();
  if (p.qpart) qd = ((s_qs[0][d] + s_qs[1][d]) + (s_qs[2][d] + s_qs[3][d])) + (p.q_bias ? p.q_bias[d] : 0.f);
  {
    float acc[EN_CP];
#pragma unroll
    for (int u = 0; u < EN_CP; ++u) acc[u] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float win[LOC_K + EN_CP - 1];
#pragma unroll
      for (int j = 0; j < LOC_K + EN_CP - 1; ++j) win[j] = s_a[c][tp * EN_CP + j];
#pragma unroll
      for (int j = 0; j < LOC_K; ++j)
#pragma unroll
        for (int u = 0; u < EN_CP; ++u) acc[u] += wl[c * LOC_K + j] * win[j + u];
    }
#pragma unroll
    for (int u = 0; u < EN_CP; ++u) s_cf[tp * EN_CP + u][f] = acc[u];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < EN_DP; ++i) {
    const int tl = tg * EN_DP + i;
    const float4* cf4 = reinterpret_cast<const float4*>(&s_cf[tl][0]);
    float loc = 0.f;
#pragma unroll
    for (int k = 0; k < LOC_F / 4; ++k) {
      const float4 c4 = cf4[k];
      loc += wd[4 * k] * c4.x + wd[4 * k + 1] * c4.y + wd[4 * k + 2] * c4.z + wd[4 * k + 3] * c4.w;
    }
    float e = vd * fast_tanhf(qd + loc + pin[i]);
    e = wave_sum(e);
    if ((tid & 63) == 0) s_part[(tid >> 6) & 1][tl] = e;
  }
  __syncthreads();
  if (!done && tid < EN_TT && t0 + tid < T)
    p.energies[(size_t)b * T + t0 + tid] = s_part[0][tid] + s_part[1][tid] + vb;
  }
};

// normalisation over the encoder axis (every workgroup of an utterance repeats it: T exps) and a
// slice of CW context columns: ctx[c] = sum_t a[t] enc[t][c].  Workgroup x == 0 also writes the
// alignment outputs and the location-attention state.  The first CTX_PRE rows of encoder outputs
// a thread needs are loaded before the normalisation, so that round trip overlaps it.
constexpr int CTX_PRE = 8, CTX_THREADS = GEMM_THREADS;
template <int CW>
struct ContextOp {
  typedef AttnP P;
  static __device__ __forceinline__ void run(const AttnP& p, int wg) {
  RunCtx* cx = p.ctx;
  const int done = step_done(cx, p.step_in_chunk);
  const int T = p.T, b = wg / p.gx, bx = wg % p.gx, tid = threadIdx.x;
  const int len = p.len[b];
  extern __shared__ float s_dyn[];
  float* s_al = s_dyn;  // [T]
  __shared__ float s_red[CTX_THREADS / 64];
  __shared__ float4 s_acc[CTX_THREADS];

  // context slice geometry: lane group = CW/4 float4 columns, row groups stride over t
  constexpr int LPR = CW / 4, RG = CTX_THREADS / LPR;
  const int cl = tid % LPR, rg = tid / LPR;
  const float4* e4 = reinterpret_cast<const float4*>(p.enc + (size_t)b * T * ENC_DIM + bx * CW) + cl;
  float4 pre[CTX_PRE];
#pragma unroll
  for (int i = 0; i < CTX_PRE; ++i) {
    const int t = rg + i * RG;
    pre[i] = t < len ? e4[(size_t)t * (ENC_DIM / 4)] : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  const bool softmax = p.norm == TTSX_NORM_SOFTMAX;
  float m = -INFINITY;
  if (softmax) {
    for (int t = tid; t < len; t += CTX_THREADS) m = fmaxf(m, p.energies[(size_t)b * T + t]);
    m = wave_max(m);
    if ((tid & 63) == 0) s_red[tid >> 6] = m;
    __syncthreads();
    m = s_red[0];
#pragma unroll
    for (int w = 1; w < CTX_THREADS / 64; ++w) m = fmaxf(m, s_red[w]);
    __syncthreads();
  }
  float sum = 0.f;
  for (int t = tid; t < T; t += CTX_THREADS) {
    float v = 0.f;
    if (t < len) {
      const float e = p.energies[(size_t)b * T + t];
      v = softmax ? expf(e - m) : sigmoidf_(e);
    }
    s_al[t] = v;
    sum += v;
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) s_red[tid >> 6] = sum;
  __syncthreads();
  sum = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) + ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7]));
  for (int t = tid; t < T; t += CTX_THREADS) s_al[t] = s_al[t] / sum;
  __syncthreads();
  if (done) return;

  if (bx == 0) {
    const int row = cx->out_base + p.step_in_chunk;
    for (int t = tid; t < T; t += CTX_THREADS) {
      const float a = s_al[t];
      p.a_prev[(size_t)b * T + t] = a;
      p.a_cum[(size_t)b * T + t] += a;
      cx->align[((size_t)b * cx->align_rows + row) * T + t] = a;
    }
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < CTX_PRE; ++i) {
    const int t = rg + i * RG;
    const float a0 = t < len ? s_al[t] : 0.f;
    acc.x += a0 * pre[i].x, acc.y += a0 * pre[i].y, acc.z += a0 * pre[i].z, acc.w += a0 * pre[i].w;
  }
  for (int t = rg + CTX_PRE * RG; t < len; t += RG) {
    const float4 v0 = e4[(size_t)t * (ENC_DIM / 4)];
    const float a0 = s_al[t];
    acc.x += a0 * v0.x, acc.y += a0 * v0.y, acc.z += a0 * v0.z, acc.w += a0 * v0.w;
  }
  s_acc[tid] = acc;
  __syncthreads();
  if (tid < LPR) {
    float4 tot = s_acc[tid];
    for (int g = 1; g < RG; ++g) {
      const float4 v = s_acc[g * LPR + tid];
      tot.x += v.x, tot.y += v.y, tot.z += v.z, tot.w += v.w;
    }
    const int quad = bx * LPR + tid;
    reinterpret_cast<float4*>(p.ctx_out)[(size_t)quad * p.Bpad + b] = tot;
  }
  }
};

// ---- workgroup-wide reductions for the one-workgroup-per-utterance attention variants
// (`s_red` / `s_redi`: one slot per wave; every thread gets the result)
__device__ __forceinline__ float block_sum(float v, float* s_red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < CTX_THREADS / 64; ++w) t += s_red[w];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* s_red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = s_red[0];
  for (int w = 1; w < CTX_THREADS / 64; ++w) t = fmaxf(t, s_red[w]);
  return t;
}
// index of the first maximum of x[0..n) (what torch.argmax / torch.max(dim) return on the CPU)
__device__ __forceinline__ int block_argmax_first(const float* x, int n, float* s_red, int* s_redi, float* max_out) {
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int t = threadIdx.x; t < n; t += CTX_THREADS) {
    const float v = x[t];
    if (v > bv || (v == bv && t < bi)) bv = v, bi = t;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) bv = ov, bi = oi;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = bv, s_redi[threadIdx.x >> 6] = bi;
  __syncthreads();
  bv = s_red[0], bi = s_redi[0];
  for (int w = 1; w < CTX_THREADS / 64; ++w)
    if (s_red[w] > bv || (s_red[w] == bv && s_redi[w] < bi)) bv = s_red[w], bi = s_redi[w];
  if (max_out) *max_out = bv;
  return bi;
}

// context of one utterance by one workgroup: ctx[c] = sum_t a[t] enc[t][c] for all 512 columns,
// written in quad layout and left in s_ctx[512]
__device__ __forceinline__ void block_context(const AttnP& p, int b, int len, const float* s_al, float4* s_acc,
                                              float* s_ctx) {
  const int tid = threadIdx.x, T = p.T;
  constexpr int LPR = ENC_DIM / 4, RG = CTX_THREADS / LPR;
  const int cl = tid % LPR, rg = tid / LPR;
  const float4* e4 = reinterpret_cast<const float4*>(p.enc + (size_t)b * T * ENC_DIM) + cl;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // (eight independent row loads per pass: a rolled loop would pay one memory round trip per row)
  for (int t0 = rg; t0 < len; t0 += 8 * RG) {
    float4 v0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + u * RG;
      v0[u] = t < len ? e4[(size_t)t * (ENC_DIM / 4)] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = t0 + u * RG;
      const float a0 = t < len ? s_al[t] : 0.f;
      acc.x += a0 * v0[u].x, acc.y += a0 * v0[u].y, acc.z += a0 * v0[u].z, acc.w += a0 * v0[u].w;
    }
  }
  __syncthreads();
  s_acc[tid] = acc;
  __syncthreads();
  if (tid < LPR) {
    float4 tot = s_acc[tid];
    for (int g = 1; g < RG; ++g) {
      const float4 v = s_acc[g * LPR + tid];
      tot.x += v.x, tot.y += v.y, tot.z += v.z, tot.w += v.w;
    }
    reinterpret_cast<float4*>(p.ctx_out)[(size_t)tid * p.Bpad + b] = tot;
    reinterpret_cast<float4*>(s_ctx)[tid] = tot;
  }
  __syncthreads();
}

// Everything OriginalAttention.forward does after the energies, with its optional branches
// (Tacotron2/TTS/models.py:328-361): length mask, apply_windowing (:275-288), normalisation,
// cumulative weights, apply_forward_attention (:290-312) with the transition agent (:357-360),
// context.  One workgroup per utterance; every utterance keeps its own window index (the
// reference evaluates it on utterance 0 and only runs at batch 1).
struct AttnPostOp {
  typedef AttnP P;
  static __device__ __forceinline__ void run(const AttnP& p, int wg) {
    RunCtx* cx = p.ctx;
    const int done = step_done(cx, p.step_in_chunk);
    const int T = p.T, b = wg, tid = threadIdx.x;
    const int len = p.len[b];
    extern __shared__ float s_dyn[];
    float* s_e = s_dyn;           // [T] energies, then scratch
    float* s_al = s_dyn + T;      // [T] alignment
    float* s_old = s_dyn + 2 * T; // [T] previous forward-attention weights
    __shared__ float s_red[CTX_THREADS / 64];
    __shared__ int s_redi[CTX_THREADS / 64];
    __shared__ float4 s_acc[CTX_THREADS];
    __shared__ __attribute__((aligned(16))) float s_ctx[ENC_DIM];

    for (int t = tid; t < T; t += CTX_THREADS) s_e[t] = t < len ? p.energies[(size_t)b * T + t] : -INFINITY;
    __syncthreads();
    if (p.windowing) {
      const int w = p.win_idx[b], back = w - 2, front = w + 6;
      for (int t = tid; t < len; t += CTX_THREADS)
        if ((back > 0 && t < back) || (front < len && t >= front)) s_e[t] = -INFINITY;
      __syncthreads();
      if (w == -1) {
        float m = -INFINITY;
        for (int t = tid; t < len; t += CTX_THREADS) m = fmaxf(m, s_e[t]);
        m = block_max(m, s_red);
        if (tid == 0) s_e[0] = m;
        __syncthreads();
      }
      const int idx = block_argmax_first(s_e, len, s_red, s_redi, nullptr);
      if (tid == 0 && !done) p.win_idx[b] = idx;
    }
    const bool softmax = p.norm == TTSX_NORM_SOFTMAX;
    float m = 0.f;
    if (softmax) {
      m = -INFINITY;
      for (int t = tid; t < len; t += CTX_THREADS) m = fmaxf(m, s_e[t]);
      m = block_max(m, s_red);
    }
    float sum = 0.f;
    for (int t = tid; t < T; t += CTX_THREADS) {
      float v = 0.f;
      if (t < len) v = softmax ? expf(s_e[t] - m) : sigmoidf_(s_e[t]);
      s_al[t] = v;
      sum += v;
    }
    sum = block_sum(sum, s_red);
    for (int t = tid; t < T; t += CTX_THREADS) {
      const float a = s_al[t] / sum;
      s_al[t] = a;
      if (!done) p.a_cum[(size_t)b * T + t] += a;
    }
    __syncthreads();
    if (p.forward_attn) {
      const float u = p.u[b];
      for (int t = tid; t < T; t += CTX_THREADS) s_old[t] = t < len ? p.alpha[(size_t)b * T + t] : 0.f;
      __syncthreads();
      // s_e <- shifted weights, s_al <- transition potentials
      float vmax = -INFINITY;
      for (int t = tid; t < len; t += CTX_THREADS) {
        const float sh = t == 0 ? 0.f : s_old[t - 1];
        s_e[t] = sh;
        const float v = ((1.f - u) * s_old[t] + u * sh + 1e-8f) * s_al[t];
        s_al[t] = v;
        vmax = fmaxf(vmax, v);
      }
      __syncthreads();
      if (p.forward_attn_mask) {
        const float val = block_max(vmax, s_red);
        const int n = block_argmax_first(s_e, len, s_red, s_redi, nullptr);
        // alpha[n+3:] = 0; alpha[:n-1] = 0; alpha[n-2] = 0.01 * val, with Python's index rules
        // (a negative stop / index counts from the end)
        int k1 = n - 1, k2 = n - 2;
        if (k1 < 0) k1 = max(k1 + len, 0);
        if (k2 < 0) k2 += len;
        for (int t = tid; t < len; t += CTX_THREADS)
          if (t >= n + 3 || t < k1) s_al[t] = 0.f;
        __syncthreads();
        if (tid == 0 && k2 >= 0 && k2 < len) s_al[k2] = 0.01f * val;
        __syncthreads();
      }
      float s2 = 0.f;
      for (int t = tid; t < len; t += CTX_THREADS) s2 += s_al[t];
      s2 = block_sum(s2, s_red);
      for (int t = tid; t < len; t += CTX_THREADS) {
        const float a = s_al[t] / s2;
        s_al[t] = a;
        if (!done) p.alpha[(size_t)b * T + t] = a;
      }
      __syncthreads();
    }
    if (!done) {
      const int row = cx->out_base + p.step_in_chunk;
      for (int t = tid; t < T; t += CTX_THREADS) {
        const float a = s_al[t];
        p.a_prev[(size_t)b * T + t] = a;
        cx->align[((size_t)b * cx->align_rows + row) * T + t] = a;
      }
    }
    if (done) return;
    block_context(p, b, len, s_al, s_acc, s_ctx);
    if (p.forward_attn && p.trans_agent) {
      float acc = 0.f;
      for (int k = tid; k < ENC_DIM + QUERY_DIM; k += CTX_THREADS)
        acc += p.ta_w[k] * (k < ENC_DIM ? s_ctx[k] : p.h_new[qidx(k - ENC_DIM, b, p.Bpad)]);
      acc = block_sum(acc, s_red);
      if (tid == 0) p.u[b] = sigmoidf_(acc + p.ta_b[0]);
    }
  }
};

// torch.nn.functional.softplus (beta 1, threshold 20)
__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// GravesAttention.forward after the N_a network (Tacotron2/TTS/models.py:151-183): mixture
// parameters from gbk = N_a(query) viewed [3][K], discretised weights
//   alpha[t] = cdf[t+1] - cdf[t],  cdf[j] = sum_k g_k / (1 + sigmoid((mu_k - (j + 0.5)) / sig_k)),
// exact zeros replaced by 1e-8, context.  One workgroup per utterance; positions beyond the
// utterance's length get weight 0 (a batch-1 run of the unpadded text does not have them).
constexpr int GRAVES_MAX_K = 16;
struct GravesOp {
  typedef AttnP P;
  static __device__ __forceinline__ void run(const AttnP& p, int wg) {
    RunCtx* cx = p.ctx;
    const int done = step_done(cx, p.step_in_chunk);
    const int T = p.T, b = wg, tid = threadIdx.x, K = p.graves_K;
    const int len = p.len[b];
    extern __shared__ float s_dyn[];
    float* s_al = s_dyn;          // [T]
    float* s_cdf = s_dyn + T;     // [T+1]
    __shared__ float s_g[GRAVES_MAX_K], s_sig[GRAVES_MAX_K], s_mu[GRAVES_MAX_K];
    __shared__ float4 s_acc[CTX_THREADS];
    __shared__ __attribute__((aligned(16))) float s_ctx[ENC_DIM];
    if (tid < K) {
      s_g[tid] = p.gbk[qidx(tid, b, p.Bpad)];
      s_sig[tid] = softplusf_(p.gbk[qidx(K + tid, b, p.Bpad)]) + 1e-5f;
      s_mu[tid] = p.mu[b * K + tid] + softplusf_(p.gbk[qidx(2 * K + tid, b, p.Bpad)]);
    }
    __syncthreads();
    if (tid == 0) {
      float m = -INFINITY, s = 0.f;
      for (int k = 0; k < K; ++k) m = fmaxf(m, s_g[k]);
      for (int k = 0; k < K; ++k) s += expf(s_g[k] - m);
      for (int k = 0; k < K; ++k) s_g[k] = expf(s_g[k] - m) / s + 1e-5f;
    }
    __syncthreads();
    for (int j = tid; j <= len; j += CTX_THREADS) {
      float c = 0.f;
      for (int k = 0; k < K; ++k) c += s_g[k] * (1.f / (1.f + sigmoidf_((s_mu[k] - ((float)j + 0.5f)) / s_sig[k])));
      s_cdf[j] = c;
    }
    __syncthreads();
    for (int t = tid; t < T; t += CTX_THREADS) {
      float a = 0.f;
      if (t < len) {
        a = s_cdf[t + 1] - s_cdf[t];
        if (a == 0.f) a = 1e-8f;
      }
      s_al[t] = a;
    }
    __syncthreads();
    if (done) return;
    if (tid < K) p.mu[b * K + tid] = s_mu[tid];
    const int row = cx->out_base + p.step_in_chunk;
    for (int t = tid; t < T; t += CTX_THREADS) {
      p.a_prev[(size_t)b * T + t] = s_al[t];
      cx->align[((size_t)b * cx->align_rows + row) * T + t] = s_al[t];
    }
    block_context(p, b, len, s_al, s_acc, s_ctx);
  }
};

// The whole attention of one utterance in ONE workgroup, for encoder lengths up to 512: query (sum of
// the shares the attention LSTM left), location features, energies, normalisation, alignment
// outputs, context.  It replaces three launches of the serial chain (query, energies, context).
// The two contractions run on the matrix cores, one wave per tile of 16 encoder positions:
//   cf[f][t]  = sum_k W_lc[f][k] X[k][t],  X[k][t] = a_c[t + j - 15] (k = 31 c + j): a Toeplitz
//               operand read straight from the zero-padded alignment rows in LDS  (32 MFMAs / tile)
//   loc[d][t] = sum_f W_ld[d][f] cf[f][t]                                            (64 MFMAs / tile)
// with the weights staged in LDS as A-operand fragments and cf handed from the first product to the
// second through a wave-private LDS tile.  NTT = tiles per wave (T <= 128 NTT); EDIM = encoder width
// (512 Tacotron2, 256 Tacotron).
// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup fence, which
// waits for every outstanding global load as well - including loads issued early on purpose.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int FA_TMAX = 512, FA_CF_PITCH = 36, FA_SA_LEN = FA_TMAX + 48;
constexpr int FA_LDS_FLOATS = 2048 + 4096 + 2 * FA_SA_LEN + GEMM_WAVES * 16 * FA_CF_PITCH + 512 + 128 + 128 + 2 * FA_TMAX;
// PRE: encoder rows per thread that are requested ahead of the context phase (0: all of them at one tile per
// wave, else 16).  With all 32 in flight the kernel needs 194 registers and its launch runs one workgroup per
// CU; PRE = 16 brings it to 119, so that LSTM pieces sharing the launch can run two workgroups per CU.
template <int NTT, int EDIM, int PRE = 0>
struct FusedAttnOp {
  typedef AttnP P;
  static __device__ __forceinline__ void run(const AttnP& p, int wg) {
    RunCtx* cx = p.ctx;
    const int done = step_done(cx, p.step_in_chunk);
    const int T = p.T, b = wg, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int len = p.len[b];
    extern __shared__ float s_dyn[];
    float* s_wlc = s_dyn;                      // [2][16][64]
    float* s_wld = s_wlc + 2048;               // [8][8][64]; the context partial sums reuse it
    float* s_a = s_wld + 4096;                 // [2][FA_SA_LEN]: a_c[i - 15]
    float* s_cf = s_a + 2 * FA_SA_LEN;         // [wave][16][FA_CF_PITCH]
    float* s_q4 = s_cf + GEMM_WAVES * 16 * FA_CF_PITCH;  // (spare)
    float* s_q = s_q4 + 512;                   // [128]
    float* s_v = s_q + 128;                    // [128]
    float* s_e = s_v + 128;                    // [FA_TMAX]
    float* s_al = s_e + FA_TMAX;               // [FA_TMAX]
    __shared__ float s_red[CTX_THREADS / 64];
    FA_MARK(0);
    // This workgroup is on the step's serial path and shares its CU with a piece workgroup that keeps the matrix
    // pipe full: its waves go first wherever both have an instruction ready.
    __builtin_amdgcn_s_setprio(3);

    // ---- loads.  Vector loads return in order, so what the first barrier needs is requested first
    // (query shares, weight fragments, alignment rows) and the encoder rows of the context - 128 KB
    // or more, wanted only at the very end - after it: the barrier then waits for the early group only.
    // Query shares: a row of 128 floats is 32 float4 lanes; the 512 threads read 16 rows at a time.
    constexpr int QROWS = CTX_THREADS / 32, QMAX = (QUERY_DIM / 4) / QROWS;  // 16 rows in parallel, <= 16 passes
    float4 qv[QMAX];
    {
      const float4* qp = reinterpret_cast<const float4*>(p.qpart + (size_t)b * ATT_DIM) + (tid & 31);
      const size_t gs = (size_t)p.Bpad * (ATT_DIM / 4);
#pragma unroll
      for (int u = 0; u < QMAX; ++u) {
        // (unconditional loads from clamped addresses, masked when summed: a load under a lane
        // condition becomes a branch, and the waits at its join serialise the whole load phase)
        const int g = min((tid >> 5) + QROWS * u, p.n_qgroups - 1);
        qv[u] = qp[(size_t)g * gs];
      }
    }
    const float4 w0 = reinterpret_cast<const float4*>(p.wlc_p)[tid];
    const float4 w1 = reinterpret_cast<const float4*>(p.wld_p)[tid];
    const float4 w2 = reinterpret_cast<const float4*>(p.wld_p)[tid + 512];
    constexpr int SA_U = (2 * FA_SA_LEN + CTX_THREADS - 1) / CTX_THREADS;
    float sa[SA_U];
#pragma unroll
    for (int u = 0; u < SA_U; ++u) {
      const int i = tid + u * CTX_THREADS;
      const int c = i >= FA_SA_LEN ? 1 : 0, t = i - c * FA_SA_LEN - ATT_PAD;
      const float* src = c ? p.a_cum : p.a_prev;
      const float val = src[(size_t)b * T + min(max(t, 0), T - 1)];
      sa[u] = (i < 2 * FA_SA_LEN && t >= 0 && t < T) ? val : 0.f;
    }
    const float vw = p.v_w[tid & (ATT_DIM - 1)];
    const float vb = p.v_b[0];
    // processed inputs of the wave's first position tile: requested with the early group where the registers allow
    // (not in the two-workgroups-per-CU variant), so that the energies start without a memory round trip of their own
    constexpr bool P_EARLY = PRE == 0;
    float4 pv0[P_EARLY ? 8 : 1];
    if (P_EARLY) {
      const int t = 16 * wave + (lane & 15);
      const float* Prow = p.P + ((size_t)b * T + (t < T ? t : 0)) * ATT_DIM + 4 * (lane >> 4);
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) pv0[dt] = *reinterpret_cast<const float4*>(Prow + 16 * dt);
    }
    // ---- stage the early group
    {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < QMAX; ++u) {
        const float on = (tid >> 5) + QROWS * u < p.n_qgroups ? 1.f : 0.f;
        acc.x += on * qv[u].x, acc.y += on * qv[u].y, acc.z += on * qv[u].z, acc.w += on * qv[u].w;
      }
      reinterpret_cast<float4*>(s_cf)[tid] = acc;  // [16 rows][32 float4]: partial sums by row group
    }
    reinterpret_cast<float4*>(s_wlc)[tid] = w0;
    reinterpret_cast<float4*>(s_wld)[tid] = w1;
    reinterpret_cast<float4*>(s_wld)[tid + 512] = w2;
#pragma unroll
    for (int u = 0; u < SA_U; ++u)
      if (tid + u * CTX_THREADS < 2 * FA_SA_LEN) s_a[tid + u * CTX_THREADS] = sa[u];
    if (tid < ATT_DIM) s_v[tid] = vw;
    __syncthreads();
    FA_MARK(1);
    // The encoder rows of the first positions wait in registers for the context (one workgroup per CU:
    // the register file is otherwise idle), so it costs no memory round trip at the end.  They are
    // requested in the shadow of the dense MFMAs below, a few per d-tile: 256 KB through one CU's
    // address unit is ~2.5 us of issue when requested in one go, and requested before the first barrier
    // they would have delayed it as well (vector loads return in order: 4.3 us instead of 2.1 us).
    constexpr int LPR = EDIM / 4, RG = CTX_THREADS / LPR;
    const int cl = tid % LPR, rg = tid / LPR;
    const float4* e4 = reinterpret_cast<const float4*>(p.enc + (size_t)b * T * EDIM) + cl;
    constexpr int FA_PRE = PRE ? PRE : (NTT == 1 ? 128 / RG : 16);  // (more tiles per wave leave fewer registers free)
    float4 pre[FA_PRE];
    const int t_last = max(len - 1, 0);
    constexpr int PRE_DT = FA_PRE / 8;  // loads per d-tile
    const bool has_tile = wave < ((T + 15) >> 4);
    if (!has_tile) {
      // a wave without a position tile has nothing to hide them behind
#pragma unroll
      for (int i = 0; i < FA_PRE; ++i) pre[i] = e4[(size_t)min(rg + i * RG, t_last) * (EDIM / 4)];  // (masked at use)
    }
    if (tid < ATT_DIM) {
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < QROWS; ++r) acc += s_cf[r * ATT_DIM + tid];
      s_q[tid] = acc + (p.q_bias ? p.q_bias[tid] : 0.f);
    }
    lds_barrier();  // (the encoder-row loads stay in flight)
    FA_MARK(2);

    // ---- energies, one tile of 16 positions per wave and pass
    const int ntt = (T + 15) >> 4, lq = lane >> 4, lt = lane & 15;
    float* cfw = s_cf + wave * 16 * FA_CF_PITCH;
#pragma unroll
    for (int i = 0; i < NTT; ++i) {
      const int tt = wave + GEMM_WAVES * i;
      if (tt >= ntt) break;  // wave-uniform; no workgroup barrier inside this loop
      const int t = 16 * tt + lt;
      const float* Prow = p.P + ((size_t)b * T + (t < T ? t : 0)) * ATT_DIM + 4 * lq;
      float4 pv[8];
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        if (P_EARLY && i == 0) pv[dt] = pv0[dt];
        else pv[dt] = *reinterpret_cast<const float4*>(Prow + 16 * dt);
      }
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int k = 16 * (kk >> 2) + 4 * lq + (kk & 3);
        const int c = k >= LOC_K ? 1 : 0, j = k - LOC_K * c;  // (k = 62, 63: zero weight, any finite operand)
        const float xv = s_a[c * FA_SA_LEN + 16 * tt + lt + j];
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(s_wlc[kk * 64 + lane], xv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(s_wlc[(16 + kk) * 64 + lane], xv, c1, 0, 0, 0);
      }
      *reinterpret_cast<float4*>(&cfw[lt * FA_CF_PITCH + 4 * lq]) = make_float4(c0[0], c0[1], c0[2], c0[3]);
      *reinterpret_cast<float4*>(&cfw[lt * FA_CF_PITCH + 16 + 4 * lq]) = make_float4(c1[0], c1[1], c1[2], c1[3]);
      __builtin_amdgcn_wave_barrier();
      const float4 bq0 = *reinterpret_cast<const float4*>(&cfw[lt * FA_CF_PITCH + 4 * lq]);
      const float4 bq1 = *reinterpret_cast<const float4*>(&cfw[lt * FA_CF_PITCH + 16 + 4 * lq]);
      float esum = 0.f;
#pragma unroll
      for (int dt = 0; dt < 8; ++dt) {
        if (i == 0) {
#pragma unroll
          for (int u = 0; u < PRE_DT; ++u)
            pre[dt * PRE_DT + u] = e4[(size_t)min(rg + (dt * PRE_DT + u) * RG, t_last) * (EDIM / 4)];
        }
        const float* wa = s_wld + dt * 8 * 64 + lane;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0 * 64], bq0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1 * 64], bq0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2 * 64], bq0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[3 * 64], bq0.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[4 * 64], bq1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[5 * 64], bq1.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[6 * 64], bq1.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[7 * 64], bq1.w, acc, 0, 0, 0);
        const float4 qv = *reinterpret_cast<const float4*>(&s_q[16 * dt + 4 * lq]);
        const float4 vv = *reinterpret_cast<const float4*>(&s_v[16 * dt + 4 * lq]);
        esum += vv.x * fast_tanhf(qv.x + acc[0] + pv[dt].x) + vv.y * fast_tanhf(qv.y + acc[1] + pv[dt].y) +
                vv.z * fast_tanhf(qv.z + acc[2] + pv[dt].z) + vv.w * fast_tanhf(qv.w + acc[3] + pv[dt].w);
      }
      esum += __shfl_xor(esum, 16, 64);
      esum += __shfl_xor(esum, 32, 64);
      if (lane < 16 && t < FA_TMAX) s_e[t] = esum + vb;
      __builtin_amdgcn_wave_barrier();
    }
    lds_barrier();
    FA_MARK(3);

    // ---- normalisation over the utterance's own positions
    const bool softmax = p.norm == TTSX_NORM_SOFTMAX;
    float m = 0.f;
    if (softmax) {
      m = tid < len ? s_e[tid] : -INFINITY;
      m = block_max(m, s_red);
    }
    float v = 0.f;
    if (tid < len) v = softmax ? expf(s_e[tid] - m) : sigmoidf_(s_e[tid]);
    const float sum = block_sum(v, s_red);
    const float a = v / sum;
    if (tid < FA_TMAX) s_al[tid] = a;
    __syncthreads();
    FA_MARK(4);
    if (done) return;
    if (tid < T) {
      const int row = cx->out_base + p.step_in_chunk;
      p.a_prev[(size_t)b * T + tid] = a;
      p.a_cum[(size_t)b * T + tid] = s_a[FA_SA_LEN + ATT_PAD + tid] + a;
      if (!(cx->mask_finished && cx->n_steps[b] != 0)) cx->align[((size_t)b * cx->align_rows + row) * T + tid] = a;
    }
    // ---- context
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < FA_PRE; ++i) {
      const int t = rg + i * RG;
      const float a0 = t < len ? s_al[t] : 0.f;
      acc.x += a0 * pre[i].x, acc.y += a0 * pre[i].y, acc.z += a0 * pre[i].z, acc.w += a0 * pre[i].w;
    }
    if (FA_PRE < 128 * NTT / RG) {
      // the remaining positions, in batches of independent row loads
#pragma unroll 1
      for (int base = FA_PRE; base < 128 * NTT / RG; base += FA_PRE) {
        if (base * RG >= len) break;
#pragma unroll
        for (int i = 0; i < FA_PRE; ++i) pre[i] = e4[(size_t)min(rg + (base + i) * RG, t_last) * (EDIM / 4)];
#pragma unroll
        for (int i = 0; i < FA_PRE; ++i) {
          const int t = rg + (base + i) * RG;
          const float a0 = t < len ? s_al[t] : 0.f;
          acc.x += a0 * pre[i].x, acc.y += a0 * pre[i].y, acc.z += a0 * pre[i].z, acc.w += a0 * pre[i].w;
        }
      }
    }
    float4* s_acc = reinterpret_cast<float4*>(s_wld);  // the dense weights are no longer needed
    s_acc[tid] = acc;
    FA_MARK(5);
    __syncthreads();
    if (tid < LPR) {
      float4 tot = s_acc[tid];
#pragma unroll
      for (int g = 1; g < RG; ++g) {
        const float4 w = s_acc[g * LPR + tid];
        tot.x += w.x, tot.y += w.y, tot.z += w.z, tot.w += w.w;
      }
      if (p.late_flags) {
        // read by the late decoder-LSTM workgroups of this launch: past the L2, then this wave's step tag
        coh_store16(p.ctx_out, (unsigned)(((size_t)tid * p.Bpad + b) * sizeof(float4)), (f32x4){tot.x, tot.y, tot.z, tot.w});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
          __hip_atomic_store(p.late_flags + 2 * b + wave, (unsigned)(cx->base + p.step_in_chunk + 1), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      } else {
        reinterpret_cast<float4*>(p.ctx_out)[(size_t)tid * p.Bpad + b] = tot;
      }
    }
    FA_MARK(6);
  }
};

// processed_inputs P[b][t][d] = sum_c enc[b][t][c] * w_in[d][c]
// (OriginalAttention.preprocess_inputs, once per utterance).
__global__ __launch_bounds__(256) void inputs_proj_kernel(float* __restrict__ P, const float* __restrict__ enc,
                                                          const float* __restrict__ w_in, int T) {
  // block: 16 positions of one utterance; thread (d = tid&127, tg = tid>>7) -> positions tg*8 .. tg*8+7
  const int b = blockIdx.y, t0 = blockIdx.x * 16, tid = threadIdx.x;
  __shared__ __attribute__((aligned(16))) float s_enc[16][ENC_DIM];
  for (int i = tid; i < 16 * ENC_DIM / 4; i += 256) {
    const int tt = i / (ENC_DIM / 4), c4 = i % (ENC_DIM / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t0 + tt < T) v = reinterpret_cast<const float4*>(enc + ((size_t)b * T + t0 + tt) * ENC_DIM)[c4];
    reinterpret_cast<float4*>(&s_enc[tt][0])[c4] = v;
  }
  __syncthreads();
  const int d = tid & 127, tg = tid >> 7;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  const float4* w4 = reinterpret_cast<const float4*>(w_in + (size_t)d * ENC_DIM);
  for (int c4 = 0; c4 < ENC_DIM / 4; ++c4) {
    const float4 w = w4[c4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 x = reinterpret_cast<const float4*>(&s_enc[tg * 8 + i][0])[c4];
      acc[i] += w.x * x.x + w.y * x.y + w.z * x.z + w.w * x.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int t = t0 + tg * 8 + i;
    if (t < T) P[((size_t)b * T + t) * ATT_DIM + d] = acc[i];
  }
}

// location-layer weights in the orders the energy kernel reads them coalesced
__global__ void transpose_loc_kernel(float* __restrict__ conv_t, float* __restrict__ dense_t,
                                     const float* __restrict__ conv_w, const float* __restrict__ dense_w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < LOC_F * 2 * LOC_K) {
    const int f = i / (2 * LOC_K), j = i % (2 * LOC_K);
    conv_t[j * LOC_F + f] = conv_w[i];
  }
  if (i < ATT_DIM * LOC_F) {
    const int d = i / LOC_F, f = i % LOC_F;
    dense_t[((f >> 2) * ATT_DIM + d) * 4 + (f & 3)] = dense_w[i];
  }
}

// location-layer weights as A-operand fragments of v_mfma_f32_16x16x4_f32 (lane l: row l&15, issue
// kk covers k = 16*(kk>>2) + 4*(l>>4) + (kk&3)):
//   wlc_p[(ft*16 + kk)*64 + l] = W_lc[16 ft + (l&15)][k]   (k = channel*31 + tap, zero for k >= 62)
//   wld_p[(dt*8  + kk)*64 + l] = W_ld[16 dt + (l&15)][k]   (k = filter)
// and the query weight transposed, wq_t[u][d] = W_q[d][u].
__global__ void pack_attn_frag_kernel(float* __restrict__ wlc_p, float* __restrict__ wld_p, float* __restrict__ wq_t,
                                      const float* __restrict__ conv_w, const float* __restrict__ dense_w,
                                      const float* __restrict__ query_w, int query_dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * 16 * 64) {
    const int l = i & 63, kk = (i >> 6) & 15, ft = i >> 10;
    const int k = 16 * (kk >> 2) + 4 * (l >> 4) + (kk & 3), f = 16 * ft + (l & 15);
    wlc_p[i] = (conv_w && k < 2 * LOC_K) ? conv_w[f * 2 * LOC_K + k] : 0.f;
  }
  if (i < 8 * 8 * 64) {
    const int l = i & 63, kk = (i >> 6) & 7, dt = i >> 9;
    const int k = 16 * (kk >> 2) + 4 * (l >> 4) + (kk & 3), dd = 16 * dt + (l & 15);
    wld_p[i] = dense_w ? dense_w[dd * LOC_F + k] : 0.f;
  }
  if (i < query_dim * ATT_DIM) {
    const int u = i / ATT_DIM, dd = i % ATT_DIM;
    wq_t[i] = query_w[(size_t)dd * query_dim + u];
  }
}

// ------------------------------------------------------------------ small utility kernels
// (unconditional: steps enqueued after the last stop must keep counting, or step_done() would take them for live)
__global__ void ctx_advance_kernel(RunCtx* c, int n) {
  c->base += n;
  if (c->host_words) __hip_atomic_store(c->host_words + 1, c->base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  c->out_base += n;
}
__global__ void finalize_steps_kernel(int32_t* n_steps, int B, int steps_launched) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B && n_steps[b] == 0) n_steps[b] = steps_launched;
}
__global__ void fill_len_kernel(int32_t* __restrict__ dst, const int32_t* __restrict__ src, int B, int T) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) dst[b] = src ? min(max(src[b], 0), T) : T;
}
// x1 for the first step: memory = 0, so layer 1 gives relu(bias)
__global__ void init_x1_kernel(float* __restrict__ x1, const float* __restrict__ b1f, int Bpad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < PRENET_DIM * Bpad) x1[qidx(i / Bpad, i % Bpad, Bpad)] = fmaxf(b1f[i / Bpad], 0.f);
}
// prenet layer 1 on an explicit memory frame (row-major [B,80]), any batch size: tiny, one thread per output
__global__ void prenet1_explicit_kernel(float* __restrict__ x1, const float* __restrict__ mem,
                                        const float* __restrict__ a1f, const float* __restrict__ b1f, int B,
                                        int Bpad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * PRENET_DIM) return;
  const int b = i / PRENET_DIM, n = i % PRENET_DIM;
  float acc = b1f[n];
  for (int k = 0; k < MEL_DIM; ++k) acc += a1f[n * MEL_DIM + k] * mem[b * MEL_DIM + k];
  x1[qidx(n, b, Bpad)] = fmaxf(acc, 0.f);
}
// initial state of the attention variants: forward-attention weights [1, 1e-7, 1e-7, ...] over the
// utterance's own positions (OriginalAttention.init_forward_attn, Tacotron2/TTS/models.py:227-233),
// transition probability 0.5, window index -1 (:222-225), Graves means 0 (:136)
__global__ void init_variant_state_kernel(float* __restrict__ alpha, float* __restrict__ u, int32_t* __restrict__ win,
                                          float* __restrict__ mu, const int32_t* __restrict__ len, int B, int T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * T) {
    const int b = i / T, t = i % T;
    alpha[i] = t >= len[b] ? 0.f : (t == 0 ? 1.f : 1e-7f);
  }
  if (i < B) u[i] = 0.5f, win[i] = -1;
  if (i < B * GRAVES_MAX_K) mu[i] = 0.f;
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
  // source-layout copies needed to compose rows for a new r
  float *proj_w_src, *proj_b_src, *stop_w_src, *stop_b_src, *a1f, *b1f;
  // packed weights
  float *w_lstm1, *b_lstm1, *w_lstm2, *b_lstm2, *w_query, *b_query, *w_pre2, *b_pre2;
  float *w_proj, *b_proj;  // mel rows + composed rows, valid for proj_r
  int proj_r;
  float *inputs_w, *v_w, *v_b, *loc_conv_w, *loc_dense_w;
  float *wlc_p, *wld_p, *wq_t;  // fused attention kernel: MFMA fragments of the location layer, W_q transposed
  float* qpart;                 // workspace: shares of the query projection, [groups][Bpad][128]
  int qpart_groups;
  bool fused;                   // this run uses the fused attention kernel (T <= 256, no variant)
  // optional attention variants
  bool variant;      // any of windowing / forward attention: the one-workgroup-per-utterance tail
  int windowing, forward_attn, forward_attn_mask, trans_agent, graves_K;
  float *ta_w, *ta_b;
  float *w_na1, *b_na1, *w_na2, *b_na2;  // packed GravesAttention.N_a layers
  float *alpha, *u_fa, *mu, *n1, *gbk;   // workspace of the variants
  int32_t* win_idx;
  // one-shot settings of the next begin(): keep the recurrent state, first fed-back frame
  bool keep_next, has_state;
  float* mem0;       // [cap_B][80] copy of the caller's frame, valid when mem0_set
  bool mem0_set;
  int mem0_cap;
  // per-run workspace
  int B, T, Bpad, cap_B, cap_T, ws_gen;
  float *x1, *x2, *ctx, *h_a[2], *c_a, *h_d[2], *c_d, *q, *o, *a_prev, *a_cum, *energies, *P;
  float* enc_ws;     // [B][T][512] copy of the encoder outputs (kernel arguments must not move between runs)
  int32_t* len_ws;   // [B] encoder lengths (T where the caller gave none)
  int parity0;       // decoder steps completed before the launches being enqueued, & 1
  bool split;        // LSTMs cut into early/late columns with the early part riding along other launches
  bool staged;       // (MFMA engine, padded batch 64) the early-column pieces stage their activations in LDS
  int plan_staged;   // ttsx_t2_decoder_set_pieces: -1 by shape (default), 0 register tiles, 1 staged where possible
  float *g_a, *g_d;  // raw gate sums of the partial LSTM launches, quad layout [4096/4][Bpad] float4
  RunCtx* rc;        // device
  RunCtx* rc_host;   // pinned staging copy
  int* host_flag;    // pinned
  int steps_done;    // host mirror of rc.base
  int last_launched; // decoder steps the last run() enqueued (>= the steps any utterance needed)
  bool begun;
  // cached chunk graphs: slot 0 for the long chunk (fixed-step runs, batches), slot 1 for the short chunk of a
  // free-running decode at batch 1-2
  struct ChunkGraph {
    hipGraphExec_t exec;
    int B, T, r, gen, chunk, split, fused;
  } graphs[3];  // chunks of CHUNK, of the short (polling) size, of CHUNK_LONG steps
  hipStream_t cap_stream;  // capture-only stream: the caller's stream may be the legacy default stream
  int* host_words_dev;     // device-side address of host_flag (mapped pinned memory)
  int poll_chunk, poll_ahead;  // free run at batch 1-2: steps per graph and steps kept in flight (0: defaults)
  int plan_fused, plan_split, plan_share_q;  // ttsx_t2_decoder_set_plan: -1 by shape (default), 0 off, 1 on
  bool share_q;  // three-launch attention without the query launch: the energies op adds up the query shares
  unsigned* x1_flags;  // [Bpad / 16][16] step tags of the prenet layer-1 row blocks (Prenet2AfterProjOp)
  unsigned* late_flags;  // [2 Bpad + STAGED_PARTS * 256] step tags for the late decoder-LSTM columns inside the attention launch
  int plan_fuse_p2;    // ttsx_t2_decoder_set_fusion: prenet layer 2 of the next step inside the projection launch (1), and
                       // the late decoder-LSTM columns inside the attention launch as well (2); -1: where measured faster
  bool fuse_gave_up;   // a prenet workgroup of that launch once waited too long: the default plan no longer fuses
  // one utterance, short text, plain attention: the whole run as one persistent launch with the weights on chip
  t2_team* team;
  int plan_team;      // ttsx_t2_decoder_set_team: -1 by shape (default), 0 never, 1 wherever the shapes allow
  TeamCoolDown team_cool;  // a team run timed out waiting for its workgroups: the default plan uses the launch chain for a while
};

namespace {

// Steps per cached graph.  A free run notices its last stop through a word the device writes into pinned host
// memory (no stream synchronisation, no copy).  At batch 1-2 every step enqueued past the stop frame is pure
// latency, so the host enqueues short graphs, one at a time (the next one when the device reports the previous
// one complete): fewer than CHUNK_SHORT steps go past the stop.  Measured on a 308-step utterance (decoder time /
// steps enqueued): 4 steps x 2 in flight 10.55 ms / 312, 8 x 1 10.42 / 312, 16 x 1 10.28 / 320, 32 x 1 10.14 / 320,
// 32 x 2 10.96 / 352 - a short graph costs ~0.5 us more per step (31.7 us per step at 32, 33.4 at 8), about what
// the smaller overshoot gives back; larger batches use the long chunk.  (Both even: the h ping-pong parity of a
// chunk's steps is fixed at capture.)
constexpr int CHUNK = 32, CHUNK_SHORT = 8;
// Runs with a fixed number of steps poll nothing, so their graphs can be long: the first step of a graph has no
// projection launch of a step before it to carry its prenet layer 2 and its early attention-LSTM columns (two
// launches of its own, ~13 us at padded batch 64) - once per 160 steps instead of once per 32.
constexpr int CHUNK_LONG = 160;
int gemm_shape(int Bpad, int nrb);
int query_groups(int B, int Bpad);
hipError_t set_fused_lds();

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
  if (!*wp) {
    int rc = dev_alloc(wp, n);
    if (rc) return rc;
    rc = dev_alloc(bp, (size_t)nrb * 16);
    if (rc) return rc;
  }
  pack_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(*wp, w1, k1, w2, k2, rows, KG, lstm_hidden);
  TTSX_LAUNCH_CHECK();
  pack_bias_kernel<<<ceil_div(nrb * 16, 256), 256, 0, st>>>(*bp, b1, b2, rows, lstm_hidden);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

void drop_graph(ttsx_t2_decoder* d) {
  for (auto& g : d->graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    g.exec = nullptr;
  }
}

// Buffers sized by the (padded) batch alone - they hold the recurrent state, which
// ttsx_t2_decoder_set_initial can carry into a run over a longer text - and buffers sized by batch x T.
void free_len_workspace(ttsx_t2_decoder* d) {
  float** bufs[] = {&d->a_prev, &d->a_cum, &d->energies, &d->P, &d->enc_ws, &d->alpha};
  for (float** b : bufs) {
    if (*b) (void)hipFree(*b);
    *b = nullptr;
  }
  d->cap_T = 0;
  drop_graph(d);
}

void free_workspace(ttsx_t2_decoder* d) {
  free_len_workspace(d);
  float** bufs[] = {&d->x1, &d->x2, &d->ctx, &d->h_a[0], &d->h_a[1], &d->c_a, &d->h_d[0], &d->h_d[1], &d->c_d,
                    &d->q,  &d->o,  &d->g_a, &d->g_d, &d->u_fa, &d->mu, &d->n1, &d->gbk, &d->qpart};
  for (float** b : bufs) {
    if (*b) (void)hipFree(*b);
    *b = nullptr;
  }
  if (d->len_ws) (void)hipFree(d->len_ws);
  d->len_ws = nullptr;
  if (d->win_idx) (void)hipFree(d->win_idx);
  d->win_idx = nullptr;
  if (d->x1_flags) (void)hipFree(d->x1_flags);
  d->x1_flags = nullptr;
  if (d->late_flags) (void)hipFree(d->late_flags);
  d->late_flags = nullptr;
  d->cap_B = 0;
}

// Rows of the projection launch for reduction factor r: the 80 r mel rows a step keeps, 16 rows for the composed
// stop logit (one used) and the 256 composed prenet rows.  The remaining 80 (r_init - r) rows of linear_projection
// feed nothing but the stopnet, whose logit is already one composed row: they are never multiplied.
int proj_rows(int r) { return MEL_DIM * r + 16 + PRENET_DIM; }

int ensure_workspace(ttsx_t2_decoder* d, int B, int T) {
  // column padding: 16 / 32 / multiples of 64 (a count of 3 column tiles would leave the LSTM launches
  // with one and a half waves of workgroups: slower than padding to 64)
  const int Bpad = B <= GEMV_MAX_B ? B : (B <= 32 ? round_up(B, 16) : round_up(B, 64));
  const size_t cb = Bpad, ct = T;
  int rc = 0;
  if (d->cap_B != Bpad) {
    free_workspace(d);
    rc |= dev_alloc(&d->x1, PRENET_DIM * cb);
    rc |= dev_alloc(&d->x2, PRENET_DIM * cb);
    rc |= dev_alloc(&d->ctx, ENC_DIM * cb);
    for (int i = 0; i < 2; ++i) {
      rc |= dev_alloc(&d->h_a[i], QUERY_DIM * cb);
      rc |= dev_alloc(&d->h_d[i], DEC_DIM * cb);
    }
    rc |= dev_alloc(&d->c_a, QUERY_DIM * cb);
    rc |= dev_alloc(&d->c_d, DEC_DIM * cb);
    rc |= dev_alloc(&d->g_a, 2 * 4 * QUERY_DIM * cb);  // two K parts
    rc |= dev_alloc(&d->g_d, STAGED_PARTS * 4 * DEC_DIM * cb);  // two K parts, or those of the staged pieces
    rc |= dev_alloc(&d->q, ATT_DIM * cb);
    rc |= dev_alloc(&d->o, (size_t)round_up(proj_rows(d->r_init), 16) * cb);
    rc |= dev_alloc(&d->len_ws, cb);
    rc |= dev_alloc(&d->u_fa, cb);
    rc |= dev_alloc(&d->win_idx, cb);
    rc |= dev_alloc(&d->x1_flags, (size_t)ceil_div((int)cb, 16) * (PRENET_DIM / 16));
    rc |= dev_alloc(&d->late_flags, 2 * cb + (size_t)STAGED_PARTS * (DEC_DIM / 4));
    rc |= dev_alloc(&d->mu, cb * GRAVES_MAX_K);
    rc |= dev_alloc(&d->n1, QUERY_DIM * cb);
    rc |= dev_alloc(&d->gbk, (size_t)round_up(3 * GRAVES_MAX_K, 16) * cb);
    d->qpart_groups = query_groups(B, Bpad);
    rc |= dev_alloc(&d->qpart, (size_t)d->qpart_groups * cb * ATT_DIM);
    if (rc) return TTSX_ERR_HIP;
    d->cap_B = Bpad;
    d->ws_gen += 1;
  }
  if (d->cap_T < T) {
    free_len_workspace(d);
    rc |= dev_alloc(&d->a_prev, cb * ct);
    rc |= dev_alloc(&d->a_cum, cb * ct);
    rc |= dev_alloc(&d->energies, cb * ct);
    rc |= dev_alloc(&d->P, cb * ATT_DIM * ct);
    rc |= dev_alloc(&d->enc_ws, cb * ENC_DIM * ct);
    rc |= dev_alloc(&d->alpha, cb * ct);
    if (rc) return TTSX_ERR_HIP;
    d->cap_T = T;
    d->ws_gen += 1;
  }
  d->Bpad = Bpad;
  return TTSX_OK;
}

// (Re)build the projection rows for reduction factor r.
int ensure_proj(ttsx_t2_decoder* d, int r, hipStream_t st) {
  if (d->proj_r == r) return TTSX_OK;
  const int K = DEC_DIM + ENC_DIM, ext_rows = 16 + PRENET_DIM;
  float *ext_w = nullptr, *ext_b = nullptr, *all_w = nullptr, *all_b = nullptr;
  if (dev_alloc(&ext_w, (size_t)ext_rows * K) || dev_alloc(&ext_b, ext_rows)) return TTSX_ERR_HIP;
  compose_proj_ext_kernel<<<ext_rows, 256, 0, st>>>(ext_w, ext_b, d->proj_w_src, d->proj_b_src, d->stop_w_src,
                                                   d->stop_b_src, d->a1f, d->b1f, d->odim, r);
  TTSX_LAUNCH_CHECK();
  const int rows = proj_rows(r), kept = MEL_DIM * r;
  if (dev_alloc(&all_w, (size_t)rows * K) || dev_alloc(&all_b, rows)) return TTSX_ERR_HIP;
  TTSX_HIP(hipMemcpyAsync(all_w, d->proj_w_src, sizeof(float) * (size_t)kept * K, hipMemcpyDeviceToDevice, st));
  TTSX_HIP(hipMemcpyAsync(all_w + (size_t)kept * K, ext_w, sizeof(float) * (size_t)ext_rows * K,
                          hipMemcpyDeviceToDevice, st));
  TTSX_HIP(hipMemcpyAsync(all_b, d->proj_b_src, sizeof(float) * kept, hipMemcpyDeviceToDevice, st));
  TTSX_HIP(hipMemcpyAsync(all_b + kept, ext_b, sizeof(float) * ext_rows, hipMemcpyDeviceToDevice, st));
  int rc = pack_layer(&d->w_proj, &d->b_proj, all_w, K, nullptr, 0, all_b, nullptr, rows, 0, st);
  if (!rc && d->team) rc = t2_team_set_proj(d->team, all_w, all_b, r, st);
  TTSX_HIP(hipStreamSynchronize(st));
  (void)hipFree(ext_w), (void)hipFree(ext_b), (void)hipFree(all_w), (void)hipFree(all_b);
  if (rc) return rc;
  d->proj_r = r;
  drop_graph(d);
  return TTSX_OK;
}

// A piece of partial LSTM work: row blocks [rb0, rb0+nrb) of the launch described by `g`.
struct Piece {
  GemmP g;
  bool on;
  int nk = 1;  // MFMA engine: K parts this piece covers (g.kpart0 .. g.kpart0 + nk - 1 of g.kparts)
};

// The two partial launches of a step, each cut into `parts` pieces by row blocks.
struct Partial {
  GemmP g;
  int total_rb;
  Piece piece(int k, int parts = 3) const {
    Piece pc{g, true};
    const int lo = total_rb * k / parts, hi = total_rb * (k + 1) / parts;
    pc.g.rb0 = lo, pc.g.nrb = hi - lo;
    return pc;
  }
};

// Launch op A (params pa, na workgroups, extra dynamic LDS `smem`) together with a partial piece.
// PK = k-groups per wave of the partial piece (12: attention LSTM, 16: decoder LSTM).  On the MFMA engine the
// pieces are cut along K instead of into narrow tiles (32 x 32 tiles keep the activation re-reads down, and half
// the K range keeps a piece workgroup short enough to pack into the CUs that op A leaves idle).
template <class A>
struct piece_in_dyn_lds {
  static constexpr bool value = false;
};
template <int NTT, int EDIM, int PRE>
struct piece_in_dyn_lds<FusedAttnOp<NTT, EDIM, PRE>> {
  static constexpr bool value = true;
};

// Ops that share a launch with staged pieces (the fused attention; none)
template <class A>
struct staged_partner {
  static constexpr bool value = false;
};
template <>
struct staged_partner<NoOp> {
  static constexpr bool value = true;
};
template <int NTT, int EDIM, int PRE>
struct staged_partner<FusedAttnOp<NTT, EDIM, PRE>> {
  static constexpr bool value = true;
};
// What a launch with staged pieces asks for per workgroup: more than half the LDS of a CU
constexpr size_t STAGED_LDS_MIN = 84 * 1024;

template <class A, int PK>
int launch_with_piece(const ttsx_t2_decoder* d, const typename A::P& pa, int na, size_t smem, const Piece* pc,
                      hipStream_t st) {
  constexpr bool DYN = piece_in_dyn_lds<A>::value;
  static_assert(sizeof(f32x4) * GEMM_WAVES * 4 * 64 <= sizeof(float) * FA_LDS_FLOATS,
                "the combine buffer of a 2 x 2 tile must fit the fused attention kernel's dynamic LDS");
  if (!pc || !pc->on) {
    dual_kernel<A, NoOp><<<na, GEMM_THREADS, smem, st>>>(pa, 0, na);
  } else if (d->B <= GEMV_MAX_B) {
    if (pc->g.kparts > 1) {
      // both K parts of the piece's row blocks: twice the workgroups, half the stream each
      const int nb = pc->g.nrb * 2;
      if (d->B == 1) dual_kernel<A, GemvOp<1, PK / 2, MODE_PARTIAL, 2>><<<na + nb, GEMM_THREADS, smem, st>>>(pa, pc->g, na);
      else dual_kernel<A, GemvOp<2, PK / 2, MODE_PARTIAL, 2>><<<na + nb, GEMM_THREADS, smem, st>>>(pa, pc->g, na);
    } else {
      const int nb = pc->g.nrb;
      if (d->B == 1) dual_kernel<A, GemvOp<1, PK, MODE_PARTIAL>><<<na + nb, GEMM_THREADS, smem, st>>>(pa, pc->g, na);
      else dual_kernel<A, GemvOp<2, PK, MODE_PARTIAL>><<<na + nb, GEMM_THREADS, smem, st>>>(pa, pc->g, na);
    }
  } else if (d->staged && pc->g.kparts == STAGED_PARTS) {
    // 128 rows x 64 columns x one of STAGED_PARTS parts of K per workgroup, the activation slice through LDS, one
    // workgroup per CU (the LDS request is what keeps a second one off it)
    if constexpr (!staged_partner<A>::value || PK != (QUERY_DIM + DEC_DIM) / 16 / GEMM_WAVES) {
      ttsx_set_error("launch_with_piece: no staged piece beside this op");
      return TTSX_ERR_STATE;
    } else {
      typedef StagedPieceOp<PK * GEMM_WAVES, STAGED_PARTS> Op;
      const int nb = pc->g.nrb / GEMM_WAVES * pc->nk;
      const size_t lds = std::max(std::max(smem, (size_t)Op::LDS_BYTES), STAGED_LDS_MIN);
      static bool raised = false;   // (per instantiation)
      if (!raised) {
        TTSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<A, Op>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
      }
      dual_kernel<A, Op><<<na + nb, GEMM_THREADS, lds, st>>>(pa, pc->g, na);
    }
  } else {
    // K in two parts of PK/2 k-groups per wave, 32 x 32 tiles (16 x 16 when the batch is one column tile)
    const int nt = d->Bpad / 16;
    if (nt % 2 == 0 && pc->g.nrb % 2 == 0) {
      const int nb = pc->g.nrb / 2 * (nt / 2) * pc->nk;
      dual_kernel<A, GemmOp<2, 2, PK / 2, MODE_PARTIAL, DYN>><<<na + nb, GEMM_THREADS, smem, st>>>(pa, pc->g, na);
    } else {
      const int nb = pc->g.nrb * nt * pc->nk;
      dual_kernel<A, GemmOp<1, 1, PK / 2, MODE_PARTIAL, DYN>><<<na + nb, GEMM_THREADS, smem, st>>>(pa, pc->g, na);
    }
  }
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

// A GEMM/GEMV launch on the serial path (`g` covers all `nrb` row blocks) with an optional piece.
// KGW = k-groups per wave of the main op; `big`: a weight-streaming launch (LSTM) as opposed to one
// of the small ones, which need many workgroups more than they need reuse.
// Tile of a weight-streaming launch: R row blocks x NT column tiles per workgroup, code 10 R + NT.
// Bigger tiles re-read fewer activations, but the launch must still cover the chip: take the
// biggest tile that leaves at least one workgroup per CU (measured at batch 16..128).
int gemm_shape(int Bpad, int nrb) {
  const int nt_total = Bpad / 16;
  int shape = 11;
  const int cus = 256;
  if (nrb % 2 == 0 && nrb / 2 * nt_total >= cus) shape = 21;
  if (nrb % 2 == 0 && nt_total % 2 == 0 && nrb / 2 * (nt_total / 2) >= cus) shape = 22;
  if (nrb % 4 == 0 && nt_total % 2 == 0 && nrb / 4 * (nt_total / 2) >= cus) shape = 42;
  // 64 x 64 (padded batch 256 and above): a third less L2 -> CU traffic than 64 x 32, which is what bounds the LSTM
  // launches there; its combine buffer (128 KB) lives in dynamic LDS, one workgroup per CU
  if (nrb % 4 == 0 && nt_total % 4 == 0 && nrb / 4 * (nt_total / 4) >= cus) shape = 44;
  return shape;
}
// unit groups of the attention LSTM launch (= workgroups along the row axis): one share of the query each
int query_groups(int B, int Bpad) {
  if (B <= GEMV_MAX_B) return QUERY_DIM / 4;
  return QUERY_DIM / 4 / (gemm_shape(Bpad, QUERY_DIM / 4) / 10);
}

constexpr size_t GEMM44_LDS_BYTES = sizeof(f32x4) * GEMM_WAVES * 16 * 64;  // combine buffer of a 4 x 4 tile

template <int MODE, int KGW, int PK>
int launch_gemm(const ttsx_t2_decoder* d, GemmP g, int nrb, bool big, const Piece* pc, hipStream_t st) {
  g.rb0 = 0, g.nrb = nrb;
  if (d->B <= GEMV_MAX_B) {
    if (d->B == 1) return launch_with_piece<GemvOp<1, KGW, MODE>, PK>(d, g, nrb, 0, pc, st);
    return launch_with_piece<GemvOp<2, KGW, MODE>, PK>(d, g, nrb, 0, pc, st);
  }
  const int nt_total = d->Bpad / 16;
  if (!big) return launch_with_piece<GemmOp<1, 1, KGW, MODE>, PK>(d, g, nrb * nt_total, 0, pc, st);
  const int shape = gemm_shape(d->Bpad, nrb);
  switch (shape) {
    case 21: return launch_with_piece<GemmOp<2, 1, KGW, MODE>, PK>(d, g, nrb / 2 * nt_total, 0, pc, st);
    case 22: return launch_with_piece<GemmOp<2, 2, KGW, MODE>, PK>(d, g, nrb / 2 * (nt_total / 2), 0, pc, st);
    case 42: return launch_with_piece<GemmOp<4, 2, KGW, MODE>, PK>(d, g, nrb / 4 * (nt_total / 2), 0, pc, st);
    case 24: return launch_with_piece<GemmOp<2, 4, KGW, MODE>, PK>(d, g, nrb / 2 * (nt_total / 4), 0, pc, st);
    case 44: {
      typedef GemmOp<4, 4, KGW, MODE, true> Op;
      TTSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<Op, NoOp>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM44_LDS_BYTES));
      dual_kernel<Op, NoOp><<<nrb / 4 * (nt_total / 4), GEMM_THREADS, GEMM44_LDS_BYTES, st>>>(g, 0, nrb / 4 * (nt_total / 4));
      TTSX_LAUNCH_CHECK();
      return TTSX_OK;
    }
    default: return launch_with_piece<GemmOp<1, 1, KGW, MODE>, PK>(d, g, nrb * nt_total, 0, pc, st);
  }
}

constexpr int KG1 = (PRENET_DIM + ENC_DIM + QUERY_DIM) / 16, KG2 = (ENC_DIM + QUERY_DIM + DEC_DIM) / 16;
constexpr int PK1 = (ENC_DIM + QUERY_DIM) / 16 / GEMM_WAVES, PK2 = (QUERY_DIM + DEC_DIM) / 16 / GEMM_WAVES;

// Dynamic LDS (the alignment row, T floats) above the default limit: raise it on every launch
// variant that carries the context op.
template <class B>
hipError_t set_context_lds_for(size_t bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<ContextOp<128>, B>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<ContextOp<32>, B>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
hipError_t set_context_lds(size_t bytes, bool variant, bool graves) {
  hipError_t e;
  // the one-workgroup-per-utterance variants keep up to three rows of T floats (only asked for when they run:
  // at the longest texts of the plain attention that request would exceed the LDS of a CU)
  if (variant && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<AttnPostOp, NoOp>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * bytes))) != hipSuccess)
    return e;
  if (graves && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<GravesOp, NoOp>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * bytes + 64))) != hipSuccess)
    return e;
  if ((e = set_context_lds_for<NoOp>(bytes)) != hipSuccess) return e;
  if ((e = set_context_lds_for<GemvOp<1, PK2 / 2, MODE_PARTIAL, 2>>(bytes)) != hipSuccess) return e;
  if ((e = set_context_lds_for<GemvOp<2, PK2 / 2, MODE_PARTIAL, 2>>(bytes)) != hipSuccess) return e;
  if ((e = set_context_lds_for<GemvOp<1, PK2, MODE_PARTIAL>>(bytes)) != hipSuccess) return e;
  return set_context_lds_for<GemvOp<2, PK2, MODE_PARTIAL>>(bytes);
}

// The fused attention kernel keeps ~50 KB of dynamic LDS beside the static arrays of its launch partner.
template <class A>
hipError_t set_fused_lds_for() {
  const int bytes = FA_LDS_FLOATS * (int)sizeof(float);
  hipError_t e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<A, NoOp>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<A, GemvOp<1, PK2 / 2, MODE_PARTIAL, 2>>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<A, GemvOp<1, PK2, MODE_PARTIAL>>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<A, GemvOp<2, PK2, MODE_PARTIAL>>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<A, GemvOp<2, PK2 / 2, MODE_PARTIAL, 2>>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<A, GemmOp<1, 1, PK2 / 2, MODE_PARTIAL, true>>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes)) != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<A, GemmOp<2, 2, PK2 / 2, MODE_PARTIAL, true>>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
hipError_t set_fused_lds() {
  hipError_t e = set_fused_lds_for<FusedAttnOp<1, ENC_DIM>>();
  if (e != hipSuccess) return e;
  if ((e = set_fused_lds_for<FusedAttnOp<1, ENC_DIM, 16>>()) != hipSuccess) return e;
  if ((e = set_fused_lds_for<FusedAttnOp<2, ENC_DIM>>()) != hipSuccess) return e;
  return set_fused_lds_for<FusedAttnOp<4, ENC_DIM>>();
}

// (K parts of the partial launches: two on the MFMA engine, and on the GEMV engine next to the fused attention -
// measured there 32.3 -> 31.8 us/step at T=100, but 33.5 -> 35.0 at T=250, where the energies / context launches
// bring many workgroups of their own and the pieces stay whole.)
// Early columns of the attention LSTM, [ctx(t-1); h_a(t-1)] -> g_a, for the step with chunk index i.
Partial partial_attn(const ttsx_t2_decoder* d, int i) {
  GemmP g{};
  g.Bpad = d->Bpad, g.ctx = d->rc, g.step_in_chunk = i, g.parity = (d->parity0 + i) & 1;
  g.Wp = d->w_lstm1, g.nseg = 2, g.KG = (ENC_DIM + QUERY_DIM) / 16, g.kg0 = PRENET_DIM / 16, g.kg_stride = KG1;
  g.gates = d->g_a;
  g.kparts = (d->B > GEMV_MAX_B || d->fused) ? 2 : 1, g.gates_part_stride = QUERY_DIM * d->Bpad;
  g.seg[0] = Seg{{d->ctx, nullptr}, 0, ENC_DIM / 16};
  g.seg[1] = Seg{{d->h_a[0], d->h_a[1]}, 1, QUERY_DIM / 16};
  return Partial{g, QUERY_DIM / 4};
}
// Early columns of the decoder LSTM, [h_a(t); h_d(t-1)] -> g_d.
Partial partial_dec(const ttsx_t2_decoder* d, int i) {
  GemmP g{};
  g.Bpad = d->Bpad, g.ctx = d->rc, g.step_in_chunk = i, g.parity = (d->parity0 + i) & 1;
  g.Wp = d->w_lstm2, g.nseg = 2, g.KG = (QUERY_DIM + DEC_DIM) / 16, g.kg0 = ENC_DIM / 16, g.kg_stride = KG2;
  g.gates = d->g_d;
  g.kparts = d->staged ? STAGED_PARTS : (d->B > GEMV_MAX_B || d->fused) ? 2 : 1, g.gates_part_stride = DEC_DIM * d->Bpad;
  g.seg[0] = Seg{{d->h_a[0], d->h_a[1]}, 2, QUERY_DIM / 16};
  g.seg[1] = Seg{{d->h_d[0], d->h_d[1]}, 1, DEC_DIM / 16};
  return Partial{g, DEC_DIM / 4};
}

// Enqueue one decoder step; `i` is the step's index inside the current chunk.
//   first: (split mode) the early attention-LSTM columns of this step have not been computed by
//          the previous step's launches (first step of a chunk, or a single teacher-forced step);
//   more:  another step follows in this chunk, so its early attention-LSTM columns ride along with
//          this step's decoder-LSTM and projection launches (and with its own prenet launch).
int enqueue_step(ttsx_t2_decoder* d, int i, bool first, bool more, hipStream_t st) {
  const int Bpad = d->Bpad, B = d->B, T = d->T;
  const bool split = d->split;
  GemmP g{};
  g.Bpad = Bpad, g.ctx = d->rc, g.step_in_chunk = i, g.parity = (d->parity0 + i) & 1;
  int rc;
  const Partial pa = partial_attn(d, i), pd = partial_dec(d, i);
  const Partial pa_next = partial_attn(d, i + 1);  // reads the state this step leaves behind
  Piece pc{};
  if (split && first) {
    // nothing to hide behind yet: the whole partial as a launch of its own
    pc = Piece{pa.g, true};
    pc.g.rb0 = 0, pc.g.nrb = pa.total_rb, pc.nk = pa.g.kparts;
    if ((rc = launch_with_piece<NoOp, PK1>(d, 0, 0, 0, &pc, st))) return rc;
  }
  const bool mfma = B > GEMV_MAX_B;
  // all rows, one K part (MFMA engine)
  auto k_piece = [](const Partial& pt, int kpart) {
    Piece q{pt.g, true};
    const int per = pt.g.kparts / 2;  // (one half of K: one part, or four staged parts)
    q.g.rb0 = 0, q.g.nrb = pt.total_rb, q.g.kpart0 = kpart * per, q.nk = per;
    return q;
  };
  // prenet layer 2: x2 = relu(A2f x1 + b2f)   (+ last third of this step's attention partial)
  g.Wp = d->w_pre2, g.bias = d->b_pre2, g.nseg = 1, g.KG = PRENET_DIM / 16, g.kg0 = 0, g.kg_stride = g.KG;
  g.seg[0] = Seg{{d->x1, nullptr}, 0, PRENET_DIM / 16};
  g.out[0] = d->x2, g.relu = 1;
  const GemmP g_pre2 = g;
  // MFMA engine: inside a chunk, prenet layer 2 of a step (and both K halves of its early attention-LSTM columns)
  // ran in the projection launch of the step before it
  // (measured per step, own launch / fused, T=128 unless noted: B=32 38.7 / 39.9 us, B=48 48.1 / 46.9, B=64 48.2 /
  // 46.6, B=64 T=256 52.2 / 50.9, B=64 T=64 (no split) 49.1 / 51.4, B=128 (no split) 73.3 / 76.4: it pays where the
  // early/late split is on at padded batch 64, and costs where the fused launch has no piece work to hide behind)
  const bool fuse_p2 = mfma && (d->plan_fuse_p2 >= 1 || (d->plan_fuse_p2 == -1 && !d->fuse_gave_up && split && Bpad == 64));
  // the late decoder-LSTM columns inside the attention launch: where that launch gives every workgroup a CU of
  // its own (staged pieces) and the plain fused attention runs
  const bool late_in_attn = mfma && split && d->fused && d->staged && pd.g.kparts == STAGED_PARTS && T <= 128 && Bpad == 64 &&
                            d->graves_K == 0 && (d->plan_fuse_p2 == 2 || (d->plan_fuse_p2 == -1 && !d->fuse_gave_up));
  // (pieces of the attention partial: thirds of the rows behind decoder LSTM / projection / prenet at batch 1-2;
  // on the MFMA engine the two halves of K behind projection / prenet)
  pc = mfma ? k_piece(pa, 1) : pa.piece(2);
  pc.on = split && !first;
  if (!(fuse_p2 && !first) && (rc = launch_gemm<MODE_LINEAR, 2, PK1>(d, g, PRENET_DIM / 16, false, &pc, st))) return rc;
  // attention LSTM: [x2; ctx; h_a] (split: late columns [x2] + the partial sums) + cell update
  g.Wp = d->w_lstm1, g.bias = d->b_lstm1, g.kg0 = 0, g.kg_stride = KG1;
  g.seg[0] = Seg{{d->x2, nullptr}, 0, PRENET_DIM / 16};
  g.seg[1] = Seg{{d->ctx, nullptr}, 0, ENC_DIM / 16};
  g.seg[2] = Seg{{d->h_a[0], d->h_a[1]}, 1, QUERY_DIM / 16};
  g.out[0] = d->h_a[0], g.out[1] = d->h_a[1], g.c = d->c_a, g.relu = 0;
  if (d->fused || d->share_q) g.wq_t = d->wq_t, g.qpart = d->qpart;  // the launch also leaves the shares of the query
  if (split) {
    g.nseg = 1, g.KG = PRENET_DIM / 16, g.gates = d->g_a;
    g.kparts = pa.g.kparts, g.gates_part_stride = pa.g.gates_part_stride;
    rc = launch_gemm<MODE_LSTM, 2, PK1>(d, g, QUERY_DIM / 4, true, nullptr, st);
    g.kparts = 0;
  } else {
    g.nseg = 3, g.KG = KG1, g.gates = nullptr;
    rc = launch_gemm<MODE_LSTM, KG1 / GEMM_WAVES, PK1>(d, g, QUERY_DIM / 4, true, nullptr, st);
  }
  if (rc) return rc;
  g.wq_t = nullptr, g.qpart = nullptr;
  AttnP ap{};
  ap.q = d->q, ap.P = d->P;
  ap.loc_conv_t = d->loc_conv_w, ap.loc_dense_t = d->loc_dense_w, ap.v_w = d->v_w, ap.v_b = d->v_b;
  ap.a_prev = d->a_prev, ap.a_cum = d->a_cum, ap.energies = d->energies, ap.ctx_out = d->ctx;
  ap.enc = d->enc_ws, ap.len = d->len_ws, ap.T = T;
  ap.ctx = d->rc, ap.step_in_chunk = i, ap.Bpad = Bpad, ap.norm = d->norm;
  ap.windowing = d->windowing, ap.forward_attn = d->forward_attn, ap.forward_attn_mask = d->forward_attn_mask;
  ap.trans_agent = d->trans_agent, ap.win_idx = d->win_idx, ap.alpha = d->alpha, ap.u = d->u_fa;
  ap.ta_w = d->ta_w, ap.ta_b = d->ta_b, ap.h_new = g.parity ? d->h_a[0] : d->h_a[1];
  ap.gbk = d->gbk, ap.mu = d->mu, ap.graves_K = d->graves_K;
  if (d->graves_K > 0) {
    // GravesAttention: gbk = N_a.2(relu(N_a.0(h_a))) as two row-block launches, then the mixture
    g.Wp = d->w_na1, g.bias = d->b_na1, g.nseg = 1, g.KG = QUERY_DIM / 16, g.kg0 = 0, g.kg_stride = g.KG;
    g.gates = nullptr;
    g.seg[0] = Seg{{d->h_a[0], d->h_a[1]}, 2, QUERY_DIM / 16};
    g.out[0] = d->n1, g.out[1] = nullptr, g.c = nullptr, g.relu = 1;
    if ((rc = launch_gemm<MODE_LINEAR, 8, PK2>(d, g, QUERY_DIM / 16, false, nullptr, st))) return rc;
    g.Wp = d->w_na2, g.bias = d->b_na2;
    g.seg[0] = Seg{{d->n1, nullptr}, 0, QUERY_DIM / 16};
    g.out[0] = d->gbk, g.relu = 0;
    if ((rc = launch_gemm<MODE_LINEAR, 8, PK2>(d, g, ceil_div(3 * d->graves_K, 16), false, nullptr, st))) return rc;
    if ((rc = launch_with_piece<GravesOp, PK2>(d, ap, B, sizeof(float) * (2 * T + 1), nullptr, st))) return rc;
  } else if (d->fused) {
    // query sum + energies + normalisation + context in one launch   (+ the whole decoder partial)
    ap.qpart = d->qpart, ap.n_qgroups = d->qpart_groups, ap.wlc_p = d->wlc_p, ap.wld_p = d->wld_p;
    pc = Piece{pd.g, split};
    pc.g.rb0 = 0, pc.g.nrb = pd.total_rb, pc.nk = pd.g.kparts;
    const size_t smem = sizeof(float) * FA_LDS_FLOATS;
    if (late_in_attn) {
      // + the late decoder-LSTM columns behind the step tags of this launch's own attention workgroups and pieces:
      // one launch boundary less
      typedef FusedAttnOp<1, ENC_DIM, 16> AttnOp;
      typedef StagedPieceOp<PK2 * GEMM_WAVES, STAGED_PARTS> PieceOp;
      typedef GemmOp<2, 2, 4, MODE_LSTM, false, true> LateOp;
      ap.late_flags = d->late_flags, pc.g.late_flags = d->late_flags;
      GemmP gl = g;
      gl.Wp = d->w_lstm2, gl.bias = d->b_lstm2, gl.kg0 = 0, gl.kg_stride = KG2;
      gl.seg[0] = Seg{{d->ctx, nullptr}, 0, ENC_DIM / 16};
      gl.out[0] = d->h_d[0], gl.out[1] = d->h_d[1], gl.c = d->c_d, gl.relu = 0;
      gl.nseg = 1, gl.KG = ENC_DIM / 16, gl.gates = d->g_d;
      gl.kparts = pd.g.kparts, gl.gates_part_stride = pd.g.gates_part_stride;
      gl.rb0 = 0, gl.nrb = DEC_DIM / 4, gl.late_flags = d->late_flags;
      const int nb = pc.g.nrb / GEMM_WAVES * pc.nk, nl = (DEC_DIM / 4) / 2 * (Bpad / 32);
      const size_t lds = std::max(std::max(smem, (size_t)PieceOp::LDS_BYTES), STAGED_LDS_MIN);
      static bool raised = false;
      if (!raised) {
        TTSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(triple_kernel_b_first<AttnOp, LateOp, PieceOp>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
      }
      triple_kernel_b_first<AttnOp, LateOp, PieceOp><<<B + nb + nl, GEMM_THREADS, lds, st>>>(ap, gl, pc.g, B, nb);
      TTSX_LAUNCH_CHECK();
    } else if (T <= 128 && split && Bpad >= 64) rc = launch_with_piece<FusedAttnOp<1, ENC_DIM, 16>, PK2>(d, ap, B, smem, &pc, st);
    else if (T <= 128) rc = launch_with_piece<FusedAttnOp<1, ENC_DIM>, PK2>(d, ap, B, smem, &pc, st);
    else if (T <= 256) rc = launch_with_piece<FusedAttnOp<2, ENC_DIM>, PK2>(d, ap, B, smem, &pc, st);
    else rc = launch_with_piece<FusedAttnOp<4, ENC_DIM>, PK2>(d, ap, B, smem, &pc, st);
    if (rc) return rc;
  } else {
  const bool share_q = d->share_q;
  if (share_q) {
    // no query launch: the energies op sums the shares   (decoder partial in two halves instead of three thirds)
    ap.qpart = d->qpart, ap.n_qgroups = d->qpart_groups;
  } else {
  // query projection   (+ first third of the decoder partial)
  g.Wp = d->w_query, g.bias = d->b_query, g.nseg = 1, g.KG = QUERY_DIM / 16, g.kg0 = 0, g.kg_stride = g.KG;
  g.gates = nullptr;
  g.seg[0] = Seg{{d->h_a[0], d->h_a[1]}, 2, QUERY_DIM / 16};
  g.out[0] = d->q, g.out[1] = nullptr, g.c = nullptr;
  pc = pd.piece(0);
  pc.on = split;
  if ((rc = launch_gemm<MODE_LINEAR, 8, PK2>(d, g, ATT_DIM / 16, false, &pc, st))) return rc;
  }
  // attention   (+ second and last third of the decoder partial)
  pc = share_q ? pd.piece(0, 2) : pd.piece(1);
  pc.on = split;
  if (B >= 16) {
    ap.gx = ceil_div(T, 32);
    rc = launch_with_piece<EnergyOp<32>, PK2>(d, ap, ap.gx * B, 0, &pc, st);
  } else {
    ap.gx = ceil_div(T, 16);
    rc = launch_with_piece<EnergyOp<16>, PK2>(d, ap, ap.gx * B, 0, &pc, st);
  }
  if (rc) return rc;
  pc = share_q ? pd.piece(1, 2) : pd.piece(2);
  pc.on = split;
  if (d->variant) {
    // windowing / forward attention: the whole tail of an utterance in one workgroup
    rc = launch_with_piece<AttnPostOp, PK2>(d, ap, B, sizeof(float) * 3 * T, nullptr, st);
  } else if (B >= 16) {
    ap.gx = ENC_DIM / 128;
    rc = launch_with_piece<ContextOp<128>, PK2>(d, ap, ap.gx * B, sizeof(float) * T, &pc, st);
  } else {
    ap.gx = ENC_DIM / 32;
    rc = launch_with_piece<ContextOp<32>, PK2>(d, ap, ap.gx * B, sizeof(float) * T, &pc, st);
  }
  if (rc) return rc;
  }
  // decoder LSTM: [ctx; h_a(new); h_d] (split: late columns [ctx] + the partial sums) + cell update
  //   (+ first third of the next attention partial)
  g.Wp = d->w_lstm2, g.bias = d->b_lstm2, g.kg0 = 0, g.kg_stride = KG2;
  g.seg[0] = Seg{{d->ctx, nullptr}, 0, ENC_DIM / 16};
  g.seg[1] = Seg{{d->h_a[0], d->h_a[1]}, 2, QUERY_DIM / 16};
  g.seg[2] = Seg{{d->h_d[0], d->h_d[1]}, 1, DEC_DIM / 16};
  g.out[0] = d->h_d[0], g.out[1] = d->h_d[1], g.c = d->c_d;
  pc = pa_next.piece(0);
  pc.on = split && more && !mfma;
  if (late_in_attn) {
    rc = TTSX_OK;  // (ran inside the attention launch)
  } else if (split) {
    g.nseg = 1, g.KG = ENC_DIM / 16, g.gates = d->g_d;
    g.kparts = pd.g.kparts, g.gates_part_stride = pd.g.gates_part_stride;
    rc = launch_gemm<MODE_LSTM, 4, PK1>(d, g, DEC_DIM / 4, true, &pc, st);
    g.kparts = 0;
  } else {
    g.nseg = 3, g.KG = KG2, g.gates = nullptr;
    rc = launch_gemm<MODE_LSTM, KG2 / GEMM_WAVES, PK1>(d, g, DEC_DIM / 4, true, nullptr, st);
  }
  if (rc) return rc;
  // projection: mel rows, stop row, next step's prenet layer-1 rows   (+ second third)
  g.Wp = d->w_proj, g.bias = d->b_proj, g.nseg = 2, g.KG = (DEC_DIM + ENC_DIM) / 16, g.kg0 = 0, g.kg_stride = g.KG;
  g.gates = nullptr;
  g.seg[0] = Seg{{d->h_d[0], d->h_d[1]}, 2, DEC_DIM / 16};
  g.seg[1] = Seg{{d->ctx, nullptr}, 0, ENC_DIM / 16};
  g.out[0] = d->o, g.out[1] = nullptr, g.c = nullptr, g.x1 = d->x1, g.o_quads = MEL_DIM * d->proj_r / 4;
  pc = mfma ? k_piece(pa_next, 0) : pa_next.piece(1);
  pc.on = split && more;
  if (fuse_p2 && more) {
    // + prenet layer 2 of the next step behind flags of this launch's own row blocks, + both K halves of the piece
    pc.nk = 2;
    GemmP g2 = g_pre2;
    g2.x1_flags = d->x1_flags;  // (step_in_chunk stays this step's: the tag the row blocks raise)
    g.x1_flags = d->x1_flags;
    const int nt = Bpad / 16, nrb = ceil_div(proj_rows(d->proj_r), 16);
    g.rb0 = 0, g.nrb = nrb;
    const int na = nrb * nt, nc = (PRENET_DIM / 16) * nt;
    typedef GemmOp<1, 1, 12, MODE_PROJ> ProjOp;
    if (!pc.on) {
      triple_kernel<ProjOp, Prenet2AfterProjOp, NoOp><<<na + nc, GEMM_THREADS, 0, st>>>(g, g2, 0, na, nc);
    } else if (nt % 2 == 0 && pc.g.nrb % 2 == 0) {
      const int nb = pc.g.nrb / 2 * (nt / 2) * pc.nk;
      triple_kernel_b_first<ProjOp, Prenet2AfterProjOp, GemmOp<2, 2, PK1 / 2, MODE_PARTIAL>>
          <<<na + nb + nc, GEMM_THREADS, 0, st>>>(g, g2, pc.g, na, nb);
    } else {
      const int nb = pc.g.nrb * nt * pc.nk;
      triple_kernel<ProjOp, Prenet2AfterProjOp, GemmOp<1, 1, PK1 / 2, MODE_PARTIAL>>
          <<<na + nc + nb, GEMM_THREADS, 0, st>>>(g, g2, pc.g, na, nc);
    }
    TTSX_LAUNCH_CHECK();
    return TTSX_OK;
  }
  return launch_gemm<MODE_PROJ, 12, PK1>(d, g, ceil_div(proj_rows(d->proj_r), 16), false, &pc, st);
}

int push_ctx(ttsx_t2_decoder* d, hipStream_t st) {
  TTSX_HIP(hipMemcpyAsync(d->rc, d->rc_host, sizeof(RunCtx), hipMemcpyHostToDevice, st));
  // the pinned staging copy is reused by the next call: make sure this one has been read
  TTSX_HIP(hipStreamSynchronize(st));
  return TTSX_OK;
}

// Launch `n` steps; a full chunk (n == chunk, CHUNK or CHUNK_SHORT) goes as one graph, captured once per
// (B, T, r, workspace, launch plan) and chunk size.
int launch_chunk(ttsx_t2_decoder* d, int n, int chunk, hipStream_t st) {
  if (n == chunk) {
    ttsx_t2_decoder::ChunkGraph& g = d->graphs[chunk == CHUNK ? 0 : chunk == CHUNK_LONG ? 2 : 1];   // (the cache key includes the chunk size)
    const int split = (int)d->split + 2 * (int)d->share_q + 4 * (d->plan_fuse_p2 + 1) + 16 * (int)d->fuse_gave_up + 32 * (int)d->staged;
    const bool cached = g.exec && g.B == d->B && g.T == d->T && g.r == d->rc_host->r && g.gen == d->ws_gen &&
                        g.chunk == n && g.split == split && g.fused == (int)d->fused;
    if (!cached) {
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
      hipGraph_t graph = nullptr;
      hipStream_t cs = d->cap_stream;
      TTSX_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
      int rc = TTSX_OK;
      for (int i = 0; i < n && !rc; ++i) rc = enqueue_step(d, i, i == 0, i + 1 < n, cs);
      if (!rc) ctx_advance_kernel<<<1, 1, 0, cs>>>(d->rc, n);
      hipError_t e = hipStreamEndCapture(cs, &graph);
      if (rc) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      TTSX_HIP(e);
      TTSX_HIP(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
      g.B = d->B, g.T = d->T, g.r = d->rc_host->r, g.gen = d->ws_gen, g.chunk = n, g.split = split, g.fused = d->fused;
    }
    TTSX_HIP(hipGraphLaunch(g.exec, st));
    return TTSX_OK;
  }
  for (int i = 0; i < n; ++i) {
    int rc = enqueue_step(d, i, i == 0, i + 1 < n, st);
    if (rc) return rc;
  }
  ctx_advance_kernel<<<1, 1, 0, st>>>(d->rc, n);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

// Busy-wait until the device has reported `want` completed steps or the end of every utterance in the pinned
// words.  The stream is queried now and then: a failed stream ends the wait with an error, a drained one with
// whatever the words hold.
int wait_progress(const volatile int* words, int want, hipStream_t st) {
  for (long spin = 0;; ++spin) {
    if (words[0] || words[1] >= want) return TTSX_OK;
    if ((spin & 0xfff) == 0xfff) {
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) return TTSX_OK;
      if (q != hipErrorNotReady) {
        ttsx_set_error("decoder stream failed while waiting for the stop flag: %s", hipGetErrorString(q));
        return TTSX_ERR_HIP;
      }
    }
    __builtin_ia32_pause();
  }
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
  float *w_ih2 = nullptr, *a2f = nullptr, *b2f = nullptr;  // temporaries of the packing
  ScopeExit on_fail([&] { (void)ttsx_t2_decoder_destroy(d); });
  ScopeExit free_tmp([&] { (void)hipFree(w_ih2), (void)hipFree(a2f), (void)hipFree(b2f); });
  d->r_init = r_init, d->norm = attn_norm, d->odim = MEL_DIM * r_init;
  d->plan_fused = d->plan_split = d->plan_share_q = d->plan_team = d->plan_fuse_p2 = d->plan_staged = -1;
  int rc = 0;
  rc |= pack_layer(&d->w_lstm1, &d->b_lstm1, w->attn_rnn_w_ih, PRENET_DIM + ENC_DIM, w->attn_rnn_w_hh, QUERY_DIM,
                   w->attn_rnn_b_ih, w->attn_rnn_b_hh, 4 * QUERY_DIM, QUERY_DIM, st);
  // decoder LSTM columns reordered to [ctx; h_a; h_d] so that the late (context) columns and the
  // early ones are each one contiguous run of k-groups
  if (dev_alloc(&w_ih2, (size_t)4 * DEC_DIM * (QUERY_DIM + ENC_DIM))) return TTSX_ERR_HIP;
  swap_cols_kernel<<<4 * DEC_DIM, 256, 0, st>>>(w_ih2, w->dec_rnn_w_ih, QUERY_DIM, ENC_DIM);
  TTSX_LAUNCH_CHECK();
  rc |= pack_layer(&d->w_lstm2, &d->b_lstm2, w_ih2, ENC_DIM + QUERY_DIM, w->dec_rnn_w_hh, DEC_DIM,
                   w->dec_rnn_b_ih, w->dec_rnn_b_hh, 4 * DEC_DIM, DEC_DIM, st);
  // (a decoder with GravesAttention has no query / inputs / v / location tensors)
  if (w->query_w)
    rc |= pack_layer(&d->w_query, &d->b_query, w->query_w, QUERY_DIM, nullptr, 0, nullptr, nullptr, ATT_DIM, 0, st);
  if (rc) return TTSX_ERR_HIP;
  // prenet: fold BatchNorm, keep layer 1 in source layout for the composed projection rows
  if (dev_alloc(&d->a1f, PRENET_DIM * MEL_DIM) || dev_alloc(&d->b1f, PRENET_DIM) ||
      dev_alloc(&a2f, PRENET_DIM * PRENET_DIM) || dev_alloc(&b2f, PRENET_DIM))
    return TTSX_ERR_HIP;
  fold_prenet_kernel<<<ceil_div(PRENET_DIM * MEL_DIM, 256), 256, 0, st>>>(
      d->a1f, d->b1f, w->prenet_w[0], w->prenet_bn_weight[0], w->prenet_bn_bias[0], w->prenet_bn_mean[0],
      w->prenet_bn_var[0], PRENET_DIM, MEL_DIM);
  TTSX_LAUNCH_CHECK();
  fold_prenet_kernel<<<ceil_div(PRENET_DIM * PRENET_DIM, 256), 256, 0, st>>>(
      a2f, b2f, w->prenet_w[1], w->prenet_bn_weight[1], w->prenet_bn_bias[1], w->prenet_bn_mean[1],
      w->prenet_bn_var[1], PRENET_DIM, PRENET_DIM);
  TTSX_LAUNCH_CHECK();
  rc |= pack_layer(&d->w_pre2, &d->b_pre2, a2f, PRENET_DIM, nullptr, 0, b2f, nullptr, PRENET_DIM, 0, st);
  {
    T2TeamWeights tw{};
    tw.attn_w_ih = w->attn_rnn_w_ih, tw.attn_w_hh = w->attn_rnn_w_hh, tw.attn_b_ih = w->attn_rnn_b_ih;
    tw.attn_b_hh = w->attn_rnn_b_hh, tw.dec_w_ih = w->dec_rnn_w_ih, tw.dec_w_hh = w->dec_rnn_w_hh;
    tw.dec_b_ih = w->dec_rnn_b_ih, tw.dec_b_hh = w->dec_rnn_b_hh, tw.query_w = w->query_w, tw.v_w = w->v_w;
    tw.v_b = w->v_b, tw.loc_conv_w = w->loc_conv_w, tw.loc_dense_w = w->loc_dense_w;
    tw.a2f = a2f, tw.b2f = b2f, tw.b1f = d->b1f;
    const int trc = t2_team_create(&tw, st, &d->team);
    if (trc) return trc;
  }
  rc |= copy_dev(&d->proj_w_src, w->proj_w, (size_t)d->odim * (DEC_DIM + ENC_DIM), st);
  rc |= copy_dev(&d->proj_b_src, w->proj_b, d->odim, st);
  rc |= copy_dev(&d->stop_w_src, w->stop_w, DEC_DIM + d->odim, st);
  rc |= copy_dev(&d->stop_b_src, w->stop_b, 1, st);
  if (w->inputs_w) rc |= copy_dev(&d->inputs_w, w->inputs_w, ATT_DIM * ENC_DIM, st);
  if (w->v_w) rc |= copy_dev(&d->v_w, w->v_w, ATT_DIM, st);
  if (w->v_b) rc |= copy_dev(&d->v_b, w->v_b, 1, st);
  rc |= dev_alloc(&d->loc_conv_w, LOC_F * 2 * LOC_K);
  rc |= dev_alloc(&d->loc_dense_w, ATT_DIM * LOC_F);
  if (rc) return TTSX_ERR_HIP;
  if (w->loc_conv_w && w->loc_dense_w) {
    transpose_loc_kernel<<<ceil_div(ATT_DIM * LOC_F, 256), 256, 0, st>>>(d->loc_conv_w, d->loc_dense_w,
                                                                        w->loc_conv_w, w->loc_dense_w);
    TTSX_LAUNCH_CHECK();
  } else {
    // location_attn = false (OriginalAttention.get_attention, Tacotron2/TTS/models.py:269-273): the
    // location term is absent, which all-zero location weights reproduce exactly
    TTSX_HIP(hipMemsetAsync(d->loc_conv_w, 0, sizeof(float) * LOC_F * 2 * LOC_K, st));
    TTSX_HIP(hipMemsetAsync(d->loc_dense_w, 0, sizeof(float) * ATT_DIM * LOC_F, st));
  }
  if (w->query_w) {
    if (dev_alloc(&d->wlc_p, 2 * 16 * 64) || dev_alloc(&d->wld_p, 8 * 8 * 64) ||
        dev_alloc(&d->wq_t, (size_t)QUERY_DIM * ATT_DIM))
      return TTSX_ERR_HIP;
    pack_attn_frag_kernel<<<ceil_div(QUERY_DIM * ATT_DIM, 256), 256, 0, st>>>(
        d->wlc_p, d->wld_p, d->wq_t, w->loc_conv_w, w->loc_conv_w ? w->loc_dense_w : nullptr, w->query_w, QUERY_DIM);
    TTSX_LAUNCH_CHECK();
    TTSX_HIP(set_fused_lds());
  }
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&d->rc), sizeof(RunCtx)));
  TTSX_HIP(hipMemsetAsync(d->rc, 0, sizeof(RunCtx), st));
  TTSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&d->rc_host), sizeof(RunCtx), hipHostMallocDefault));
  TTSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&d->host_flag), 4 * sizeof(int), hipHostMallocMapped));
  TTSX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&d->host_words_dev), d->host_flag, 0));
  memset(d->host_flag, 0, 4 * sizeof(int));
  memset(d->rc_host, 0, sizeof(RunCtx));
  TTSX_HIP(hipStreamCreateWithFlags(&d->cap_stream, hipStreamNonBlocking));
  TTSX_HIP(hipStreamSynchronize(st));
  rc = ensure_proj(d, r_init, st);
  if (rc) return rc;
  on_fail.dismiss();
  *out = d;
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_destroy(ttsx_t2_decoder* d) {
  if (!d) return TTSX_OK;
  free_workspace(d);
  t2_team_destroy(d->team);
  float* bufs[] = {d->w_lstm1, d->b_lstm1, d->w_lstm2, d->b_lstm2, d->w_query, d->b_query,
                   d->w_pre2,  d->b_pre2,  d->w_proj,  d->b_proj,  d->proj_w_src, d->proj_b_src, d->stop_w_src,
                   d->stop_b_src, d->a1f, d->b1f, d->inputs_w, d->v_w, d->v_b, d->loc_conv_w, d->loc_dense_w,
                   d->ta_w, d->ta_b, d->w_na1, d->b_na1, d->w_na2, d->b_na2, d->mem0, d->wlc_p, d->wld_p, d->wq_t};
  for (float* b : bufs)
    if (b) (void)hipFree(b);
  if (d->cap_stream) (void)hipStreamDestroy(d->cap_stream);
  if (d->rc) (void)hipFree(d->rc);
  if (d->rc_host) (void)hipHostFree(d->rc_host);
  if (d->host_flag) (void)hipHostFree(d->host_flag);
  delete d;
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_set_attention(ttsx_t2_decoder* d, const ttsx_t2_attention_options* o, void* stream) {
  TTSX_ARG(d && o, "ttsx_t2_decoder_set_attention: null argument");
  TTSX_ARG(!o->trans_agent || (o->forward_attn && o->ta_w && o->ta_b),
           "ttsx_t2_decoder_set_attention: the transition agent needs forward attention and its weights");
  TTSX_ARG(o->graves_K >= 0 && o->graves_K <= GRAVES_MAX_K, "ttsx_t2_decoder_set_attention: graves_K=%d outside 0..%d",
           o->graves_K, GRAVES_MAX_K);
  TTSX_ARG(o->graves_K == 0 || (o->graves_w1 && o->graves_b1 && o->graves_w2 && o->graves_b2),
           "ttsx_t2_decoder_set_attention: GravesAttention weights missing");
  TTSX_ARG(o->graves_K == 0 || !(o->windowing || o->forward_attn),
           "ttsx_t2_decoder_set_attention: windowing / forward attention belong to the original attention");
  hipStream_t st = static_cast<hipStream_t>(stream);
  d->windowing = o->windowing != 0, d->forward_attn = o->forward_attn != 0;
  d->forward_attn_mask = o->forward_attn_mask != 0, d->trans_agent = o->trans_agent != 0;
  d->graves_K = o->graves_K;
  d->variant = d->windowing || d->forward_attn;
  int rc = 0;
  if (d->trans_agent) {
    if (d->ta_w) (void)hipFree(d->ta_w), (void)hipFree(d->ta_b);
    d->ta_w = d->ta_b = nullptr;
    rc |= copy_dev(&d->ta_w, o->ta_w, ENC_DIM + QUERY_DIM, st);
    rc |= copy_dev(&d->ta_b, o->ta_b, 1, st);
  }
  if (d->graves_K > 0) {
    float** old[] = {&d->w_na1, &d->b_na1, &d->w_na2, &d->b_na2};  // sizes depend on K: pack afresh
    for (float** b : old) {
      if (*b) (void)hipFree(*b);
      *b = nullptr;
    }
    rc |= pack_layer(&d->w_na1, &d->b_na1, o->graves_w1, QUERY_DIM, nullptr, 0, o->graves_b1, nullptr, QUERY_DIM, 0, st);
    rc |= pack_layer(&d->w_na2, &d->b_na2, o->graves_w2, QUERY_DIM, nullptr, 0, o->graves_b2, nullptr,
                     3 * d->graves_K, 0, st);
  }
  if (rc) return TTSX_ERR_HIP;
  TTSX_HIP(hipStreamSynchronize(st));
  drop_graph(d);
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_set_initial(ttsx_t2_decoder* d, int keep_states, const float* memory, int B,
                                           void* stream) {
  TTSX_ARG(d, "ttsx_t2_decoder_set_initial: null handle");
  TTSX_ARG(B >= 1 && B <= 4096, "ttsx_t2_decoder_set_initial: B=%d", B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  d->keep_next = keep_states != 0;
  d->mem0_set = false;
  if (memory) {
    if (d->mem0_cap < B) {
      if (d->mem0) (void)hipFree(d->mem0);
      d->mem0 = nullptr;
      if (dev_alloc(&d->mem0, (size_t)B * MEL_DIM)) return TTSX_ERR_HIP;
      d->mem0_cap = B;
    }
    TTSX_HIP(hipMemcpyAsync(d->mem0, memory, sizeof(float) * (size_t)B * MEL_DIM, hipMemcpyDeviceToDevice, st));
    d->mem0_set = true;
  }
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_begin(ttsx_t2_decoder* d, const float* enc, const int32_t* enc_len, int B, int T,
                                     void* stream) {
  TTSX_ARG(d && enc, "ttsx_t2_decoder_begin: null argument");
  TTSX_ARG(B >= 1 && B <= 4096 && T >= 1, "ttsx_t2_decoder_begin: bad shape B=%d T=%d", B, T);
  TTSX_ARG(T <= 16000, "ttsx_t2_decoder_begin: encoder length %d exceeds the LDS budget of the context kernel", T);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = ensure_workspace(d, B, T);
  if (rc) return rc;
  const bool keep = d->keep_next && d->has_state && d->B == B;
  if (d->keep_next && !keep) {
    d->keep_next = false, d->mem0_set = false;
    ttsx_set_error("ttsx_t2_decoder_begin: no recurrent state of a %d-utterance run to keep", B);
    return TTSX_ERR_STATE;
  }
  const bool graves = d->graves_K > 0;
  if (!graves && !(d->w_query && d->inputs_w && d->v_w && d->v_b)) {
    ttsx_set_error("ttsx_t2_decoder_begin: attention weights missing (query / inputs / v)");
    return TTSX_ERR_ARG;
  }
  TTSX_ARG(!(d->variant || graves) || T <= 12000,
           "ttsx_t2_decoder_begin: encoder length %d exceeds the LDS budget of the attention variants", T);
  d->B = B, d->T = T;
  {
    // The launch plan of a run, by shape; ttsx_t2_decoder_set_plan overrides each choice (A/B timing, and the
    // parity tests that hold the alternatives against each other).
    // Early/late LSTM split: always in the GEMV regime, where every launch is latency-bound (rule for the MFMA
    // engine below).
    // One-workgroup-per-utterance attention: one workgroup walks all positions of its utterance through ONE CU's
    // vector memory pipe (~2 KB per position), so it pays up to a length that depends on how much of the chip the
    // batch fills.  Measured per step, fused / three launches:
    //   B=1:  T=64 30.6 / 33.0 us, T=128 32.5 / 33.7, T=250 39.9 / 33.5      -> up to 128 positions at batch 1-2
    //   B=16: T=128 37.9 / 40.7, T=512 59 / 49;  B=64: T=128 51.3 / 54.5, T=256 58.3 / 62.9, T=512 72 / 79;
    //   B=128: T=128 79.4 / 87.0                                              -> 256, and 512 from batch 48 on
    const int t_max = B >= 48 ? FA_TMAX : (B > GEMV_MAX_B || d->plan_fused == 1) ? 256 : 128;
    d->fused = T <= t_max && !d->variant && !graves && d->plan_fused != 0;
    // Batch 1-2 without the fused kernel: the query launch is dropped and the energies workgroups add up the
    // query shares themselves.
    d->share_q = B <= GEMV_MAX_B && !d->fused && !d->variant && !graves && d->plan_share_q != 0;
    // On the MFMA engine the split needs the fused attention launch (one CU per utterance busy for ~11 us: the
    // shadow the early decoder-LSTM columns run in; the three-launch attention fills the chip itself), and it
    // pays while the pieces fit into that shadow.  Measured per step, split / whole LSTM launches (T=128 unless
    // noted): B=4 34.1 / 37.2 us, B=16 33.7 / 37.4, B=32 38.2 / 41.0, B=32 T=256 45.2 / 48.8, B=64 T=64 48.1 / 47.6,
    // B=64 48.1 / 49.1, B=64 T=200 50.4 / 56.7, B=96 75.0 / 73.6.
    // Staged pieces for the decoder LSTM's early columns (StagedPieceOp): 64 columns exactly (the slice of a K
    // part is then one contiguous block of the quad layout).  Measured per step, staged / 32 x 32 register tiles:
    // B=64 T=128 45.3 / 46.5 us, B=48 45.3 / 46.3, B=64 T=64 45.3 (49.1 without the split), B=64 T=200 50.8 / 49.8
    // (beyond 128 positions the attention workgroup is the longer one either way) -> up to 128 positions.
    const bool can_stage = B > GEMV_MAX_B && d->Bpad == 64 && d->fused &&
                           (d->plan_staged == 1 || (d->plan_staged == -1 && T <= 128));
    const bool pays = d->Bpad <= 32 || (d->Bpad == 64 && (T > 64 || can_stage)) || d->plan_split == 1;
    d->split = (B <= GEMV_MAX_B || (d->fused && pays)) && d->plan_split != 0 && !d->variant && !graves;
    d->staged = can_stage && d->split;
  }
  d->parity0 = 0;
  const size_t cb = d->Bpad;
  TTSX_HIP(hipMemsetAsync(d->x2, 0, sizeof(float) * PRENET_DIM * cb, st));
  if (keep) {
    // Decoder._init_states(keep_states=True), Tacotron2/TTS/models.py:536-552: the LSTM states and the
    // context stay; the current hidden states move to the buffers a run starts from
    if (d->steps_done & 1) {
      TTSX_HIP(hipMemcpyAsync(d->h_a[0], d->h_a[1], sizeof(float) * QUERY_DIM * cb, hipMemcpyDeviceToDevice, st));
      TTSX_HIP(hipMemcpyAsync(d->h_d[0], d->h_d[1], sizeof(float) * DEC_DIM * cb, hipMemcpyDeviceToDevice, st));
    }
  } else {
    TTSX_HIP(hipMemsetAsync(d->ctx, 0, sizeof(float) * ENC_DIM * cb, st));
    for (int i = 0; i < 2; ++i) {
      TTSX_HIP(hipMemsetAsync(d->h_a[i], 0, sizeof(float) * QUERY_DIM * cb, st));
      TTSX_HIP(hipMemsetAsync(d->h_d[i], 0, sizeof(float) * DEC_DIM * cb, st));
    }
    TTSX_HIP(hipMemsetAsync(d->c_a, 0, sizeof(float) * QUERY_DIM * cb, st));
    TTSX_HIP(hipMemsetAsync(d->c_d, 0, sizeof(float) * DEC_DIM * cb, st));
  }
  TTSX_HIP(hipMemsetAsync(d->q, 0, sizeof(float) * ATT_DIM * cb, st));
  TTSX_HIP(hipMemsetAsync(d->o, 0, sizeof(float) * (size_t)round_up(proj_rows(d->r_init), 16) * cb, st));
  TTSX_HIP(hipMemsetAsync(d->n1, 0, sizeof(float) * QUERY_DIM * cb, st));
  TTSX_HIP(hipMemsetAsync(d->gbk, 0, sizeof(float) * (size_t)round_up(3 * GRAVES_MAX_K, 16) * cb, st));
  TTSX_HIP(hipMemsetAsync(d->x1_flags, 0, sizeof(unsigned) * ceil_div((int)cb, 16) * (PRENET_DIM / 16), st));
  TTSX_HIP(hipMemsetAsync(d->late_flags, 0, sizeof(unsigned) * (2 * cb + (size_t)STAGED_PARTS * (DEC_DIM / 4)), st));
  TTSX_HIP(hipMemsetAsync(d->a_prev, 0, sizeof(float) * (size_t)B * T, st));
  TTSX_HIP(hipMemsetAsync(d->a_cum, 0, sizeof(float) * (size_t)B * T, st));
  if (d->mem0_set) {
    // first step fed a given frame instead of the all-zero go frame
    prenet1_explicit_kernel<<<ceil_div(B * PRENET_DIM, 256), 256, 0, st>>>(d->x1, d->mem0, d->a1f, d->b1f, B, (int)cb);
  } else if (!keep) {
    init_x1_kernel<<<ceil_div(PRENET_DIM * (int)cb, 256), 256, 0, st>>>(d->x1, d->b1f, (int)cb);
  }  // (keep without a frame: x1 already holds layer 1 of the last emitted frame)
  TTSX_LAUNCH_CHECK();
  d->keep_next = false, d->mem0_set = false;
  if (3 * sizeof(float) * T > 48 * 1024) {
    TTSX_HIP(set_context_lds(sizeof(float) * T, d->variant, graves));
  }
  TTSX_HIP(hipMemcpyAsync(d->enc_ws, enc, sizeof(float) * (size_t)B * T * ENC_DIM, hipMemcpyDeviceToDevice, st));
  fill_len_kernel<<<ceil_div(B, 256), 256, 0, st>>>(d->len_ws, enc_len, B, T);
  TTSX_LAUNCH_CHECK();
  init_variant_state_kernel<<<ceil_div(std::max(B * T, B * GRAVES_MAX_K), 256), 256, 0, st>>>(
      d->alpha, d->u_fa, d->win_idx, d->mu, d->len_ws, B, T);
  TTSX_LAUNCH_CHECK();
  if (!graves) {
    inputs_proj_kernel<<<dim3(ceil_div(T, 16), B), 256, 0, st>>>(d->P, d->enc_ws, d->inputs_w, T);
    TTSX_LAUNCH_CHECK();
  }
  RunCtx* h = d->rc_host;
  memset(h, 0, sizeof(*h));
  h->B = B, h->T = T, h->r = d->proj_r, h->threshold = 2.f, h->done_at = NEVER_DONE;
  d->steps_done = 0;
  d->begun = true;
  d->has_state = true;
  return push_ctx(d, st);
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
  int rc = ensure_proj(d, r, st);
  if (rc) return rc;
  RunCtx* h = d->rc_host;
  h->base = d->steps_done, h->out_base = 0, h->all_done = 0, h->done_at = NEVER_DONE, h->n_done = 0;
  h->r = r, h->free_running = 0, h->threshold = 2.f, h->host_words = nullptr;
  h->mel = frames, h->mel_rows = r, h->align = align, h->align_rows = 1, h->stop = stop, h->stop_rows = 1;
  // n_steps bookkeeping is not wanted here: point it at the energies scratch, never read back
  h->n_steps = reinterpret_cast<int32_t*>(d->energies);
  TTSX_HIP(hipMemsetAsync(d->energies, 0, sizeof(float) * d->B, st));
  rc = push_ctx(d, st);
  if (rc) return rc;
  prenet1_explicit_kernel<<<ceil_div(d->B * PRENET_DIM, 256), 256, 0, st>>>(d->x1, memory, d->a1f, d->b1f, d->B,
                                                                           d->Bpad);
  TTSX_LAUNCH_CHECK();
  d->parity0 = d->steps_done & 1;
  rc = enqueue_step(d, 0, true, false, st);
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
  int rc = ensure_proj(d, r, st);
  if (rc) return rc;
  const bool carried = d->keep_next || d->mem0_set;  // state or a first frame handed over by set_initial
  rc = ttsx_t2_decoder_begin(d, enc, enc_len, B, T, stream);
  if (rc) return rc;
  TTSX_HIP(hipMemsetAsync(n_steps, 0, sizeof(int32_t) * B, st));
  // One or two utterances of up to 1024 symbols with the plain location-sensitive attention: the run of an
  // utterance is one persistent launch of a 216-workgroup team (240 beyond 256 symbols) that keeps every weight in registers and LDS
  // (t2_team.hip; 11.7 us per step against 31.7 for the chain at batch 1, and two such runs one after the other
  // still beat the chain's 35 us per step at batch 2).
  if (B <= 2 && t2_team_supports(d->team, 1, T, r) && !carried && !d->variant && d->graves_K == 0 &&
      (d->plan_team == 1 || (d->plan_team == -1 && !d->team_cool.cooling()))) {
    T2TeamState so{};
    so.h_a = d->h_a[0], so.h_a_alt = d->h_a[1], so.c_a = d->c_a, so.h_d = d->h_d[0], so.h_d_alt = d->h_d[1];
    so.c_d = d->c_d, so.ctx = d->ctx, so.x1 = d->x1;
    int run = 0, run_of[2] = {0, 0};
    for (int b = 0; b < B && rc == TTSX_OK; ++b) {
      rc = t2_team_run(d->team, d->enc_ws + (size_t)b * T * ENC_DIM, d->P + (size_t)b * T * ATT_DIM, d->len_ws + b, T, r,
                       max_steps, stop_threshold, fixed_steps, d->norm, mel + (size_t)b * max_steps * r * MEL_DIM,
                       align + (size_t)b * max_steps * T, stop + (size_t)b * max_steps, n_steps + b,
                       B == 1 ? &so : nullptr, &run_of[b], st);
      run = std::max(run, run_of[b]);
    }
    // Two utterances are two runs, each ending at its own stop frame: the rows of the earlier one between its
    // stop and the later one's are zero-filled, so that everything below *steps_run_host is defined.
    for (int b = 0; b < B && rc == TTSX_OK; ++b) {
      const size_t a = run_of[b], n = (size_t)run - a;
      if (n == 0) continue;
      TTSX_HIP(hipMemsetAsync(mel + ((size_t)b * max_steps + a) * r * MEL_DIM, 0, sizeof(float) * n * r * MEL_DIM, st));
      TTSX_HIP(hipMemsetAsync(align + ((size_t)b * max_steps + a) * T, 0, sizeof(float) * n * T, st));
      TTSX_HIP(hipMemsetAsync(stop + (size_t)b * max_steps + a, 0, sizeof(float) * n, st));
    }
    if (rc == TTSX_OK) {
      if (steps_run_host) *steps_run_host = run;
      d->last_launched = run, d->steps_done = run, d->begun = false;
      d->has_state = B == 1;  // (the final state is written back in the chain's batch-1 layout only)
      return TTSX_OK;
    }
    // TTSX_ERR_STATE: the team did not become resident together and its bounded waits gave up.  Under the default
    // plan the launch chain starts over from a fresh state, and the handle tries the team again after a cool-down.
    if (rc != TTSX_ERR_STATE || d->plan_team == 1) return rc;
    d->team_cool.start();
    rc = ttsx_t2_decoder_begin(d, enc, enc_len, B, T, stream);
    if (rc) return rc;
    TTSX_HIP(hipMemsetAsync(n_steps, 0, sizeof(int32_t) * B, st));
  }
  RunCtx* h = d->rc_host;
  h->r = r, h->free_running = fixed_steps ? 0 : 1, h->threshold = stop_threshold;
  h->mel = mel, h->mel_rows = max_steps * r, h->align = align, h->align_rows = max_steps;
  h->stop = stop, h->stop_rows = max_steps, h->n_steps = n_steps;
  h->sync_failed = ttsx_take_wait_failure(TTSX_WAIT_T2_STEP_FUSION) ? 1 : 0;
  h->host_words = fixed_steps ? nullptr : d->host_words_dev;
  d->host_flag[0] = 0, d->host_flag[1] = 0;
  rc = push_ctx(d, st);
  if (rc) return rc;
  int launched = 0;
  if (fixed_steps) {
    while (launched < max_steps) {
      const int chunk = max_steps - launched >= CHUNK_LONG ? CHUNK_LONG : CHUNK;
      const int n = std::min(chunk, max_steps - launched);
      if ((rc = launch_chunk(d, n, chunk, st))) return rc;
      launched += n;
    }
  } else {
    // chunks in flight: two short ones at batch 1-2, one long one otherwise
    int chunk = CHUNK, ahead = CHUNK;
    if (B <= GEMV_MAX_B) {
      chunk = d->poll_chunk ? d->poll_chunk : CHUNK_SHORT;
      ahead = d->poll_ahead ? d->poll_ahead : chunk;
    }
    const volatile int* words = d->host_flag;
    while (launched < max_steps && !words[0]) {
      // enqueue while fewer than `ahead` steps are outstanding, else wait for the device to report progress
      if (launched - words[1] >= ahead) {
        if ((rc = wait_progress(words, launched - ahead + 1, st))) return rc;
        if (!words[0] && launched - words[1] >= ahead) {   // the stream drained without reporting
          ttsx_set_error("ttsx_t2_decoder_run: the device reported no progress for the enqueued steps");
          return TTSX_ERR_STATE;
        }
        continue;
      }
      const int n = std::min(chunk, max_steps - launched);
      if ((rc = launch_chunk(d, n, chunk, st))) return rc;
      launched += n;
    }
  }
  finalize_steps_kernel<<<ceil_div(B, 256), 256, 0, st>>>(n_steps, B, launched);
  TTSX_LAUNCH_CHECK();
  std::vector<int32_t> host(B);
  int sync_failed = 0;
  TTSX_HIP(hipMemcpyAsync(host.data(), n_steps, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipMemcpyAsync(&sync_failed, &d->rc->sync_failed, sizeof(int), hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipStreamSynchronize(st));
  if (sync_failed) {
    // (workgroups of one launch that were not resident together.)  Under the default plan the handle stops fusing
    // and the run starts over with prenet layer 2 and the late decoder LSTM as launches of their own; state handed over by set_initial was
    // consumed by the failed attempt, so such a run reports the error instead.
    if (d->plan_fuse_p2 == -1 && !d->fuse_gave_up && !carried) {
      d->fuse_gave_up = true;
      return ttsx_t2_decoder_run(d, enc, enc_len, B, T, r, max_steps, stop_threshold, fixed_steps, mel, align, stop,
                                 n_steps, steps_run_host, stream);
    }
    ttsx_set_error("ttsx_t2_decoder_run: a workgroup waited too long for the step tags of its own launch (ttsx_t2_decoder_set_fusion)");
    return TTSX_ERR_STATE;
  }
  int run = 0;
  for (int b = 0; b < B; ++b) run = std::max(run, (int)host[b]);
  if (fixed_steps) run = launched;
  if (steps_run_host) *steps_run_host = run;
  d->last_launched = launched;
  d->steps_done = run;
  d->begun = false;  // the state belongs to a finished batch
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_set_plan(ttsx_t2_decoder* d, int fused_attention, int split_lstm, int share_query) {
  TTSX_ARG(d, "ttsx_t2_decoder_set_plan: null handle");
  for (int v : {fused_attention, split_lstm, share_query})
    TTSX_ARG(v >= -1 && v <= 1, "ttsx_t2_decoder_set_plan: %d is not -1 (by shape), 0 (off) or 1 (on)", v);
  d->plan_fused = fused_attention, d->plan_split = split_lstm, d->plan_share_q = share_query;
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_set_fusion(ttsx_t2_decoder* d, int prenet_in_projection) {
  TTSX_ARG(d, "ttsx_t2_decoder_set_fusion: null handle");
  TTSX_ARG(prenet_in_projection >= -1 && prenet_in_projection <= 2,
           "ttsx_t2_decoder_set_fusion: %d is not -1 (default), 0 (off), 1 (prenet layer 2) or 2 (+ late decoder LSTM)",
           prenet_in_projection);
  d->plan_fuse_p2 = prenet_in_projection;
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_set_pieces(ttsx_t2_decoder* d, int staged) {
  TTSX_ARG(d, "ttsx_t2_decoder_set_pieces: null handle");
  TTSX_ARG(staged >= -1 && staged <= 1, "ttsx_t2_decoder_set_pieces: the argument is -1 (by shape), 0 or 1");
  d->plan_staged = staged;
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_set_team(ttsx_t2_decoder* d, int team) {
  TTSX_ARG(d, "ttsx_t2_decoder_set_team: null handle");
  TTSX_ARG(team >= -1 && team <= 1, "ttsx_t2_decoder_set_team: %d is not -1 (by shape), 0 (never) or 1 (wherever possible)", team);
  d->plan_team = team;
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_set_polling(ttsx_t2_decoder* d, int chunk, int ahead) {
  TTSX_ARG(d, "ttsx_t2_decoder_set_polling: null handle");
  TTSX_ARG(chunk == 0 || (chunk >= 2 && chunk <= CHUNK && chunk % 2 == 0),
           "ttsx_t2_decoder_set_polling: chunk=%d must be 0 or an even number in 2..%d", chunk, CHUNK);
  TTSX_ARG(ahead == 0 || (chunk > 0 && ahead >= chunk && ahead <= 4 * CHUNK),
           "ttsx_t2_decoder_set_polling: ahead=%d must be 0 or in chunk..%d", ahead, 4 * CHUNK);
  d->poll_chunk = chunk, d->poll_ahead = ahead;
  return TTSX_OK;
}

extern "C" int ttsx_t2_decoder_steps_launched(const ttsx_t2_decoder* d) {
  TTSX_ARG(d, "ttsx_t2_decoder_steps_launched: null handle");
  return d->last_launched;
}

// ====================================================================== encoder BiLSTM
// Encoder.inference's bidirectional LSTM (Tacotron2/TTS/models.py:419-454: nn.LSTM(512, 256,
// bidirectional, batch_first)) on the same row-block engine.
//   * input projection of every time step at once: [1024 x 512] x [512 x T*Bc] per direction,
//     a MODE_PARTIAL launch whose "utterance" axis is (t, b) (column t*Bc + b);
//   * T step launches; each is a dual launch (forward direction at t = s, backward at t = T-1-s)
//     of the K = 256 recurrent product with the cell update in the epilogue, which adds the
//     input-projection sums of its time step and writes h to out[b][t][dir*256 + unit].
// Utterances shorter than T hold zero state outside their length (what packing the sequence does).
namespace {

constexpr int ENC_H = 256, ENC_IN = 512;

// y [B][T][512] (frame-major) -> quad layout [128][Ct] float4 with column t*Bc + b; padding columns zero
__global__ void enc_to_quad_kernel(float4* __restrict__ xq, const float4* __restrict__ y, int B, int T, int Bc, int Ct) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)Ct * (ENC_IN / 4)) return;
  const int k4 = (int)(i % (ENC_IN / 4)), col = (int)(i / (ENC_IN / 4));
  const int t = col / Bc, b = col % Bc;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < T && b < B) v = y[((size_t)b * T + t) * (ENC_IN / 4) + k4];
  xq[(size_t)k4 * Ct + col] = v;
}

}  // namespace

struct ttsx_encoder_lstm {
  float *w_ih[2], *b_unused[2], *w_hh[2], *bias[2];
  RunCtx* rc;
  // workspace
  int cap_B, cap_T, Bc, Ct;
  float *xq, *gi[2], *h[2][2], *c[2];
  // the recurrence as one persistent launch (enc_scan.hip)
  float* scan_hx;
  unsigned* scan_steps;
  int* scan_abort;
  int plan_scan;            // ttsx_encoder_lstm_set_plan: -1 by shape (default), 0 launch chain, 1 scan wherever possible
  TeamCoolDown scan_cool;   // a scan whose workgroups were not resident together: launch chain for a while
};

namespace {

void enc_lstm_free_ws(ttsx_encoder_lstm* e) {
  float** bufs[] = {&e->xq, &e->gi[0], &e->gi[1], &e->h[0][0], &e->h[0][1], &e->h[1][0], &e->h[1][1], &e->c[0], &e->c[1],
                    &e->scan_hx};
  for (float** b : bufs) {
    if (*b) (void)hipFree(*b);
    *b = nullptr;
  }
  if (e->scan_steps) (void)hipFree(e->scan_steps);
  if (e->scan_abort) (void)hipFree(e->scan_abort);
  e->scan_steps = nullptr, e->scan_abort = nullptr;
  e->cap_B = e->cap_T = 0;
}

// one launch of the engine with a GemmP prepared by the caller; cols = column count, step = recurrent step launch
template <int MODE, int KGW>
int enc_launch_one(const GemmP& g, int nrb, int Bc, int cols, hipStream_t st) {
  const int nt_total = cols / 16;
  if (Bc == 1 && MODE == MODE_LSTM) dual_kernel<GemvOp<1, KGW, MODE>, NoOp><<<nrb, GEMM_THREADS, 0, st>>>(g, 0, nrb);
  else if (Bc == 2 && MODE == MODE_LSTM) dual_kernel<GemvOp<2, KGW, MODE>, NoOp><<<nrb, GEMM_THREADS, 0, st>>>(g, 0, nrb);
  else if (MODE == MODE_PARTIAL && nrb % 4 == 0 && nt_total % 4 == 0 && nrb / 4 * (nt_total / 4) >= 512) {
    // the input projection of a whole batch (thousands of columns): 64 x 64 tiles, combine buffer in dynamic LDS
    typedef GemmOp<4, 4, KGW, MODE, true> Op;
    static bool raised = false;
    if (!raised) {
      TTSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<Op, NoOp>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM44_LDS_BYTES));
      raised = true;
    }
    const int n = nrb / 4 * (nt_total / 4);
    dual_kernel<Op, NoOp><<<n, GEMM_THREADS, GEMM44_LDS_BYTES, st>>>(g, 0, n);
  } else if (nt_total % 2 == 0 && nrb / 2 * (nt_total / 2) >= 256) {
    const int n = nrb / 2 * (nt_total / 2);
    dual_kernel<GemmOp<2, 2, KGW, MODE>, NoOp><<<n, GEMM_THREADS, 0, st>>>(g, 0, n);
  } else {
    const int n = nrb * nt_total;
    dual_kernel<GemmOp<1, 1, KGW, MODE>, NoOp><<<n, GEMM_THREADS, 0, st>>>(g, 0, n);
  }
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

}  // namespace

extern "C" int ttsx_encoder_lstm_create(const ttsx_encoder_lstm_weights* w, void* stream, ttsx_encoder_lstm** out) {
  TTSX_ARG(w && out, "ttsx_encoder_lstm_create: null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  ttsx_encoder_lstm* e = new ttsx_encoder_lstm();
  e->plan_scan = -1;
  ScopeExit on_fail([&] { (void)ttsx_encoder_lstm_destroy(e); });
  int rc = 0;
  for (int d = 0; d < 2; ++d) {
    TTSX_ARG(w->w_ih[d] && w->w_hh[d] && w->b_ih[d] && w->b_hh[d], "ttsx_encoder_lstm_create: null weight");
    // input projection carries no bias (MODE_PARTIAL); both biases are added by the step launch
    rc |= pack_layer(&e->w_ih[d], &e->b_unused[d], w->w_ih[d], ENC_IN, nullptr, 0, nullptr, nullptr, 4 * ENC_H, ENC_H, st);
    rc |= pack_layer(&e->w_hh[d], &e->bias[d], w->w_hh[d], ENC_H, nullptr, 0, w->b_ih[d], w->b_hh[d], 4 * ENC_H, ENC_H, st);
  }
  if (rc) return TTSX_ERR_HIP;
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&e->rc), sizeof(RunCtx)));
  TTSX_HIP(hipMemsetAsync(e->rc, 0, sizeof(RunCtx), st));
  TTSX_HIP(hipStreamSynchronize(st));
  on_fail.dismiss();
  *out = e;
  return TTSX_OK;
}

extern "C" int ttsx_encoder_lstm_destroy(ttsx_encoder_lstm* e) {
  if (!e) return TTSX_OK;
  enc_lstm_free_ws(e);
  for (int d = 0; d < 2; ++d) {
    float* bufs[] = {e->w_ih[d], e->b_unused[d], e->w_hh[d], e->bias[d]};
    for (float* b : bufs)
      if (b) (void)hipFree(b);
  }
  if (e->rc) (void)hipFree(e->rc);
  delete e;
  return TTSX_OK;
}

extern "C" int ttsx_encoder_lstm_set_plan(ttsx_encoder_lstm* e, int scan) {
  TTSX_ARG(e, "ttsx_encoder_lstm_set_plan: null handle");
  TTSX_ARG(scan >= -1 && scan <= 1, "ttsx_encoder_lstm_set_plan: the argument is -1 (by shape), 0 or 1");
  e->plan_scan = scan;
  return TTSX_OK;
}

extern "C" int ttsx_encoder_lstm_run(ttsx_encoder_lstm* e, const float* x, const int32_t* lengths, int B, int T,
                                     float* out, void* stream) {
  return ttsx_encoder_lstm_run_stateful(e, x, lengths, B, T, out, nullptr, nullptr, stream);
}

extern "C" int ttsx_encoder_lstm_run_stateful(ttsx_encoder_lstm* e, const float* x, const int32_t* lengths, int B,
                                              int T, float* out, float* h_state, float* c_state, void* stream) {
  TTSX_ARG(e && x && out, "ttsx_encoder_lstm_run: null argument");
  TTSX_ARG((h_state == nullptr) == (c_state == nullptr), "ttsx_encoder_lstm_run_stateful: give both states or neither");
  TTSX_ARG(!h_state || !lengths, "ttsx_encoder_lstm_run_stateful: a carried state needs full-length utterances");
  TTSX_ARG(B >= 1 && B <= 4096 && T >= 1, "ttsx_encoder_lstm_run: bad shape B=%d T=%d", B, T);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int Bc = B <= GEMV_MAX_B ? B : (B <= 32 ? round_up(B, 16) : round_up(B, 64));
  const int Ct = round_up(T * Bc, 64);
  if (e->cap_B != Bc || e->cap_T < T) {
    enc_lstm_free_ws(e);
    const size_t ct = Ct;
    int rc = dev_alloc(&e->xq, ENC_IN * ct);
    for (int d = 0; d < 2; ++d) {
      rc |= dev_alloc(&e->gi[d], 4 * ENC_H * ct);
      rc |= dev_alloc(&e->c[d], (size_t)ENC_H * Bc);
      for (int i = 0; i < 2; ++i) rc |= dev_alloc(&e->h[d][i], (size_t)ENC_H * Bc);
    }
    if (enc_scan_supports(Bc)) {
      rc |= dev_alloc(&e->scan_hx, enc_scan_hx_floats(Bc));
      rc |= dev_alloc(&e->scan_steps, enc_scan_step_words(Bc));
      rc |= dev_alloc(&e->scan_abort, 1);
    }
    if (rc) return TTSX_ERR_HIP;
    e->cap_B = Bc, e->cap_T = T;
  }
  e->Bc = Bc, e->Ct = Ct;
  {
    const size_t n = (size_t)Ct * (ENC_IN / 4);
    enc_to_quad_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(reinterpret_cast<float4*>(e->xq),
                                                                  reinterpret_cast<const float4*>(x), B, T, Bc, Ct);
    TTSX_LAUNCH_CHECK();
  }
  int rc;
  // The recurrence as one persistent launch (enc_scan.hip) or as a chain of T step launches.  Measured, encoder
  // in all (scan / chain, the chain's launches queued ahead of the GPU): B=1 T=100 0.46 / 0.50 ms, T=300 1.03 / 1.25,
  // T=1000 3.05 / 4.16; B=2 T=60 0.35 / 0.36; B=16 T=77 0.55 / 0.55; B=32 T=50 0.56 / 0.53; B=64 T=128 1.73 / 1.63:
  // a step of the scan is two memory-side round trips (counters, then h) plus the store acknowledgement, 2.7 us at
  // one utterance and 4.7 us at 64 (every workgroup reads all of h past the L2), a queued step launch 3.9 / 3.8 us.
  // By shape: the scan at batch 1-2.
  const bool by_shape = B <= GEMV_MAX_B && !e->scan_cool.cooling();
  const bool scan = e->scan_hx && enc_scan_supports(Bc) && (e->plan_scan == 1 || (e->plan_scan == -1 && by_shape));
  if (scan) {
    EncScanArgs a{};
    for (int d = 0; d < 2; ++d) {
      GemmP g{};
      g.Wp = e->w_ih[d], g.nseg = 1, g.KG = ENC_IN / 16, g.Bpad = Ct, g.kg0 = 0, g.kg_stride = ENC_IN / 16;
      g.rb0 = 0, g.nrb = ENC_H / 4, g.seg[0] = Seg{{e->xq, nullptr}, 0, ENC_IN / 16};
      g.gates = e->gi[d], g.ctx = e->rc;
      if ((rc = enc_launch_one<MODE_PARTIAL, ENC_IN / 16 / GEMM_WAVES>(g, ENC_H / 4, Bc, Ct, st))) return rc;
      a.w_hh[d] = e->w_hh[d], a.bias[d] = e->bias[d], a.gi[d] = e->gi[d];
    }
    a.hx = e->scan_hx, a.steps = e->scan_steps, a.abort_word = e->scan_abort;
    a.out = out, a.len = lengths, a.h_state = h_state, a.c_state = c_state, a.T = T, a.B = B, a.Bc = Bc, a.Ct = Ct;
    rc = enc_scan_run(a, st);
    if (rc != TTSX_ERR_STATE || e->plan_scan == 1) return rc;
    e->scan_cool.start();   // (h_state / c_state are only written by the last step of a scan that ran through)
  }
  for (int d = 0; d < 2; ++d) {
    TTSX_HIP(hipMemsetAsync(e->h[d][0], 0, sizeof(float) * ENC_H * Bc, st));
    TTSX_HIP(hipMemsetAsync(e->c[d], 0, sizeof(float) * ENC_H * Bc, st));
    if (h_state) {
      to_quad_kernel<<<ceil_div(B * ENC_H, 256), 256, 0, st>>>(e->h[d][0], h_state + (size_t)d * B * ENC_H, B, ENC_H, Bc);
      to_quad_kernel<<<ceil_div(B * ENC_H, 256), 256, 0, st>>>(e->c[d], c_state + (size_t)d * B * ENC_H, B, ENC_H, Bc);
      TTSX_LAUNCH_CHECK();
    }
    GemmP g{};
    g.Wp = e->w_ih[d], g.nseg = 1, g.KG = ENC_IN / 16, g.Bpad = Ct, g.kg0 = 0, g.kg_stride = ENC_IN / 16;
    g.rb0 = 0, g.nrb = ENC_H / 4, g.seg[0] = Seg{{e->xq, nullptr}, 0, ENC_IN / 16};
    g.gates = e->gi[d], g.ctx = e->rc;
    if ((rc = enc_launch_one<MODE_PARTIAL, ENC_IN / 16 / GEMM_WAVES>(g, ENC_H / 4, Bc, Ct, st))) return rc;
  }
  for (int s = 0; s < T; ++s) {
    GemmP g[2];
    for (int d = 0; d < 2; ++d) {
      const int t = d == 0 ? s : T - 1 - s;
      GemmP& q = g[d];
      q = GemmP{};
      q.Wp = e->w_hh[d], q.bias = e->bias[d], q.nseg = 1, q.KG = ENC_H / 16, q.Bpad = Bc, q.kg0 = 0;
      q.kg_stride = ENC_H / 16, q.rb0 = 0, q.nrb = ENC_H / 4, q.parity = s & 1;
      q.seg[0] = Seg{{e->h[d][0], e->h[d][1]}, 1, ENC_H / 16};
      q.gates = e->gi[d], q.gates_pitch = Ct, q.gates_col0 = t * Bc;
      q.out[0] = e->h[d][0], q.out[1] = e->h[d][1], q.c = e->c[d], q.ctx = e->rc;
      q.seq_out = out + (size_t)t * (2 * ENC_H) + d * ENC_H, q.seq_pitch = (size_t)T * 2 * ENC_H;
      q.len = lengths, q.t_index = t, q.n_valid = B;
    }
    // both directions in one launch
    constexpr int KGW = ENC_H / 16 / GEMM_WAVES;
    const int nrb = ENC_H / 4;
    if (Bc == 1) dual_kernel<GemvOp<1, KGW, MODE_LSTM>, GemvOp<1, KGW, MODE_LSTM>><<<2 * nrb, GEMM_THREADS, 0, st>>>(g[0], g[1], nrb);
    else if (Bc == 2) dual_kernel<GemvOp<2, KGW, MODE_LSTM>, GemvOp<2, KGW, MODE_LSTM>><<<2 * nrb, GEMM_THREADS, 0, st>>>(g[0], g[1], nrb);
    else {
      const int n = nrb * (Bc / 16);
      dual_kernel<GemmOp<1, 1, KGW, MODE_LSTM>, GemmOp<1, 1, KGW, MODE_LSTM>><<<2 * n, GEMM_THREADS, 0, st>>>(g[0], g[1], n);
    }
    TTSX_LAUNCH_CHECK();
  }
  if (h_state) {
    // step s writes the buffer of parity (s & 1) ^ 1, so the state after T steps sits in h[T & 1]
    for (int d = 0; d < 2; ++d) {
      from_quad_kernel<<<ceil_div(B * ENC_H, 256), 256, 0, st>>>(h_state + (size_t)d * B * ENC_H, e->h[d][T & 1], B, ENC_H, Bc);
      from_quad_kernel<<<ceil_div(B * ENC_H, 256), 256, 0, st>>>(c_state + (size_t)d * B * ENC_H, e->c[d], B, ENC_H, Bc);
    }
    TTSX_LAUNCH_CHECK();
  }
  return TTSX_OK;
}

// Tacotron2._add_speaker_embedding (Tacotron2/TTS/models.py:838-849): the speaker's embedding row is
// added to every encoder position.  Ids outside the table are clamped (the caller validates them).
namespace {
__global__ void add_speaker_kernel(float4* __restrict__ enc, const float4* __restrict__ table,
                                   const int32_t* __restrict__ ids, int n_speakers, int B, int T) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * T * (ENC_DIM / 4)) return;
  const int b = (int)(i / ((size_t)T * (ENC_DIM / 4))), c4 = (int)(i % (ENC_DIM / 4));
  const int id = min(max(ids[b], 0), n_speakers - 1);
  const float4 e = enc[i], s = table[(size_t)id * (ENC_DIM / 4) + c4];
  enc[i] = make_float4(e.x + s.x, e.y + s.y, e.z + s.z, e.w + s.w);
}
}  // namespace

extern "C" int ttsx_add_speaker_embedding(float* enc, const float* table, const int32_t* speaker_ids, int n_speakers,
                                          int B, int T, void* stream) {
  TTSX_ARG(enc && table && speaker_ids, "ttsx_add_speaker_embedding: null argument");
  TTSX_ARG(n_speakers >= 1 && B >= 1 && T >= 1, "ttsx_add_speaker_embedding: bad shape");
  const size_t n = (size_t)B * T * (ENC_DIM / 4);
  add_speaker_kernel<<<(unsigned)((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      reinterpret_cast<float4*>(enc), reinterpret_cast<const float4*>(table), speaker_ids, n_speakers, B, T);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

#ifdef TTSX_FA_TIMING
extern "C" int ttsx_debug_fa_timing(long long* out_host) {
  TTSX_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_fa_t), sizeof(long long) * 16));
  return TTSX_OK;
}
// grid > 0: record the workgroups of launches with that grid size and that many serial-path workgroups from now on; out_host: [2048][4] of the last one
extern "C" int ttsx_debug_wg_marks(int grid, int n_first, long long* out_host) {
  if (grid > 0) {
    TTSX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wg_grid), &grid, sizeof(int)));
    TTSX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wg_first), &n_first, sizeof(int)));
  }
  if (out_host) TTSX_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_wg_t), sizeof(long long) * 2048 * 4));
  return TTSX_OK;
}
#endif

// ====================================================================== Tacotron (v1) decoder steps
// Decoder.forward in eval mode (Tacotron/TTS/models.py:558-645) with AttentionRNNCell (:124-156),
// LocationSensitiveAttention (:71-88), Prenet (:250-253) and StopNet (:663-667) as a chain of eight
// launches per step on the row-block engine:
//   prenet-1 (+ the stop rule of the previous step as a side op) | prenet-2 | attention GRU (+ query
//   shares) | fused attention | GRU 1 (+ project_to_decoder_in as a side op) | GRU 2 (+ residuals) |
//   proj_to_mel + sigmoid (+ the memory-queue shift as a side op)
// project_to_decoder_in is linear, so GRU 1 takes its input weights composed with it and runs straight
// on [h_att; context; h] while d0 = Wd [h_att; context] + bd is produced next to it in the same launch;
// the residual d1 = h1 + d0 is never formed: GRU 2 multiplies its input weights with h1 and with d0 as two
// segments and adds both to its own output (d2 = h2 + h1 + d0).
// A GRU cell fits the engine's four-rows-per-hidden-unit layout as rows (r, z, n_input, n_hidden) over
// K = [x; h], the two n rows zero-padded over the half they do not see.  All biases of the energy
// (query, annotation, location-linear) are constants of the step and are added once (P carries the
// annotation bias, q_bias the other two).
#include "t1_steps.h"

namespace {

constexpr int T1_ENC = 256, T1_RNN = 256, T1_PRE0 = 256, T1_PRE1 = 128;

// rows (r, z, n_i, n_h) of every hidden unit over K = [x (kx); h (kh)], source row order
// g * H + u  (g = 0..3) so that pack_rows_kernel's unit interleave applies; biases likewise
__global__ void gru_expand_kernel(float* __restrict__ w4, float* __restrict__ b4, const float* __restrict__ w_ih,
                                  const float* __restrict__ w_hh, const float* __restrict__ b_ih,
                                  const float* __restrict__ b_hh, int H, int kx, int kh) {
  const int K = kx + kh;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)4 * H) {
    const int g = (int)(i / H), u = (int)(i % H);
    b4[i] = g == 0 ? b_ih[u] + b_hh[u] : g == 1 ? b_ih[H + u] + b_hh[H + u] : g == 2 ? b_ih[2 * H + u] : b_hh[2 * H + u];
  }
  if (i >= (size_t)4 * H * K) return;
  const int row = (int)(i / K), k = (int)(i % K), g = row / H, u = row % H;
  float v = 0.f;
  if (k < kx) {
    if (g <= 2) v = w_ih[(size_t)(g * H + u) * kx + k];
  } else {
    if (g <= 1) v = w_hh[(size_t)(g * H + u) * kh + (k - kx)];
    else if (g == 3) v = w_hh[(size_t)(2 * H + u) * kh + (k - kx)];
  }
  w4[i] = v;
}

// GRU input weights composed with a preceding bias-carrying linear layer x = Wd y + bd (no activation
// between the two): w_c[row][k] = sum_j w_ih[row][j] wd[j][k], b_c[row] = b_ih[row] + sum_j w_ih[row][j] bd[j]
// (fp64 accumulation, once at create).  One block per row.
__global__ void compose_gru_input_kernel(float* __restrict__ w_c, float* __restrict__ b_c, const float* __restrict__ w_ih,
                                         const float* __restrict__ b_ih, const float* __restrict__ wd,
                                         const float* __restrict__ bd, int kin, int ky) {
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
// [w | w]: the same input weights over two activation segments whose sum is the layer's input
__global__ void dup_cols_kernel(float* __restrict__ dst, const float* __restrict__ src, int rows, int k) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * 2 * k) return;
  const int row = (int)(i / (2 * k)), c = (int)(i % (2 * k));
  dst[i] = src[(size_t)row * k + (c % k)];
}

// the part of the memory queue that survives a step: quads [drop, drop + n) of the old queue -> [0, n)
struct MemShiftOp {
  struct P {
    const float4* src;
    float4* dst;
    int n;  // float4 elements
    RunCtx* ctx;
    int step_in_chunk;
  };
  static __device__ __forceinline__ void run(const P& p, int wg) {
    if (step_done(p.ctx, p.step_in_chunk)) return;
    for (int i = wg * GEMM_THREADS + threadIdx.x; i < p.n; i += 4 * GEMM_THREADS) p.dst[i] = p.src[i];
  }
};

// stop token and stop rule of step `idx` of the chunk (:624-640): stop = sigmoid(w . [d; out] + b);
// the utterance ends after step t (1-based) when t > T/4 and (stop > 0.6 or the attention on its last
// position > 0.6), or when t > max_decoder_steps.  One workgroup; a wave per utterance.
struct T1StopOp {
  struct P {
    const float* d2;      // quad [256]
    const float* out;     // quad [odim]
    const float* stop_w;  // [256 + odim]
    const float* stop_b;
    const float* a_prev;  // [B][T]
    const int32_t* len;
    RunCtx* ctx;
    int idx, Bpad, odim;
  };
  static __device__ __forceinline__ void run(const P& p, int) {
    RunCtx* cx = p.ctx;
    if (cx->all_done) return;   // (raised only by this op, which runs once per step: never under its own launch)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = wave; b < cx->B; b += GEMM_WAVES) {
      float acc = 0.f;
      for (int k = lane; k < T1_RNN + p.odim; k += 64)
        acc += p.stop_w[k] * (k < T1_RNN ? p.d2[qidx(k, b, p.Bpad)] : p.out[qidx(k - T1_RNN, b, p.Bpad)]);
      acc = wave_sum(acc);
      if (lane == 0 && cx->n_steps[b] == 0) {
        const float s = sigmoidf_(acc + p.stop_b[0]);
        cx->stop[(size_t)b * cx->stop_rows + cx->out_base + p.idx] = s;
        const int tn = cx->base + p.idx + 1, len = p.len[b];
        const bool end = ((float)tn > (float)len / 4.f && (s > 0.6f || p.a_prev[(size_t)b * cx->T + len - 1] > 0.6f)) ||
                         tn > cx->max_steps;
        if (end && cx->n_steps[b] == 0) {
          cx->n_steps[b] = tn;
          // this op rides along the first launch of step idx + 1, whose other workgroups are already running:
          // that step completes (its output rows are masked), the ones after it keep nothing
          if (atomicAdd(&cx->n_done, 1) + 1 == cx->B) {
            cx->done_at = cx->base + p.idx + 1, cx->all_done = 1;
            if (cx->host_words) __hip_atomic_store(cx->host_words, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
    }
  }
};

__global__ void t1_init_state_kernel(float* __restrict__ mem, float* __restrict__ h_att, float* __restrict__ h_d1,
                                     float* __restrict__ h_d2, const float* __restrict__ mem_init,
                                     const float* __restrict__ att_init, const float* __restrict__ rnn_inits,
                                     int mem_dim, int Bpad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < mem_dim * Bpad) mem[qidx(i / Bpad, i % Bpad, Bpad)] = mem_init[i / Bpad];
  if (i < T1_RNN * Bpad) {
    const int u = i / Bpad, b = i % Bpad;
    h_att[qidx(u, b, Bpad)] = att_init[u];
    h_d1[qidx(u, b, Bpad)] = rnn_inits[u];
    h_d2[qidx(u, b, Bpad)] = rnn_inits[T1_RNN + u];
  }
}

__global__ void t1_len_kernel(int32_t* __restrict__ dst, const int32_t* __restrict__ src, int B, int T) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) dst[b] = src ? min(max(src[b], 1), T) : T;
}

__global__ void t1_qbias_kernel(float* __restrict__ qb, const float* __restrict__ query_b,
                                const float* __restrict__ loc_lin_b) {
  const int i = threadIdx.x;
  if (i < ATT_DIM) qb[i] = query_b[i] + loc_lin_b[i];
}

}  // namespace

struct t1_steps {
  int mem_dim, odim, mem_k;  // mem_k: mem_dim rounded up to 128 (the engine's K granularity)
  float *w_pre0, *b_pre0, *w_pre1, *b_pre1, *w_att, *b_att, *w_pd, *b_pd, *w_g1, *b_g1, *w_g2, *b_g2, *w_mel, *b_mel;
  float *wq_t, *wlc_p, *wld_p, *q_bias, *v_w, *zero1, *stop_w, *stop_b;
  const float *mem_init, *att_init, *rnn_inits;  // borrowed from the owner of the handle
  // workspace
  int cap_B, cap_T, Bpad;
  float *mem[2], *x1, *x2, *h_att[2], *ctx, *d0, *h_d1[2], *d1, *h_d2[2], *d2, *out, *qpart, *a_prev, *a_cum, *enc_ws, *P_ws;
  int32_t* len_ws;
  RunCtx *rc, *rc_host;
  int* host_flag;
  hipGraphExec_t graph;
  hipStream_t cap_stream;
  int g_B, g_T, g_gen, ws_gen;
};

namespace {

template <int MODE, int KGW, class Side>
int t1_launch(const t1_steps* s, int B, GemmP g, int nrb, const typename Side::P& side, int n_side, hipStream_t st) {
  g.rb0 = 0, g.nrb = nrb;
  if (B == 1) {
    dual_kernel<GemvOp<1, KGW, MODE>, Side><<<nrb + n_side, GEMM_THREADS, 0, st>>>(g, side, nrb);
  } else if (B == 2) {
    dual_kernel<GemvOp<2, KGW, MODE>, Side><<<nrb + n_side, GEMM_THREADS, 0, st>>>(g, side, nrb);
  } else {
    const int n = nrb * (s->Bpad / 16);
    dual_kernel<GemmOp<1, 1, KGW, MODE>, Side><<<n + n_side, GEMM_THREADS, 0, st>>>(g, side, n);
  }
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

T1StopOp::P t1_stop_params(const t1_steps* s, int idx) {
  T1StopOp::P q{};
  q.d2 = s->d2, q.out = s->out, q.stop_w = s->stop_w, q.stop_b = s->stop_b, q.a_prev = s->a_prev, q.len = s->len_ws;
  q.ctx = s->rc, q.idx = idx, q.Bpad = s->Bpad, q.odim = s->odim;
  return q;
}

// One decoder step (index i of the chunk; `parity` = steps completed before it, & 1).
int t1_enqueue_step(t1_steps* s, int B, int T, int i, int parity, hipStream_t st) {
  float* const h1_new = parity ? s->h_d1[0] : s->h_d1[1];  // hidden state GRU 1 writes in this step
  const int Bpad = s->Bpad;
  int rc;
  GemmP g{};
  g.Bpad = Bpad, g.ctx = s->rc, g.step_in_chunk = i, g.parity = parity;
  // prenet layer 1 on the memory queue   (+ stop rule of the previous step)
  g.Wp = s->w_pre0, g.bias = s->b_pre0, g.nseg = 1, g.KG = s->mem_k / 16, g.kg0 = 0, g.kg_stride = g.KG;
  g.seg[0] = Seg{{s->mem[0], s->mem[1]}, 1, s->mem_k / 16};
  g.out[0] = s->x1, g.relu = 1;
  {
    const T1StopOp::P sp = t1_stop_params(s, i - 1);
    const int n_side = i > 0 ? 1 : 0;
    if (s->mem_k == 512) rc = t1_launch<MODE_LINEAR, 4, T1StopOp>(s, B, g, T1_PRE0 / 16, sp, n_side, st);
    else if (s->mem_k == 256) rc = t1_launch<MODE_LINEAR, 2, T1StopOp>(s, B, g, T1_PRE0 / 16, sp, n_side, st);
    else if (s->mem_k == 384) rc = t1_launch<MODE_LINEAR, 3, T1StopOp>(s, B, g, T1_PRE0 / 16, sp, n_side, st);
    else rc = t1_launch<MODE_LINEAR, 1, T1StopOp>(s, B, g, T1_PRE0 / 16, sp, n_side, st);
    if (rc) return rc;
  }
  // prenet layer 2
  g.Wp = s->w_pre1, g.bias = s->b_pre1, g.KG = T1_PRE0 / 16, g.kg_stride = g.KG;
  g.seg[0] = Seg{{s->x1, nullptr}, 0, T1_PRE0 / 16};
  g.out[0] = s->x2;
  if ((rc = t1_launch<MODE_LINEAR, 2, NoOp>(s, B, g, T1_PRE1 / 16, 0, 0, st))) return rc;
  // attention GRU on [prenet; context], leaving the shares of the query projection
  g.Wp = s->w_att, g.bias = s->b_att, g.nseg = 3, g.KG = (T1_PRE1 + T1_ENC + T1_RNN) / 16, g.kg_stride = g.KG;
  g.seg[0] = Seg{{s->x2, nullptr}, 0, T1_PRE1 / 16};
  g.seg[1] = Seg{{s->ctx, nullptr}, 0, T1_ENC / 16};
  g.seg[2] = Seg{{s->h_att[0], s->h_att[1]}, 1, T1_RNN / 16};
  g.out[0] = s->h_att[0], g.out[1] = s->h_att[1], g.relu = 0;
  g.wq_t = s->wq_t, g.qpart = s->qpart;
  if ((rc = t1_launch<MODE_GRU, 5, NoOp>(s, B, g, T1_RNN / 4, 0, 0, st))) return rc;
  g.wq_t = nullptr, g.qpart = nullptr;
  // attention
  {
    AttnP ap{};
    ap.P = s->P_ws, ap.v_w = s->v_w, ap.v_b = s->zero1, ap.a_prev = s->a_prev, ap.a_cum = s->a_cum;
    ap.ctx_out = s->ctx, ap.enc = s->enc_ws, ap.len = s->len_ws, ap.T = T, ap.ctx = s->rc, ap.step_in_chunk = i;
    ap.Bpad = Bpad, ap.norm = TTSX_NORM_SIGMOID, ap.q_bias = s->q_bias, ap.qpart = s->qpart;
    ap.n_qgroups = T1_RNN / 4, ap.wlc_p = s->wlc_p, ap.wld_p = s->wld_p;
    const size_t smem = sizeof(float) * FA_LDS_FLOATS;
    if (T <= 128) dual_kernel<FusedAttnOp<1, T1_ENC>, NoOp><<<B, GEMM_THREADS, smem, st>>>(ap, 0, B);
    else if (T <= 256) dual_kernel<FusedAttnOp<2, T1_ENC>, NoOp><<<B, GEMM_THREADS, smem, st>>>(ap, 0, B);
    else dual_kernel<FusedAttnOp<4, T1_ENC>, NoOp><<<B, GEMM_THREADS, smem, st>>>(ap, 0, B);
    TTSX_LAUNCH_CHECK();
  }
  // GRU 1 on [h_att; context; h1] with composed input weights   (+ d0 = project_to_decoder_in([h_att; context]))
  {
    GemmP gd = g;
    gd.Wp = s->w_pd, gd.bias = s->b_pd, gd.nseg = 2, gd.KG = (T1_RNN + T1_ENC) / 16, gd.kg_stride = gd.KG;
    gd.seg[0] = Seg{{s->h_att[0], s->h_att[1]}, 2, T1_RNN / 16};
    gd.seg[1] = Seg{{s->ctx, nullptr}, 0, T1_ENC / 16};
    gd.out[0] = s->d0, gd.out[1] = nullptr, gd.relu = 0, gd.rb0 = 0, gd.nrb = T1_RNN / 16;
    g.Wp = s->w_g1, g.bias = s->b_g1, g.nseg = 3, g.KG = (2 * T1_RNN + T1_ENC) / 16, g.kg_stride = g.KG;
    g.seg[0] = Seg{{s->h_att[0], s->h_att[1]}, 2, T1_RNN / 16};
    g.seg[1] = Seg{{s->ctx, nullptr}, 0, T1_ENC / 16};
    g.seg[2] = Seg{{s->h_d1[0], s->h_d1[1]}, 1, T1_RNN / 16};
    g.out[0] = s->h_d1[0], g.out[1] = s->h_d1[1], g.rb0 = 0, g.nrb = T1_RNN / 4;
    const int nrb = T1_RNN / 4, nside = T1_RNN / 16;
    if (B == 1) {
      dual_kernel<GemvOp<1, 6, MODE_GRU>, GemvOp<1, 4, MODE_LINEAR>><<<nrb + nside, GEMM_THREADS, 0, st>>>(g, gd, nrb);
    } else if (B == 2) {
      dual_kernel<GemvOp<2, 6, MODE_GRU>, GemvOp<2, 4, MODE_LINEAR>><<<nrb + nside, GEMM_THREADS, 0, st>>>(g, gd, nrb);
    } else {
      const int nt = Bpad / 16;
      dual_kernel<GemmOp<1, 1, 6, MODE_GRU>, GemmOp<1, 1, 4, MODE_LINEAR>><<<(nrb + nside) * nt, GEMM_THREADS, 0, st>>>(
          g, gd, nrb * nt);
    }
    TTSX_LAUNCH_CHECK();
  }
  // GRU 2 on [h1; d0; h2] (its input weights over both terms of d1 = h1 + d0); d2 = h2 + h1 + d0
  g.Wp = s->w_g2, g.bias = s->b_g2;
  g.seg[0] = Seg{{s->h_d1[0], s->h_d1[1]}, 2, T1_RNN / 16};
  g.seg[1] = Seg{{s->d0, nullptr}, 0, T1_RNN / 16};
  g.seg[2] = Seg{{s->h_d2[0], s->h_d2[1]}, 1, T1_RNN / 16};
  g.out[0] = s->h_d2[0], g.out[1] = s->h_d2[1], g.resid = h1_new, g.resid2 = s->d0, g.sum_out = s->d2;
  if ((rc = t1_launch<MODE_GRU, 6, NoOp>(s, B, g, T1_RNN / 4, 0, 0, st))) return rc;
  g.resid = nullptr, g.resid2 = nullptr, g.sum_out = nullptr;
  // mel frames   (+ the surviving part of the memory queue)
  g.Wp = s->w_mel, g.bias = s->b_mel, g.nseg = 1, g.KG = T1_RNN / 16, g.kg_stride = g.KG;
  g.seg[0] = Seg{{s->d2, nullptr}, 0, T1_RNN / 16};
  g.out[0] = s->out, g.out[1] = nullptr, g.o_quads = s->odim / 4;
  float* mem_old = s->mem[parity];
  float* mem_new = s->mem[parity ^ 1];
  const int keep_quads = (s->mem_dim - s->odim) / 4;
  g.mem_new = mem_new, g.mem_keep = keep_quads;
  MemShiftOp::P ms{};
  ms.src = reinterpret_cast<const float4*>(mem_old) + (size_t)(s->odim / 4) * Bpad;
  ms.dst = reinterpret_cast<float4*>(mem_new), ms.n = keep_quads * Bpad, ms.ctx = s->rc, ms.step_in_chunk = i;
  return t1_launch<MODE_MEL, 2, MemShiftOp>(s, B, g, ceil_div(s->odim, 16), ms, keep_quads > 0 ? 4 : 0, st);
}

int t1_enqueue_chunk(t1_steps* s, int B, int T, int n, hipStream_t st) {
  for (int i = 0; i < n; ++i) {
    int rc = t1_enqueue_step(s, B, T, i, i & 1, st);
    if (rc) return rc;
  }
  // the stop rule of the chunk's last step, then move the chunk base
  dual_kernel<NoOp, T1StopOp><<<1, GEMM_THREADS, 0, st>>>(0, t1_stop_params(s, n - 1), 0);
  TTSX_LAUNCH_CHECK();
  ctx_advance_kernel<<<1, 1, 0, st>>>(s->rc, n);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

void t1_free_ws(t1_steps* s) {
  float** bufs[] = {&s->mem[0], &s->mem[1], &s->x1, &s->x2, &s->h_att[0], &s->h_att[1], &s->ctx, &s->d0, &s->h_d1[0],
                    &s->h_d1[1], &s->d1, &s->h_d2[0], &s->h_d2[1], &s->d2, &s->out, &s->qpart, &s->a_prev, &s->a_cum,
                    &s->enc_ws, &s->P_ws};
  for (float** b : bufs) {
    if (*b) (void)hipFree(*b);
    *b = nullptr;
  }
  if (s->len_ws) (void)hipFree(s->len_ws);
  s->len_ws = nullptr;
  s->cap_B = s->cap_T = 0;
  if (s->graph) (void)hipGraphExecDestroy(s->graph);
  s->graph = nullptr;
}

int t1_pack_gru(float** wp, float** bp, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                int kx, hipStream_t st) {
  const int H = T1_RNN, K = kx + H;
  float *w4 = nullptr, *b4 = nullptr;
  ScopeExit free_tmp([&] { (void)hipFree(w4), (void)hipFree(b4); });
  if (dev_alloc(&w4, (size_t)4 * H * K) || dev_alloc(&b4, 4 * H)) return TTSX_ERR_HIP;
  const size_t n = (size_t)4 * H * K;
  gru_expand_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(w4, b4, w_ih, w_hh, b_ih, b_hh, H, kx, H);
  TTSX_LAUNCH_CHECK();
  int rc = pack_layer(wp, bp, w4, K, nullptr, 0, b4, nullptr, 4 * H, H, st);
  TTSX_HIP(hipStreamSynchronize(st));
  return rc;
}

}  // namespace

int t1_steps_create(const T1StepWeights* w, hipStream_t st, t1_steps** out) {
  TTSX_ARG(w && out, "t1_steps_create: null argument");
  TTSX_ARG(w->mem_dim % 4 == 0 && w->odim % 16 == 0 && w->odim <= w->mem_dim && w->mem_dim <= 512,
           "t1_steps_create: memory %d / output %d floats not supported", w->mem_dim, w->odim);
  t1_steps* s = new t1_steps();
  memset(s, 0, sizeof(*s));
  float *w_c = nullptr, *b_c = nullptr, *w_dup = nullptr;  // temporaries of the packing
  ScopeExit on_fail([&] { t1_steps_destroy(s); });
  ScopeExit free_tmp([&] { (void)hipFree(w_c), (void)hipFree(b_c), (void)hipFree(w_dup); });
  s->mem_dim = w->mem_dim, s->odim = w->odim, s->mem_k = round_up(w->mem_dim, 128);
  s->mem_init = w->mem_init, s->att_init = w->att_init, s->rnn_inits = w->rnn_inits;
  int rc = 0;
  // prenet layer 1: K padded with zero columns up to the engine's granularity
  rc |= pack_layer(&s->w_pre0, &s->b_pre0, w->pre_w0, w->mem_dim, nullptr, s->mem_k - w->mem_dim, w->pre_b0, nullptr,
                   T1_PRE0, 0, st);
  rc |= pack_layer(&s->w_pre1, &s->b_pre1, w->pre_w1, T1_PRE0, nullptr, 0, w->pre_b1, nullptr, T1_PRE1, 0, st);
  rc |= t1_pack_gru(&s->w_att, &s->b_att, w->att_w_ih, w->att_w_hh, w->att_b_ih, w->att_b_hh, T1_PRE1 + T1_ENC, st);
  rc |= pack_layer(&s->w_pd, &s->b_pd, w->pd_w, T1_RNN + T1_ENC, nullptr, 0, w->pd_b, nullptr, T1_RNN, 0, st);
  {
    // GRU 1: input weights composed with project_to_decoder_in; GRU 2: input weights over [h1 | d0]
    const int K = T1_RNN + T1_ENC;
    if (dev_alloc(&w_c, (size_t)3 * T1_RNN * K) || dev_alloc(&b_c, 3 * T1_RNN) ||
        dev_alloc(&w_dup, (size_t)3 * T1_RNN * 2 * T1_RNN))
      return TTSX_ERR_HIP;
    compose_gru_input_kernel<<<3 * T1_RNN, 256, 0, st>>>(w_c, b_c, w->rnn_w_ih[0], w->rnn_b_ih[0], w->pd_w, w->pd_b,
                                                         T1_RNN, K);
    TTSX_LAUNCH_CHECK();
    const size_t nd = (size_t)3 * T1_RNN * 2 * T1_RNN;
    dup_cols_kernel<<<(unsigned)((nd + 255) / 256), 256, 0, st>>>(w_dup, w->rnn_w_ih[1], 3 * T1_RNN, T1_RNN);
    TTSX_LAUNCH_CHECK();
    rc |= t1_pack_gru(&s->w_g1, &s->b_g1, w_c, w->rnn_w_hh[0], b_c, w->rnn_b_hh[0], K, st);
    rc |= t1_pack_gru(&s->w_g2, &s->b_g2, w_dup, w->rnn_w_hh[1], w->rnn_b_ih[1], w->rnn_b_hh[1], 2 * T1_RNN, st);
    TTSX_HIP(hipStreamSynchronize(st));
  }
  rc |= pack_layer(&s->w_mel, &s->b_mel, w->mel_w, T1_RNN, nullptr, 0, w->mel_b, nullptr, w->odim, 0, st);
  if (rc) return TTSX_ERR_HIP;
  if (dev_alloc(&s->wlc_p, 2 * 16 * 64) || dev_alloc(&s->wld_p, 8 * 8 * 64) || dev_alloc(&s->wq_t, (size_t)QUERY_DIM * ATT_DIM) ||
      dev_alloc(&s->q_bias, ATT_DIM) || dev_alloc(&s->zero1, 4))
    return TTSX_ERR_HIP;
  TTSX_HIP(hipMemsetAsync(s->wq_t, 0, sizeof(float) * QUERY_DIM * ATT_DIM, st));
  TTSX_HIP(hipMemsetAsync(s->zero1, 0, sizeof(float) * 4, st));
  pack_attn_frag_kernel<<<ceil_div(T1_RNN * ATT_DIM, 256), 256, 0, st>>>(s->wlc_p, s->wld_p, s->wq_t, w->loc_conv_w,
                                                                        w->loc_lin_w, w->query_w, T1_RNN);
  TTSX_LAUNCH_CHECK();
  t1_qbias_kernel<<<1, ATT_DIM, 0, st>>>(s->q_bias, w->query_b, w->loc_lin_b);
  TTSX_LAUNCH_CHECK();
  rc |= copy_dev(&s->v_w, w->v_w, ATT_DIM, st);
  rc |= copy_dev(&s->stop_w, w->stop_w, T1_RNN + w->odim, st);
  rc |= copy_dev(&s->stop_b, w->stop_b, 1, st);
  if (rc) return TTSX_ERR_HIP;
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&s->rc), sizeof(RunCtx)));
  TTSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->rc_host), sizeof(RunCtx), hipHostMallocDefault));
  TTSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->host_flag), 4 * sizeof(int), hipHostMallocDefault));
  TTSX_HIP(hipStreamCreateWithFlags(&s->cap_stream, hipStreamNonBlocking));
  {
    const int bytes = FA_LDS_FLOATS * (int)sizeof(float);
    TTSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<FusedAttnOp<1, T1_ENC>, NoOp>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    TTSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<FusedAttnOp<2, T1_ENC>, NoOp>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    TTSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dual_kernel<FusedAttnOp<4, T1_ENC>, NoOp>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  }
  TTSX_HIP(hipStreamSynchronize(st));
  on_fail.dismiss();
  *out = s;
  return TTSX_OK;
}

void t1_steps_destroy(t1_steps* s) {
  if (!s) return;
  t1_free_ws(s);
  float* bufs[] = {s->w_pre0, s->b_pre0, s->w_pre1, s->b_pre1, s->w_att, s->b_att, s->w_pd, s->b_pd, s->w_g1, s->b_g1,
                   s->w_g2, s->b_g2, s->w_mel, s->b_mel, s->wq_t, s->wlc_p, s->wld_p, s->q_bias, s->v_w, s->zero1,
                   s->stop_w, s->stop_b};
  for (float* b : bufs)
    if (b) (void)hipFree(b);
  if (s->cap_stream) (void)hipStreamDestroy(s->cap_stream);
  if (s->rc) (void)hipFree(s->rc);
  if (s->rc_host) (void)hipHostFree(s->rc_host);
  if (s->host_flag) (void)hipHostFree(s->host_flag);
  delete s;
}

int t1_steps_run(t1_steps* s, const float* enc, const float* annot, const int32_t* lengths, int B, int T,
                 int max_decoder_steps, float* mel, float* align, float* stop, int32_t* n_steps, int* steps_run_host,
                 hipStream_t st) {
  TTSX_ARG(s && enc && annot && mel && align && stop && n_steps, "t1_steps_run: null argument");
  TTSX_ARG(B >= 1 && B <= 4096 && T >= 1 && T <= FA_TMAX, "t1_steps_run: bad shape B=%d T=%d (T <= %d)", B, T, FA_TMAX);
  const int Bpad = B <= GEMV_MAX_B ? B : (B <= 32 ? round_up(B, 16) : round_up(B, 64));
  if (s->cap_B != Bpad || s->cap_T < T) {
    t1_free_ws(s);
    const size_t cb = Bpad, ct = T;
    int rc = 0;
    for (int i = 0; i < 2; ++i) {
      rc |= dev_alloc(&s->mem[i], (size_t)s->mem_k * cb);
      rc |= dev_alloc(&s->h_att[i], T1_RNN * cb);
      rc |= dev_alloc(&s->h_d1[i], T1_RNN * cb);
      rc |= dev_alloc(&s->h_d2[i], T1_RNN * cb);
    }
    rc |= dev_alloc(&s->x1, T1_PRE0 * cb);
    rc |= dev_alloc(&s->x2, T1_PRE1 * cb);
    rc |= dev_alloc(&s->ctx, T1_ENC * cb);
    rc |= dev_alloc(&s->d0, T1_RNN * cb);
    rc |= dev_alloc(&s->d1, T1_RNN * cb);
    rc |= dev_alloc(&s->d2, T1_RNN * cb);
    rc |= dev_alloc(&s->out, (size_t)round_up(s->odim, 16) * cb);
    rc |= dev_alloc(&s->qpart, (size_t)(T1_RNN / 4) * cb * ATT_DIM);
    rc |= dev_alloc(&s->a_prev, cb * ct);
    rc |= dev_alloc(&s->a_cum, cb * ct);
    rc |= dev_alloc(&s->enc_ws, cb * ct * T1_ENC);
    rc |= dev_alloc(&s->P_ws, cb * ct * ATT_DIM);
    rc |= dev_alloc(&s->len_ws, cb);
    if (rc) return TTSX_ERR_HIP;
    s->cap_B = Bpad, s->cap_T = T;
    s->ws_gen += 1;
  }
  s->Bpad = Bpad;
  const size_t cb = Bpad;
  // ---- initial state (Decoder._init_states, :538-556): learned rows, zero context and alignments
  for (int i = 0; i < 2; ++i) {
    TTSX_HIP(hipMemsetAsync(s->mem[i], 0, sizeof(float) * s->mem_k * cb, st));
    TTSX_HIP(hipMemsetAsync(s->h_att[i], 0, sizeof(float) * T1_RNN * cb, st));
    TTSX_HIP(hipMemsetAsync(s->h_d1[i], 0, sizeof(float) * T1_RNN * cb, st));
    TTSX_HIP(hipMemsetAsync(s->h_d2[i], 0, sizeof(float) * T1_RNN * cb, st));
  }
  float* zero[] = {s->x1, s->x2, s->ctx, s->d0, s->d1, s->d2};
  const size_t zn[] = {T1_PRE0 * cb, T1_PRE1 * cb, T1_ENC * cb, T1_RNN * cb, T1_RNN * cb, T1_RNN * cb};
  for (int i = 0; i < 6; ++i) TTSX_HIP(hipMemsetAsync(zero[i], 0, sizeof(float) * zn[i], st));
  TTSX_HIP(hipMemsetAsync(s->out, 0, sizeof(float) * (size_t)round_up(s->odim, 16) * cb, st));
  TTSX_HIP(hipMemsetAsync(s->a_prev, 0, sizeof(float) * (size_t)B * T, st));
  TTSX_HIP(hipMemsetAsync(s->a_cum, 0, sizeof(float) * (size_t)B * T, st));
  t1_init_state_kernel<<<ceil_div(s->mem_k * Bpad, 256), 256, 0, st>>>(s->mem[0], s->h_att[0], s->h_d1[0], s->h_d2[0],
                                                                     s->mem_init, s->att_init, s->rnn_inits,
                                                                     s->mem_dim, Bpad);
  TTSX_LAUNCH_CHECK();
  TTSX_HIP(hipMemcpyAsync(s->enc_ws, enc, sizeof(float) * (size_t)B * T * T1_ENC, hipMemcpyDeviceToDevice, st));
  TTSX_HIP(hipMemcpyAsync(s->P_ws, annot, sizeof(float) * (size_t)B * T * ATT_DIM, hipMemcpyDeviceToDevice, st));
  t1_len_kernel<<<ceil_div(B, 256), 256, 0, st>>>(s->len_ws, lengths, B, T);
  TTSX_LAUNCH_CHECK();
  TTSX_HIP(hipMemsetAsync(n_steps, 0, sizeof(int32_t) * B, st));
  RunCtx* h = s->rc_host;
  memset(h, 0, sizeof(*h));
  const int rows = max_decoder_steps + 1;
  h->B = B, h->T = T, h->r = s->odim / MEL_DIM, h->free_running = 1, h->max_steps = max_decoder_steps;
  h->mask_finished = 1, h->done_at = NEVER_DONE;
  h->mel = mel, h->mel_rows = rows * (s->odim / MEL_DIM), h->align = align, h->align_rows = rows;
  h->stop = stop, h->stop_rows = rows, h->n_steps = n_steps;
  TTSX_HIP(hipMemcpyAsync(s->rc, h, sizeof(RunCtx), hipMemcpyHostToDevice, st));
  TTSX_HIP(hipStreamSynchronize(st));
  // ---- chunks of CHUNK steps as one cached graph; the stop flag is polled once per chunk
  int launched = 0;
  while (launched < rows) {
    const int n = std::min(CHUNK, rows - launched);
    if (n == CHUNK) {
      const bool cached = s->graph && s->g_B == B && s->g_T == T && s->g_gen == s->ws_gen;
      if (!cached) {
        if (s->graph) (void)hipGraphExecDestroy(s->graph);
        s->graph = nullptr;
        hipGraph_t graph = nullptr;
        TTSX_HIP(hipStreamBeginCapture(s->cap_stream, hipStreamCaptureModeThreadLocal));
        int rc = t1_enqueue_chunk(s, B, T, n, s->cap_stream);
        hipError_t e = hipStreamEndCapture(s->cap_stream, &graph);
        if (rc) {
          if (graph) (void)hipGraphDestroy(graph);
          return rc;
        }
        TTSX_HIP(e);
        TTSX_HIP(hipGraphInstantiate(&s->graph, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        s->g_B = B, s->g_T = T, s->g_gen = s->ws_gen;
      }
      TTSX_HIP(hipGraphLaunch(s->graph, st));
    } else {
      int rc = t1_enqueue_chunk(s, B, T, n, st);
      if (rc) return rc;
    }
    launched += n;
    TTSX_HIP(hipMemcpyAsync(s->host_flag, &s->rc->all_done, sizeof(int), hipMemcpyDeviceToHost, st));
    TTSX_HIP(hipStreamSynchronize(st));
    if (s->host_flag[0]) break;
  }
  std::vector<int32_t> host(B);
  TTSX_HIP(hipMemcpyAsync(host.data(), n_steps, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipStreamSynchronize(st));
  int run = 0;
  for (int b = 0; b < B; ++b) run = std::max(run, (int)host[b]);
  if (steps_run_host) *steps_run_host = run;
  return TTSX_OK;
}
