// Tacotron (v1) for gfx950: encoder prenet + CBHG, greedy attention decoder, post-CBHG and the
// linear-spectrogram projection.
//
// Follows Tacotron.forward in eval mode (Tacotron/TTS/models.py:692-700) with Encoder (:459-469),
// CBHG.forward (:412-434), BatchNormConv1d (:307-313), Highway (:331-334), Decoder.forward
// (:586-645), AttentionRNNCell.forward (:124-156), LocationSensitiveAttention.forward (:71-88),
// StopNet (:663-667) and the learned initial states (:538-556).
//
// Structure.
//   * Every convolution and every per-frame linear layer (conv banks, projections, highways, GRU
//     input projections, last_linear) is an implicit GEMM on the fp32 MFMA (conv_mfma.h), frame-major.
//   * The bidirectional GRU of a CBHG runs as a persistent scan: one workgroup per direction keeps
//     W_hh (384 x 128) in registers (64 floats per lane) and walks the time axis with two
//     workgroup barriers per step.
//   * The decoder is 1.66 M parameters and strictly serial (the stop rule is per step), so the whole
//     greedy loop of an utterance runs inside ONE workgroup, one launch, no host round trips:
//     weights stream from L2 / Infinity Cache each step, all state lives in LDS.
#include <algorithm>
#include <vector>

#include "conv_mfma.h"
#include "t1_steps.h"
#include "t1_team.h"

using convmfma::Layer;

namespace {

constexpr int GRU_H = 128, GRU_G = 384;

// ------------------------------------------------------------------ CBHG pieces
// max_pool1d(kernel 2, stride 1) after right-padding one zero: y[t] = max(x[t], t+1 < F ? x[t+1] : 0)
__global__ void maxpool2_kernel(float* __restrict__ y, const float* __restrict__ x, const int32_t* __restrict__ n_frames,
                                int B, int F, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)B * F * C;
  if (i >= n) return;
  const int t = (int)((i / C) % F);
  const int nf = n_frames ? min(max(n_frames[i / ((size_t)C * F)], 0), F) : F;
  const float nxt = t + 1 < nf ? x[i + C] : 0.f;
  y[i] = fmaxf(x[i], nxt);
}

// Highway: y = relu(h) * sigmoid(g) + x * (1 - sigmoid(g)); hg[.][0:128] = H pre-activation, [128:256] = T
__global__ void highway_kernel(float* __restrict__ y, const float* __restrict__ x, const float* __restrict__ hg,
                               size_t n_rows) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows * GRU_H) return;
  const size_t row = i / GRU_H;
  const int c = (int)(i % GRU_H);
  const float h = fmaxf(hg[row * 256 + c], 0.f);
  const float g = sigmoidf_(hg[row * 256 + 128 + c]);
  y[i] = h * g + x[i] * (1.f - g);
}

// Bidirectional GRU scan.  gi [B][F][768] = W_ih x + b_ih for both directions (forward gates
// r,z,n in 0..383, reverse in 384..767); w_hh [2][384][128]; b_hh [2][384]; out [B][F][256].
// Thread (row = tid>>1, half = tid&1) keeps W_hh[row][64*half .. +63] in registers.
__global__ __launch_bounds__(768) void gru_scan_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                       const float* __restrict__ b_hh, float* __restrict__ out,
                                                       const int32_t* __restrict__ n_frames, int Fpitch) {
  const int dir = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int F = n_frames ? min(max(n_frames[b], 0), Fpitch) : Fpitch;  // frames >= F are padding: zeros out
  const int row = tid >> 1, half = tid & 1;
  __shared__ float4 s_h[2][GRU_H / 4];
  __shared__ float s_gh[GRU_G];
  float w[64];
  {
    const float4* src = reinterpret_cast<const float4*>(w_hh + ((size_t)dir * GRU_G + row) * GRU_H + half * 64);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 v = src[i];
      w[4 * i] = v.x, w[4 * i + 1] = v.y, w[4 * i + 2] = v.z, w[4 * i + 3] = v.w;
    }
  }
  const float bh = b_hh[dir * GRU_G + row];
  if (tid < GRU_H / 4) s_h[0][tid] = make_float4(0.f, 0.f, 0.f, 0.f);
  float hprev = 0.f;
  const float* gib = gi + (size_t)b * Fpitch * 768 + dir * GRU_G;
  float* ob = out + (size_t)b * Fpitch * 256 + dir * GRU_H;
  for (int i = F * 256 + tid; i < Fpitch * 256; i += 768)
    if ((i & 255) < GRU_H) ob[i] = 0.f;
  float g_r = 0.f, g_z = 0.f, g_n = 0.f;
  if (tid < GRU_H && F > 0) {
    const int t0 = dir ? F - 1 : 0;
    g_r = gib[(size_t)t0 * 768 + tid], g_z = gib[(size_t)t0 * 768 + 128 + tid], g_n = gib[(size_t)t0 * 768 + 256 + tid];
  }
  __syncthreads();
  int cur = 0;
  for (int step = 0; step < F; ++step) {
    const int t = dir ? F - 1 - step : step;
    // input-gate terms of the next step, in flight during this step's matvec
    float n_r = 0.f, n_z = 0.f, n_n = 0.f;
    if (tid < GRU_H && step + 1 < F) {
      const int tn = dir ? t - 1 : t + 1;
      n_r = gib[(size_t)tn * 768 + tid], n_z = gib[(size_t)tn * 768 + 128 + tid], n_n = gib[(size_t)tn * 768 + 256 + tid];
    }
    const float4* h4 = s_h[cur] + half * 16;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 hv = h4[i];
      acc += w[4 * i] * hv.x + w[4 * i + 1] * hv.y + w[4 * i + 2] * hv.z + w[4 * i + 3] * hv.w;
    }
    acc += __shfl_xor(acc, 1, 64);
    if (half == 0) s_gh[row] = acc + bh;
    __syncthreads();
    if (tid < GRU_H) {
      const float r = sigmoidf_(g_r + s_gh[tid]);
      const float z = sigmoidf_(g_z + s_gh[128 + tid]);
      const float n = tanhf(g_n + r * s_gh[256 + tid]);
      const float hn = (1.f - z) * n + z * hprev;
      hprev = hn;
      reinterpret_cast<float*>(s_h[cur ^ 1])[tid] = hn;
      ob[(size_t)t * 256 + tid] = hn;
      g_r = n_r, g_z = n_z, g_n = n_n;
    }
    __syncthreads();
    cur ^= 1;
  }
}

struct Cbhg {
  int K, in, p0, p1;
  Layer bank[16], proj[2], pre_hw, hw[4], gru_in;
  bool has_pre_hw;
  float *w_hh, *b_hh;             // [2][384][128], [2][384]
  float *cat, *pool, *t0, *t1, *x0, *x1, *hg, *gi;  // workspace
  size_t cap_rows;
};

int stack2(float** dst, const float* a, const float* b, size_t na, size_t nb, hipStream_t st) {
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(dst), (na + nb) * sizeof(float)));
  TTSX_HIP(hipMemcpyAsync(*dst, a, na * sizeof(float), hipMemcpyDeviceToDevice, st));
  TTSX_HIP(hipMemcpyAsync(*dst + na, b, nb * sizeof(float), hipMemcpyDeviceToDevice, st));
  return TTSX_OK;
}

int cbhg_create(Cbhg* c, const ttsx_cbhg_weights* w, hipStream_t st) {
  memset(c, 0, sizeof(*c));
  TTSX_ARG(w->K >= 1 && w->K <= 16, "cbhg: K=%d", w->K);
  c->K = w->K, c->in = w->in_features, c->p0 = w->proj_dims[0], c->p1 = w->proj_dims[1];
  TTSX_ARG(c->p1 == c->in, "cbhg: last projection (%d) must match the input width (%d) for the residual", c->p1, c->in);
  int rc;
  for (int k = 1; k <= c->K; ++k)
    if ((rc = convmfma::layer_create(&c->bank[k - 1], w->bank_w[k - 1], nullptr, w->bank_bn_weight[k - 1],
                                     w->bank_bn_bias[k - 1], w->bank_bn_mean[k - 1], w->bank_bn_var[k - 1], 128, c->in,
                                     k, (k - 1) / 2, convmfma::ACT_RELU, 1e-3f, st)))
      return rc;
  const int pin[2] = {128 * c->K, c->p0}, pout[2] = {c->p0, c->p1};
  for (int i = 0; i < 2; ++i)
    if ((rc = convmfma::layer_create(&c->proj[i], w->proj_w[i], nullptr, w->proj_bn_weight[i], w->proj_bn_bias[i],
                                     w->proj_bn_mean[i], w->proj_bn_var[i], pout[i], pin[i], 3, 1,
                                     i == 0 ? convmfma::ACT_RELU : convmfma::ACT_NONE, 1e-3f, st)))
      return rc;
  c->has_pre_hw = w->pre_highway_w != nullptr;
  if (c->has_pre_hw) {
    if ((rc = convmfma::layer_create(&c->pre_hw, w->pre_highway_w, nullptr, nullptr, nullptr, nullptr, nullptr, 128,
                                     c->p1, 1, 0, convmfma::ACT_NONE, 0.f, st)))
      return rc;
  } else {
    TTSX_ARG(c->p1 == 128, "cbhg: no pre_highway but projection width %d != 128", c->p1);
  }
  std::vector<float*> tmp;  // concatenated source tensors, needed until the packing kernels have run
  ScopeExit free_tmp([&] {
    for (float* p : tmp) (void)hipFree(p);
  });
  for (int i = 0; i < 4; ++i) {
    float *wcat = nullptr, *bcat = nullptr;
    rc = stack2(&wcat, w->hw_H_w[i], w->hw_T_w[i], 128 * 128, 128 * 128, st);
    tmp.push_back(wcat);
    if (rc) return rc;
    rc = stack2(&bcat, w->hw_H_b[i], w->hw_T_b[i], 128, 128, st);
    tmp.push_back(bcat);
    if (rc) return rc;
    if ((rc = convmfma::layer_create(&c->hw[i], wcat, bcat, nullptr, nullptr, nullptr, nullptr, 256, 128, 1, 0,
                                     convmfma::ACT_NONE, 0.f, st)))
      return rc;
  }
  {
    float *wcat = nullptr, *bcat = nullptr;
    rc = stack2(&wcat, w->gru_w_ih[0], w->gru_w_ih[1], GRU_G * 128, GRU_G * 128, st);
    tmp.push_back(wcat);
    if (rc) return rc;
    rc = stack2(&bcat, w->gru_b_ih[0], w->gru_b_ih[1], GRU_G, GRU_G, st);
    tmp.push_back(bcat);
    if (rc) return rc;
    if ((rc = convmfma::layer_create(&c->gru_in, wcat, bcat, nullptr, nullptr, nullptr, nullptr, 768, 128, 1, 0,
                                     convmfma::ACT_NONE, 0.f, st)))
      return rc;
  }
  if ((rc = stack2(&c->w_hh, w->gru_w_hh[0], w->gru_w_hh[1], GRU_G * GRU_H, GRU_G * GRU_H, st))) return rc;
  if ((rc = stack2(&c->b_hh, w->gru_b_hh[0], w->gru_b_hh[1], GRU_G, GRU_G, st))) return rc;
  TTSX_HIP(hipStreamSynchronize(st));
  return TTSX_OK;
}

void cbhg_destroy(Cbhg* c) {
  for (int k = 0; k < c->K; ++k) convmfma::layer_destroy(&c->bank[k]);
  for (int i = 0; i < 2; ++i) convmfma::layer_destroy(&c->proj[i]);
  if (c->has_pre_hw) convmfma::layer_destroy(&c->pre_hw);
  for (int i = 0; i < 4; ++i) convmfma::layer_destroy(&c->hw[i]);
  convmfma::layer_destroy(&c->gru_in);
  float* bufs[] = {c->w_hh, c->b_hh, c->cat, c->pool, c->t0, c->t1, c->x0, c->x1, c->hg, c->gi};
  for (float* b : bufs)
    if (b) (void)hipFree(b);
}

int cbhg_ensure(Cbhg* c, size_t rows) {
  if (rows <= c->cap_rows) return TTSX_OK;
  float** bufs[] = {&c->cat, &c->pool, &c->t0, &c->t1, &c->x0, &c->x1, &c->hg, &c->gi};
  const size_t widths[] = {(size_t)128 * c->K, (size_t)128 * c->K, (size_t)c->p0, (size_t)c->p1, 128, 128, 256, 768};
  for (int i = 0; i < 8; ++i) {
    if (*bufs[i]) (void)hipFree(*bufs[i]);
    *bufs[i] = nullptr;
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(bufs[i]), rows * widths[i] * sizeof(float)));
  }
  c->cap_rows = rows;
  return TTSX_OK;
}

// x [B][F][in] -> out [B][F][256]
int cbhg_run(Cbhg* c, const float* x, const int32_t* nf, int B, int F, float* out, hipStream_t st) {
  const size_t rows = (size_t)B * F;
  int rc = cbhg_ensure(c, rows);
  if (rc) return rc;
  const int CK = 128 * c->K;
  for (int k = 0; k < c->K; ++k)
    if ((rc = convmfma::layer_run(c->bank[k], x, c->in, c->cat, CK, 128 * k, nullptr, 0, nf, B, F, st))) return rc;
  {
    const size_t n = rows * CK;
    maxpool2_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(c->pool, c->cat, nf, B, F, CK);
    TTSX_LAUNCH_CHECK();
  }
  if ((rc = convmfma::layer_run(c->proj[0], c->pool, CK, c->t0, c->p0, 0, nullptr, 0, nf, B, F, st))) return rc;
  if ((rc = convmfma::layer_run(c->proj[1], c->t0, c->p0, c->t1, c->p1, 0, x, c->in, nf, B, F, st))) return rc;
  const float* h = c->t1;
  if (c->has_pre_hw) {
    if ((rc = convmfma::layer_run(c->pre_hw, c->t1, c->p1, c->x0, 128, 0, nullptr, 0, nullptr, B, F, st))) return rc;
    h = c->x0;
  }
  float* ping[2] = {c->x1, c->x0};
  for (int i = 0; i < 4; ++i) {
    if ((rc = convmfma::layer_run(c->hw[i], h, 128, c->hg, 256, 0, nullptr, 0, nullptr, B, F, st))) return rc;
    float* y = ping[i & 1];
    if (y == h) y = ping[(i + 1) & 1];
    highway_kernel<<<(unsigned)((rows * GRU_H + 255) / 256), 256, 0, st>>>(y, h, c->hg, rows);
    TTSX_LAUNCH_CHECK();
    h = y;
  }
  if ((rc = convmfma::layer_run(c->gru_in, h, 128, c->gi, 768, 0, nullptr, 0, nullptr, B, F, st))) return rc;
  gru_scan_kernel<<<dim3(2, B), 768, 0, st>>>(c->gi, c->w_hh, c->b_hh, out, nf, F);
  TTSX_LAUNCH_CHECK();
  return TTSX_OK;
}

// ------------------------------------------------------------------ decoder
constexpr int DEC_THREADS = 1024, DEC_WAVES = DEC_THREADS / 64;
constexpr int ENC_D = 256, ATT_D = 128, LOC_F = 32, LOC_K = 31, PRE0 = 256, PRE1 = 128, RNN_D = 256;
constexpr int T1_MAX_T = 512;

struct DecP {
  const float *pre_w0, *pre_b0, *pre_w1, *pre_b1;
  const float *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh;
  const float *loc_conv_t, *loc_lin_w, *loc_lin_b, *query_w, *query_b, *v_w;
  const float *pd_w, *pd_b;
  const float *rnn_w_ih[2], *rnn_w_hh[2], *rnn_b_ih[2], *rnn_b_hh[2];
  const float *mel_w, *mel_b, *stop_w, *stop_b;
  const float *att_init, *mem_init, *rnn_inits;
  const float *enc, *annot;  // [B][T][256], [B][T][128]
  float *mel, *align, *stop;
  int32_t* n_steps;
  const int32_t* enc_len;  // [B] valid positions per utterance or null (= T)
  int T, max_steps, mem_dim, odim;
};

// out[r] = act(W[r][:] . x + bias[r]) for r < n_rows (n_rows % RP == 0, n_cols % 4 == 0); x and out in LDS.
// The weights come from L2 / Infinity Cache with microseconds of latency and a single workgroup
// has to cover it alone, so a wave takes RP rows at a time: RP independent 16-byte loads per lane
// in flight (RP = 16: 16 KiB per wave, 256 KiB per workgroup).
template <int ACT, int RP>
__device__ __forceinline__ void gemv_lds(const float* __restrict__ W, const float* __restrict__ bias,
                                         const float* x, int n_rows, int n_cols, float* out, int wave, int lane) {
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const int nc4 = n_cols >> 2;
  for (int r0 = wave * RP; r0 < n_rows; r0 += DEC_WAVES * RP) {
    float acc[RP];
#pragma unroll
    for (int i = 0; i < RP; ++i) acc[i] = 0.f;
    const float4* w0 = reinterpret_cast<const float4*>(W + (size_t)r0 * n_cols);
    for (int c = lane; c < nc4; c += 64) {
      float4 wv[RP];
#pragma unroll
      for (int i = 0; i < RP; ++i) wv[i] = w0[(size_t)i * nc4 + c];
      const float4 xv = x4[c];
#pragma unroll
      for (int i = 0; i < RP; ++i) acc[i] += wv[i].x * xv.x + wv[i].y * xv.y + wv[i].z * xv.z + wv[i].w * xv.w;
    }
    // butterfly: after the exchange every lane holds the full sum of each row
#pragma unroll
    for (int i = 0; i < RP; ++i) acc[i] = wave_sum(acc[i]);
    float mine = 0.f;
#pragma unroll
    for (int i = 0; i < RP; ++i)
      if (lane == i) mine = acc[i];
    if (lane < RP) {
      float v = mine + (bias ? bias[r0 + lane] : 0.f);
      if (ACT == 1) v = fmaxf(v, 0.f);
      else if (ACT == 2) v = sigmoidf_(v);
      out[r0 + lane] = v;
    }
  }
}

// h' = GRUCell(x, h): gi, gh are the two 768-vectors (biases included); 256 threads update h in place
__device__ __forceinline__ float gru_point(const float* gi, const float* gh, float h, int j) {
  const float r = sigmoidf_(gi[j] + gh[j]);
  const float z = sigmoidf_(gi[RNN_D + j] + gh[RNN_D + j]);
  const float n = tanhf(gi[2 * RNN_D + j] + r * gh[2 * RNN_D + j]);
  return (1.f - z) * n + z * h;
}

__global__ __launch_bounds__(DEC_THREADS) void t1_decoder_kernel(DecP p) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: weight row addresses stay in SGPRs
  const int Tp = p.T;                                                       // row pitch of enc / annot / align
  const int T = p.enc_len ? min(max(p.enc_len[b], 1), Tp) : Tp;             // positions attended
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // LDS map (floats); every vector that feeds a GEMV starts on a 16-byte boundary
  float* s_mem = sm;                          // [mem_dim <= 800]
  float* s_p1 = s_mem + 800;                  // [256]
  float* s_pc = s_p1 + PRE0;                  // [128 prenet ; 256 context]
  float* s_hc = s_pc + PRE1 + ENC_D;          // [256 h_att ; 256 context]
  float* s_gi = s_hc + RNN_D + ENC_D;         // [768]
  float* s_gh = s_gi + 3 * RNN_D;             // [768]
  float* s_d = s_gh + 3 * RNN_D;              // [256 d ; odim <= 400 out]
  float* s_h1 = s_d + RNN_D + 400;            // [256]
  float* s_h2 = s_h1 + RNN_D;                 // [256]
  float* s_q = s_h2 + RNN_D;                  // [128]
  float* s_wl = s_q + ATT_D;                  // [62][32] location conv weights, filter fastest
  float* s_wll = s_wl + 2 * LOC_K * LOC_F;    // [32][128] location linear weights, attention dim fastest
  float* s_part = s_wll + LOC_F * ATT_D;      // [4][256]
  float* s_red = s_part + 4 * ENC_D;          // [32]
  float* s_a = s_red + 32;                    // [T + 30]  previous alignment, zero padded by 15
  float* s_ac = s_a + (T + 2 * 15 + 2);       // [T + 30]  cumulative alignment, zero padded
  float* s_e = s_ac + (T + 2 * 15 + 2);       // [T]
  float* s_cf = s_e + ((T + 3) & ~3);         // [T][33]

  const float* enc = p.enc + (size_t)b * Tp * ENC_D;
  const float* annot = p.annot + (size_t)b * Tp * ATT_D;

  // ---- initial state (Decoder._init_states)
  for (int i = tid; i < p.mem_dim; i += DEC_THREADS) s_mem[i] = p.mem_init[i];
  for (int i = tid; i < RNN_D; i += DEC_THREADS) {
    s_hc[i] = p.att_init[i];
    s_hc[RNN_D + i] = 0.f;
    s_pc[PRE1 + i] = 0.f;
    s_h1[i] = p.rnn_inits[i];
    s_h2[i] = p.rnn_inits[RNN_D + i];
  }
  for (int i = tid; i < T + 32; i += DEC_THREADS) s_a[i] = 0.f, s_ac[i] = 0.f;
  for (int i = tid; i < 2 * LOC_K * LOC_F; i += DEC_THREADS) s_wl[i] = p.loc_conv_t[i];
  for (int i = tid; i < LOC_F * ATT_D; i += DEC_THREADS) s_wll[(i % LOC_F) * ATT_D + i / LOC_F] = p.loc_lin_w[i];
  float bll[2], vv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    bll[h] = p.loc_lin_b[lane + 64 * h];
    vv[h] = p.v_w[lane + 64 * h];
  }
  __syncthreads();

  int t = 0;
  for (;;) {
    // an opaque copy of the lane id per step: keeps the weight-address arithmetic of the nine
    // GEMVs inside the step instead of hoisted above the loop, where it would not fit in registers
    int ln = lane;
    asm volatile("" : "+v"(ln));
    // ---- prenet (dropout is identity in eval)
    gemv_lds<1, 16>(p.pre_w0, p.pre_b0, s_mem, PRE0, p.mem_dim, s_p1, wave, ln);
    __syncthreads();
    gemv_lds<1, 8>(p.pre_w1, p.pre_b1, s_p1, PRE1, PRE0, s_pc, wave, ln);
    __syncthreads();
    // ---- attention GRU on [prenet ; context]
    gemv_lds<0, 16>(p.att_w_ih, p.att_b_ih, s_pc, 3 * RNN_D, PRE1 + ENC_D, s_gi, wave, ln);
    gemv_lds<0, 16>(p.att_w_hh, p.att_b_hh, s_hc, 3 * RNN_D, RNN_D, s_gh, wave, ln);
    __syncthreads();
    if (tid < RNN_D) s_hc[tid] = gru_point(s_gi, s_gh, s_hc[tid], tid);
    __syncthreads();
    // ---- query, location features
    gemv_lds<0, 8>(p.query_w, p.query_b, s_hc, ATT_D, RNN_D, s_q, wave, ln);
    {
      const int f = tid & 31;
      for (int tt = tid >> 5; tt < T; tt += DEC_THREADS / 32) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < LOC_K; ++j) acc += s_wl[j * LOC_F + f] * s_a[tt + j];
#pragma unroll
        for (int j = 0; j < LOC_K; ++j) acc += s_wl[(LOC_K + j) * LOC_F + f] * s_ac[tt + j];
        s_cf[tt * (LOC_F + 1) + f] = acc;
      }
    }
    __syncthreads();
    // ---- energies: one wave per position, lane owns attention dims lane and lane + 64
    for (int tt = wave; tt < T; tt += DEC_WAVES) {
      float e = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int d = lane + 64 * h;
        float loc = bll[h];
#pragma unroll
        for (int f = 0; f < LOC_F; ++f) loc += s_wll[f * ATT_D + d] * s_cf[tt * (LOC_F + 1) + f];
        e += vv[h] * tanhf(s_q[d] + annot[(size_t)tt * ATT_D + d] + loc);
      }
      e = wave_sum(e);
      if (lane == 0) s_e[tt] = e;
    }
    __syncthreads();
    // ---- sigmoid normalisation, state update
    {
      float part = 0.f;
      for (int tt = tid; tt < T; tt += DEC_THREADS) {
        const float s = sigmoidf_(s_e[tt]);
        s_e[tt] = s;
        part += s;
      }
      part = wave_sum(part);
      if (lane == 0) s_red[wave] = part;
      __syncthreads();
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < DEC_WAVES; ++w) sum += s_red[w];
      for (int tt = tid; tt < T; tt += DEC_THREADS) {
        const float a = s_e[tt] / sum;
        s_a[15 + tt] = a;
        s_ac[15 + tt] += a;
        p.align[((size_t)b * (p.max_steps + 1) + t) * Tp + tt] = a;
      }
      for (int tt = T + tid; tt < Tp; tt += DEC_THREADS) p.align[((size_t)b * (p.max_steps + 1) + t) * Tp + tt] = 0.f;
    }
    __syncthreads();
    // ---- context = a . enc
    {
      const int c = tid & (ENC_D - 1), tg = tid >> 8;
      float acc = 0.f;
      for (int tt = tg; tt < T; tt += 4) acc += s_a[15 + tt] * enc[(size_t)tt * ENC_D + c];
      s_part[tg * ENC_D + c] = acc;
    }
    __syncthreads();
    if (tid < ENC_D) {
      const float c = (s_part[tid] + s_part[ENC_D + tid]) + (s_part[2 * ENC_D + tid] + s_part[3 * ENC_D + tid]);
      s_hc[RNN_D + tid] = c;
      s_pc[PRE1 + tid] = c;
    }
    __syncthreads();
    // ---- decoder input and the two residual GRUs
    gemv_lds<0, 16>(p.pd_w, p.pd_b, s_hc, RNN_D, RNN_D + ENC_D, s_d, wave, ln);
    __syncthreads();
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      float* s_h = l == 0 ? s_h1 : s_h2;
      gemv_lds<0, 16>(p.rnn_w_ih[l], p.rnn_b_ih[l], s_d, 3 * RNN_D, RNN_D, s_gi, wave, ln);
      gemv_lds<0, 16>(p.rnn_w_hh[l], p.rnn_b_hh[l], s_h, 3 * RNN_D, RNN_D, s_gh, wave, ln);
      __syncthreads();
      if (tid < RNN_D) {
        const float hn = gru_point(s_gi, s_gh, s_h[tid], tid);
        s_h[tid] = hn;
        s_d[tid] = hn + s_d[tid];
      }
      __syncthreads();
    }
    // ---- mel frames (sigmoid) and stop token
    gemv_lds<2, 16>(p.mel_w, p.mel_b, s_d, p.odim, RNN_D, s_d + RNN_D, wave, ln);
    __syncthreads();
    {
      float part = 0.f;
      for (int i = tid; i < RNN_D + p.odim; i += DEC_THREADS) part += p.stop_w[i] * s_d[i];
      part = wave_sum(part);
      if (lane == 0) s_red[wave] = part;
    }
    for (int i = tid; i < p.odim; i += DEC_THREADS)
      p.mel[((size_t)b * (p.max_steps + 1) + t) * p.odim + i] = s_d[RNN_D + i];
    __syncthreads();
    if (tid == 0) {
      float s = p.stop_b[0];
#pragma unroll
      for (int w = 0; w < DEC_WAVES; ++w) s += s_red[w];
      s = sigmoidf_(s);
      p.stop[(size_t)b * (p.max_steps + 1) + t] = s;
      const int tn = t + 1;
      int done = 0;
      if ((float)tn > (float)T / 4.f && (s > 0.6f || s_a[15 + T - 1] > 0.6f)) done = 1;
      else if (tn > p.max_steps) done = 1;
      s_red[31] = (float)done;
    }
    // memory queue: drop the oldest r frames, append the new ones (:594)
    __syncthreads();
    const bool done = s_red[31] != 0.f;
    t += 1;
    if (done) break;
    {
      // mem_dim <= 800 <= 1024 threads: every thread reads its element, then all write
      const int keep = p.mem_dim - p.odim;
      float v = 0.f;
      if (tid < p.mem_dim) v = tid < keep ? s_mem[tid + p.odim] : s_d[RNN_D + (tid - keep)];
      __syncthreads();
      if (tid < p.mem_dim) s_mem[tid] = v;
    }
    __syncthreads();
  }
  if (tid == 0) p.n_steps[b] = t;
}

size_t dec_smem_bytes(int T) {
  const size_t fixed = 800 + PRE0 + (PRE1 + ENC_D) + (RNN_D + ENC_D) + 3 * RNN_D * 2 + (RNN_D + 400) + RNN_D * 2 + ATT_D +
                       2 * LOC_K * LOC_F + LOC_F * ATT_D + 4 * ENC_D + 32;
  const size_t var = 2 * (size_t)(T + 32) + ((T + 3) & ~3) + (size_t)T * (LOC_F + 1);
  return (fixed + var) * sizeof(float);
}

__global__ void transpose_loc_conv_kernel(float* __restrict__ conv_t, const float* __restrict__ conv_w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < LOC_F * 2 * LOC_K) conv_t[(i % (2 * LOC_K)) * LOC_F + i / (2 * LOC_K)] = conv_w[i];
}

}  // namespace

struct ttsx_t1 {
  int r, memory_size, linear_dim;
  t1_steps* steps;  // decoder as a chain of launches over the whole chip (null: memory queue too long for it)
  t1_team* team;    // decoder as one persistent launch with on-chip weights, batch 1-2 (null: other geometry)
  int plan;         // ttsx_t1_set_plan: -1 by shape, 0 launch chain, 1 team kernel, 2 one workgroup per utterance
  TeamCoolDown team_cool;  // a team run timed out waiting for co-residency: the default plan uses the launch chain for a while
  Layer enc_pre[2], annot, last;
  Cbhg enc_cbhg, post_cbhg;
  // decoder weights (own copies, reference layouts)
  std::vector<float*> owned;
  DecP dec;
  float *ws_a, *ws_b, *ws_annot;  // encoder prenet ping-pong, annotations
  size_t cap_rows_enc, cap_rows_post, cap_annot;
  float* post_out;  // [rows][256]
};

namespace {
int own(ttsx_t1* m, const float** dst, const float* src, size_t n, hipStream_t st) {
  TTSX_ARG(src, "ttsx_t1_create: null weight pointer");
  float* p = nullptr;
  TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&p), n * sizeof(float)));
  TTSX_HIP(hipMemcpyAsync(p, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  m->owned.push_back(p);
  *dst = p;
  return TTSX_OK;
}
}  // namespace

extern "C" int ttsx_t1_create(const ttsx_t1_weights* w, int r, int memory_size, int linear_dim, void* stream,
                              ttsx_t1** out) {
  TTSX_ARG(w && out, "ttsx_t1_create: null argument");
  TTSX_ARG(r >= 1 && r <= 5 && memory_size >= r && memory_size <= 10, "ttsx_t1_create: r=%d memory_size=%d", r,
           memory_size);
  hipStream_t st = static_cast<hipStream_t>(stream);
  ttsx_t1* m = new ttsx_t1();
  ScopeExit on_fail([&] { (void)ttsx_t1_destroy(m); });
  m->steps = nullptr, m->team = nullptr, m->plan = -1;
  m->r = r, m->memory_size = memory_size, m->linear_dim = linear_dim;
  m->ws_a = m->ws_b = m->ws_annot = m->post_out = nullptr;
  m->cap_rows_enc = m->cap_rows_post = m->cap_annot = 0;
  memset(&m->dec, 0, sizeof(m->dec));
  int rc = 0;
  const int mem_dim = 80 * memory_size, odim = 80 * r;
  rc |= convmfma::layer_create(&m->enc_pre[0], w->enc_prenet_w[0], w->enc_prenet_b[0], nullptr, nullptr, nullptr, nullptr,
                               256, 256, 1, 0, convmfma::ACT_RELU, 0.f, st);
  rc |= convmfma::layer_create(&m->enc_pre[1], w->enc_prenet_w[1], w->enc_prenet_b[1], nullptr, nullptr, nullptr, nullptr,
                               128, 256, 1, 0, convmfma::ACT_RELU, 0.f, st);
  rc |= convmfma::layer_create(&m->annot, w->annot_w, w->annot_b, nullptr, nullptr, nullptr, nullptr, 128, 256, 1, 0,
                               convmfma::ACT_NONE, 0.f, st);
  rc |= convmfma::layer_create(&m->last, w->last_linear_w, w->last_linear_b, nullptr, nullptr, nullptr, nullptr,
                               linear_dim, 256, 1, 0, convmfma::ACT_SIGMOID, 0.f, st);
  if (rc) return rc;
  if ((rc = cbhg_create(&m->enc_cbhg, &w->enc_cbhg, st))) return rc;
  if ((rc = cbhg_create(&m->post_cbhg, &w->post_cbhg, st))) return rc;
  DecP& d = m->dec;
  d.mem_dim = mem_dim, d.odim = odim;
  rc |= own(m, &d.pre_w0, w->dec_prenet_w[0], (size_t)PRE0 * mem_dim, st);
  rc |= own(m, &d.pre_b0, w->dec_prenet_b[0], PRE0, st);
  rc |= own(m, &d.pre_w1, w->dec_prenet_w[1], PRE1 * PRE0, st);
  rc |= own(m, &d.pre_b1, w->dec_prenet_b[1], PRE1, st);
  rc |= own(m, &d.att_w_ih, w->att_rnn_w_ih, 3 * RNN_D * (PRE1 + ENC_D), st);
  rc |= own(m, &d.att_w_hh, w->att_rnn_w_hh, 3 * RNN_D * RNN_D, st);
  rc |= own(m, &d.att_b_ih, w->att_rnn_b_ih, 3 * RNN_D, st);
  rc |= own(m, &d.att_b_hh, w->att_rnn_b_hh, 3 * RNN_D, st);
  rc |= own(m, &d.loc_lin_w, w->loc_linear_w, ATT_D * LOC_F, st);
  rc |= own(m, &d.loc_lin_b, w->loc_linear_b, ATT_D, st);
  rc |= own(m, &d.query_w, w->query_w, ATT_D * RNN_D, st);
  rc |= own(m, &d.query_b, w->query_b, ATT_D, st);
  rc |= own(m, &d.v_w, w->v_w, ATT_D, st);
  rc |= own(m, &d.pd_w, w->proj_dec_in_w, RNN_D * (RNN_D + ENC_D), st);
  rc |= own(m, &d.pd_b, w->proj_dec_in_b, RNN_D, st);
  for (int l = 0; l < 2; ++l) {
    rc |= own(m, &d.rnn_w_ih[l], w->dec_rnn_w_ih[l], 3 * RNN_D * RNN_D, st);
    rc |= own(m, &d.rnn_w_hh[l], w->dec_rnn_w_hh[l], 3 * RNN_D * RNN_D, st);
    rc |= own(m, &d.rnn_b_ih[l], w->dec_rnn_b_ih[l], 3 * RNN_D, st);
    rc |= own(m, &d.rnn_b_hh[l], w->dec_rnn_b_hh[l], 3 * RNN_D, st);
  }
  rc |= own(m, &d.mel_w, w->proj_mel_w, (size_t)odim * RNN_D, st);
  rc |= own(m, &d.mel_b, w->proj_mel_b, odim, st);
  rc |= own(m, &d.stop_w, w->stop_w, RNN_D + odim, st);
  rc |= own(m, &d.stop_b, w->stop_b, 1, st);
  rc |= own(m, &d.att_init, w->att_rnn_init, RNN_D, st);
  rc |= own(m, &d.mem_init, w->memory_init, mem_dim, st);
  rc |= own(m, &d.rnn_inits, w->dec_rnn_inits, 2 * RNN_D, st);
  if (rc) return rc;
  {
    TTSX_ARG(w->loc_conv_w, "ttsx_t1_create: null weight pointer");
    float* ct = nullptr;
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&ct), LOC_F * 2 * LOC_K * sizeof(float)));
    transpose_loc_conv_kernel<<<ceil_div(LOC_F * 2 * LOC_K, 256), 256, 0, st>>>(ct, w->loc_conv_w);
    TTSX_LAUNCH_CHECK();
    m->owned.push_back(ct);
    d.loc_conv_t = ct;
  }
  m->steps = nullptr;
  if (mem_dim <= 512) {
    T1StepWeights sw{};
    sw.pre_w0 = d.pre_w0, sw.pre_b0 = d.pre_b0, sw.pre_w1 = d.pre_w1, sw.pre_b1 = d.pre_b1;
    sw.att_w_ih = d.att_w_ih, sw.att_w_hh = d.att_w_hh, sw.att_b_ih = d.att_b_ih, sw.att_b_hh = d.att_b_hh;
    sw.loc_conv_w = w->loc_conv_w, sw.loc_lin_w = d.loc_lin_w, sw.loc_lin_b = d.loc_lin_b;
    sw.query_w = d.query_w, sw.query_b = d.query_b, sw.v_w = d.v_w, sw.pd_w = d.pd_w, sw.pd_b = d.pd_b;
    for (int l = 0; l < 2; ++l) {
      sw.rnn_w_ih[l] = d.rnn_w_ih[l], sw.rnn_w_hh[l] = d.rnn_w_hh[l];
      sw.rnn_b_ih[l] = d.rnn_b_ih[l], sw.rnn_b_hh[l] = d.rnn_b_hh[l];
    }
    sw.mel_w = d.mel_w, sw.mel_b = d.mel_b, sw.stop_w = d.stop_w, sw.stop_b = d.stop_b;
    sw.att_init = d.att_init, sw.mem_init = d.mem_init, sw.rnn_inits = d.rnn_inits;
    sw.mem_dim = mem_dim, sw.odim = odim;
    if ((rc = t1_steps_create(&sw, st, &m->steps))) return rc;
    if ((rc = t1_team_create(&sw, st, &m->team))) return rc;
  }
  TTSX_HIP(hipStreamSynchronize(st));
  on_fail.dismiss();
  *out = m;
  return TTSX_OK;
}

extern "C" int ttsx_t1_destroy(ttsx_t1* m) {
  if (!m) return TTSX_OK;
  for (int i = 0; i < 2; ++i) convmfma::layer_destroy(&m->enc_pre[i]);
  convmfma::layer_destroy(&m->annot);
  convmfma::layer_destroy(&m->last);
  cbhg_destroy(&m->enc_cbhg);
  cbhg_destroy(&m->post_cbhg);
  t1_steps_destroy(m->steps);
  t1_team_destroy(m->team);
  for (float* p : m->owned) (void)hipFree(p);
  float* bufs[] = {m->ws_a, m->ws_b, m->ws_annot, m->post_out};
  for (float* b : bufs)
    if (b) (void)hipFree(b);
  delete m;
  return TTSX_OK;
}

extern "C" int ttsx_t1_encoder_run(ttsx_t1* m, const float* emb, const int32_t* lengths, int B, int T, float* enc,
                                   void* stream) {
  TTSX_ARG(m && emb && enc, "ttsx_t1_encoder_run: null argument");
  TTSX_ARG(B >= 1 && T >= 1, "ttsx_t1_encoder_run: bad shape B=%d T=%d", B, T);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t rows = (size_t)B * T;
  if (rows > m->cap_rows_enc) {
    if (m->ws_a) (void)hipFree(m->ws_a);
    if (m->ws_b) (void)hipFree(m->ws_b);
    m->ws_a = m->ws_b = nullptr;
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&m->ws_a), rows * 256 * sizeof(float)));
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&m->ws_b), rows * 128 * sizeof(float)));
    m->cap_rows_enc = rows;
  }
  int rc;
  if ((rc = convmfma::layer_run(m->enc_pre[0], emb, 256, m->ws_a, 256, 0, nullptr, 0, nullptr, B, T, st))) return rc;
  if ((rc = convmfma::layer_run(m->enc_pre[1], m->ws_a, 256, m->ws_b, 128, 0, nullptr, 0, nullptr, B, T, st))) return rc;
  return cbhg_run(&m->enc_cbhg, m->ws_b, lengths, B, T, enc, st);
}

extern "C" int ttsx_t1_decoder_run(ttsx_t1* m, const float* enc, const int32_t* lengths, int B, int T,
                                   int max_decoder_steps, float* mel, float* align, float* stop, int32_t* n_steps,
                                   int* steps_run_host, void* stream) {
  TTSX_ARG(m && enc && mel && align && stop && n_steps, "ttsx_t1_decoder_run: null argument");
  TTSX_ARG(B >= 1 && T >= 1 && T <= T1_MAX_T, "ttsx_t1_decoder_run: bad shape B=%d T=%d (T <= %d)", B, T, T1_MAX_T);
  TTSX_ARG(max_decoder_steps >= 0, "ttsx_t1_decoder_run: max_decoder_steps=%d", max_decoder_steps);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t rows = (size_t)B * T;
  if (rows > m->cap_annot) {
    if (m->ws_annot) (void)hipFree(m->ws_annot);
    m->ws_annot = nullptr;
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&m->ws_annot), rows * ATT_D * sizeof(float)));
    m->cap_annot = rows;
  }
  // processed annotations, once per utterance (LocationSensitiveAttention caches them, :82-83)
  int rc = convmfma::layer_run(m->annot, enc, ENC_D, m->ws_annot, ATT_D, 0, nullptr, 0, nullptr, B, T, st);
  if (rc) return rc;
  // Batch 1-2 with a text of up to 128 symbols: one persistent launch of a 36-workgroup team per utterance with
  // all decoder weights on chip (t1_team.hip).  Otherwise seven launches per step over the whole chip
  // (t1_steps, in t2_decoder.hip); a memory queue longer than 512 floats leaves only the
  // one-workgroup-per-utterance kernel below.  ttsx_t1_set_plan overrides the choice where the shapes allow.
  const bool team_ok = t1_team_supports(m->team, B, T);
  if (team_ok && (m->plan == 1 || (m->plan == -1 && !m->team_cool.cooling()))) {
    rc = t1_team_run(m->team, enc, m->ws_annot, lengths, B, T, max_decoder_steps, mel, align, stop, n_steps,
                     steps_run_host, st);
    // TTSX_ERR_STATE: the team did not become resident together (other kernels held the CUs) and its bounded waits
    // gave up.  Under the default plan the launch chain rewrites every output, and the handle tries the team again
    // after a cool-down.
    if (rc != TTSX_ERR_STATE || m->plan == 1 || !m->steps) return rc;
    m->team_cool.start();
  }
  if (m->steps && m->plan != 2)
    return t1_steps_run(m->steps, enc, m->ws_annot, lengths, B, T, max_decoder_steps, mel, align, stop, n_steps,
                        steps_run_host, st);
  DecP d = m->dec;
  d.enc = enc, d.annot = m->ws_annot, d.mel = mel, d.align = align, d.stop = stop, d.n_steps = n_steps;
  d.T = T, d.max_steps = max_decoder_steps, d.enc_len = lengths;
  const size_t smem = dec_smem_bytes(T);
  TTSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(t1_decoder_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  t1_decoder_kernel<<<B, DEC_THREADS, smem, st>>>(d);
  TTSX_LAUNCH_CHECK();
  std::vector<int32_t> host(B);
  TTSX_HIP(hipMemcpyAsync(host.data(), n_steps, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipStreamSynchronize(st));
  int run = 0;
  for (int b = 0; b < B; ++b) run = std::max(run, (int)host[b]);
  if (steps_run_host) *steps_run_host = run;
  return TTSX_OK;
}

extern "C" int ttsx_t1_set_plan(ttsx_t1* m, int decoder) {
  TTSX_ARG(m, "ttsx_t1_set_plan: null handle");
  TTSX_ARG(decoder >= -1 && decoder <= 2, "ttsx_t1_set_plan: decoder=%d is not -1, 0, 1 or 2", decoder);
  m->plan = decoder;
  return TTSX_OK;
}

extern "C" int ttsx_t1_postnet_run(ttsx_t1* m, const float* mel, const int32_t* n_frames, int B, int F, float* linear,
                                   void* stream) {
  TTSX_ARG(m && mel && linear, "ttsx_t1_postnet_run: null argument");
  TTSX_ARG(B >= 1 && F >= 1, "ttsx_t1_postnet_run: bad shape B=%d F=%d", B, F);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t rows = (size_t)B * F;
  if (rows > m->cap_rows_post) {
    if (m->post_out) (void)hipFree(m->post_out);
    m->post_out = nullptr;
    TTSX_HIP(hipMalloc(reinterpret_cast<void**>(&m->post_out), rows * 256 * sizeof(float)));
    m->cap_rows_post = rows;
  }
  int rc = cbhg_run(&m->post_cbhg, mel, n_frames, B, F, m->post_out, st);
  if (rc) return rc;
  return convmfma::layer_run(m->last, m->post_out, 256, linear, m->linear_dim, 0, nullptr, 0, nullptr, B, F, st);
}
