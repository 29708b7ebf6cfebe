// Encoder BiLSTM recurrence as ONE persistent launch (Encoder.inference's nn.LSTM over the text,
// Tacotron2/TTS/models.py:449-454, 419-427): the T dependent steps of both directions, which as a chain of
// launches cost a launch edge, a weight fetch and a tail each (7 us per step at 64 utterances, 4 us at one).
//
// A workgroup owns one row block of one direction (16 gate rows = the four gates of four hidden units, rows
// interleaved 4 u + gate as everywhere in the library) and one group of up to 64 utterance columns, for all T steps:
// its 16 x 256 slice of W_hh sits in registers (two k-groups per wave), the cell state of its (unit, utterance)
// pairs in registers of the waves that finish the tiles.  Per step it waits until the 64 workgroups of its direction
// and column group have published h(t-1) (one step counter per workgroup, raised after the h stores have been
// acknowledged; every
// wave polls the eight producers of its own k-groups, so no workgroup-wide barrier stands before the loads), reads
// h(t-1) past the L2 (the producers sit on other XCDs), multiplies on the matrix cores, adds the input projection
// of the step (computed for all steps beforehand, prefetched before the wait) and the bias, updates the cell, and
// publishes its four units of h(t).  h travels through a ping-pong pair of buffers in the quad layout; a
// workgroup can only finish step s once every workgroup has finished step s - 1, so two buffers suffice.
// Columns past an utterance's length hold zero state and output (packed-sequence semantics, as the launch chain).
//
// All workgroups of a launch must be resident at once (at most 256, one or two per CU); every wait is bounded in
// wall-clock time and raises an abort word that the host reads - the caller then runs the launch chain.
#include "enc_scan.h"

namespace {

constexpr int H = 256, KG = H / 16, NRB = 4 * H / 16, WAVES = 8, THREADS = 512;
typedef unsigned long long u64;

struct ScanP {
  const float* w_hh[2];  // packed row blocks [64][16 k-groups][64 lanes][4]
  const float* bias[2];  // [1024] b_ih + b_hh, packed row order
  const float* gi[2];    // input projections, float4 [256 quads][Ct], column t * Bc + b
  float* hx[2];          // h exchange, float4 [2 parities][64 quads][Bx]
  unsigned* steps[2];    // [ncg][64] steps completed per workgroup
  float* out;            // [B][T][512]
  const int32_t* len;    // [B] or null
  float* h_state;        // [2][B][256] in/out or null
  float* c_state;
  int* abort_word;
  int T, B, Bc, Bx, Ct;
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int NT>
__global__ __launch_bounds__(THREADS) void enc_scan_kernel(ScanP p) {
  __shared__ f32x4 red[WAVES][NT][64];
  __shared__ int s_abort;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncg = p.Bx / (16 * NT);
  int wg = blockIdx.x;
  const int d = wg / (NRB * ncg);
  wg -= d * NRB * ncg;
  const int cg = wg / NRB, rb = wg % NRB;
  const int T = p.T, Bc = p.Bc, Bx = p.Bx;
  if (tid == 0) s_abort = 0;

  // weights of this wave's two k-groups (w, w + 8), bias of the quad this lane finishes
  const f32x4* W = reinterpret_cast<const f32x4*>(p.w_hh[d]) + ((size_t)rb * KG + wave) * 64 + lane;
  const f32x4 w0 = W[0], w1 = W[(size_t)WAVES * 64];
  const int quad = rb * 4 + (lane >> 4);                     // hidden unit of this lane's accumulator rows
  const int col = cg * 16 * NT + wave * 16 + (lane & 15);    // (waves < NT) utterance column this lane finishes
  const bool fin = wave < NT;
  const bool real = fin && col < p.B;
  const float4 bias = reinterpret_cast<const float4*>(p.bias[d])[quad];
  const int my_len = real ? (p.len ? p.len[col] : T) : 0;
  float c = 0.f;
  if (real && p.c_state) c = p.c_state[((size_t)d * p.B + col) * H + quad];
  unsigned* my_step = p.steps[d] + cg * NRB + rb;
  // the eight producers of this wave's k-groups: quads 4 w .. 4 w + 3 and 4 (w + 8) .. + 3 (a quad = a row block)
  const unsigned* watch = p.steps[d] + cg * NRB + (lane < 4 ? 4 * wave + lane : 4 * (wave + WAVES) + (lane - 4));
  const size_t xoff = (size_t)(lane >> 4) * Bx + cg * 16 * NT + (lane & 15);
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? s : T - 1 - s;
    // the input projection of this step: known since before the launch, requested before the wait
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fin) g = reinterpret_cast<const float4*>(p.gi[d])[(size_t)quad * p.Ct + (size_t)t * Bc + min(col, Bc - 1)];
    // ---- wait for h(t-1) of the producers this wave reads
    {
      long long t_wait = 0;
      for (long spin = 0;; ++spin) {
        const unsigned v = lane < 8 ? __hip_atomic_load(watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)s;
        if (__all((int)v >= s)) break;
        if ((spin & 63) == 63 && __hip_atomic_load(p.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          s_abort = 1;
          break;
        }
        if (team_wait_expired(spin, t_wait)) {
          __hip_atomic_store(p.abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_abort = 1;
          break;
        }
      }
    }
    // ---- h(t-1): B fragments of the two k-groups, read past the L2 (agent-scope loads, issued once the counters
    // have been seen.  No acquire fence and no release store: at agent scope they invalidate / write back the L2
    // on this part - measured 13 us per step with them against 3 without - and nothing here lives in a cache
    // that needs it: h and the counters only ever move through agent-scope accesses)
    const f32x4* X = reinterpret_cast<const f32x4*>(p.hx[d]) + (size_t)(s & 1) * (H / 4) * Bx + xoff;
    f32x4 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 xv[2][NT];
    // (16-byte loads with the agent-coherent cache policy; the wait is part of the statement because the compiler
    // does not count loads it has not issued itself)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int n = 0; n < NT; ++n)
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(xv[k][n]) : "v"(X + (size_t)(wave + k * WAVES) * 4 * Bx + n * 16) : "memory");
    if (NT == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(xv[0][0]), "+v"(xv[1][0])::"memory");
    if (NT == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(xv[0][0]), "+v"(xv[1][0]), "+v"(xv[0][NT - 1]), "+v"(xv[1][NT - 1])::"memory");
    if (NT == 4)
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(xv[0][0]), "+v"(xv[1][0]), "+v"(xv[0][1 % NT]), "+v"(xv[1][1 % NT]), "+v"(xv[0][2 % NT]), "+v"(xv[1][2 % NT]),
                     "+v"(xv[0][3 % NT]), "+v"(xv[1][3 % NT])::"memory");
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const f32x4 wv = k ? w1 : w0;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv[k][n].x, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv[k][n].y, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv[k][n].z, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv[k][n].w, acc[n], 0, 0, 0);
      }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) red[wave][n][lane] = acc[n];
    __syncthreads();
    if (s_abort) return;  // (uniform: read after the barrier)
    // ---- wave n finishes column tile n: gates i, f, g, o of (unit quad, utterance col)
    if (fin) {
      f32x4 v = red[0][wave][lane];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) v += red[w][wave][lane];
      const float gi_ = v[0] + bias.x + g.x, gf = v[1] + bias.y + g.y, gg = v[2] + bias.z + g.z, go = v[3] + bias.w + g.w;
      float c_new = sigm(gf) * c + sigm(gi_) * tanhf(gg);
      float h_new = sigm(go) * tanhf(c_new);
      if (t >= my_len) c_new = 0.f, h_new = 0.f;
      c = c_new;
      // h(t) of this lane's unit and column into the other buffer (64 lanes = 256 contiguous bytes), past the L2
      float* dst = p.hx[d] + ((size_t)((s + 1) & 1) * (H / 4) * Bx + (size_t)rb * Bx + cg * 16 * NT + wave * 16 + (lane & 15)) * 4 +
                   (lane >> 4);
      __hip_atomic_store(reinterpret_cast<unsigned*>(dst), __float_as_uint(h_new), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (real) p.out[((size_t)col * T + t) * (2 * H) + d * H + quad] = h_new;
      if (s == T - 1 && real && p.h_state) {
        p.h_state[((size_t)d * p.B + col) * H + quad] = h_new;
        p.c_state[((size_t)d * p.B + col) * H + quad] = c_new;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have left before the counter moves
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(my_step, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// h0 [2][B][256] (or zeros) into parity 0 of the exchange buffers, quad layout with Bx columns
__global__ void enc_scan_init_kernel(float* hx0, float* hx1, const float* h_state, int B, int Bx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * H * Bx) return;
  const int d = i / (H * Bx), r = i % (H * Bx);
  const int b = (r / 4) % Bx, k = (r / (4 * Bx)) * 4 + (r & 3);
  (d ? hx1 : hx0)[r] = (h_state && b < B) ? h_state[((size_t)d * B + b) * H + k] : 0.f;
}

}  // namespace

bool enc_scan_supports(int Bc) {
  const int Bx = Bc <= 16 ? 16 : Bc;
  return Bx == 16 || Bx == 32 || (Bx % 64 == 0 && 2 * NRB * (Bx / 64) <= 256);
}

int enc_scan_run(const EncScanArgs& a, hipStream_t st) {
  const int Bx = a.Bc <= 16 ? 16 : a.Bc;
  TTSX_ARG(enc_scan_supports(a.Bc), "enc_scan_run: %d columns outside what the persistent scan covers", a.Bc);
  const int NT = Bx == 16 ? 1 : Bx == 32 ? 2 : 4, ncg = Bx / (16 * NT);
  ScanP p{};
  for (int d = 0; d < 2; ++d) {
    p.w_hh[d] = a.w_hh[d], p.bias[d] = a.bias[d], p.gi[d] = a.gi[d];
    p.hx[d] = a.hx + (size_t)d * 2 * H * Bx;
    p.steps[d] = a.steps + (size_t)d * ncg * NRB;
  }
  p.out = a.out, p.len = a.len, p.h_state = a.h_state, p.c_state = a.c_state, p.abort_word = a.abort_word;
  p.T = a.T, p.B = a.B, p.Bc = a.Bc, p.Bx = Bx, p.Ct = a.Ct;
  TTSX_HIP(hipMemsetAsync(a.steps, 0, sizeof(unsigned) * 2 * ncg * NRB, st));
  TTSX_HIP(hipMemsetAsync(a.abort_word, ttsx_take_wait_failure(TTSX_WAIT_ENCODER_SCAN) ? 1 : 0, sizeof(int), st));
  enc_scan_init_kernel<<<ceil_div(2 * H * Bx, 256), 256, 0, st>>>(p.hx[0], p.hx[1], a.h_state, a.B, Bx);
  TTSX_LAUNCH_CHECK();
  const int grid = 2 * NRB * ncg;
  if (NT == 1) enc_scan_kernel<1><<<grid, THREADS, 0, st>>>(p);
  else if (NT == 2) enc_scan_kernel<2><<<grid, THREADS, 0, st>>>(p);
  else enc_scan_kernel<4><<<grid, THREADS, 0, st>>>(p);
  TTSX_LAUNCH_CHECK();
  int aborted = 0;
  TTSX_HIP(hipMemcpyAsync(&aborted, a.abort_word, sizeof(int), hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipStreamSynchronize(st));
  if (aborted) {
    ttsx_set_error("enc_scan_run: a workgroup of the BiLSTM scan waited too long for the others");
    return TTSX_ERR_STATE;
  }
  return TTSX_OK;
}

size_t enc_scan_hx_floats(int Bc) { return (size_t)2 * 2 * H * (Bc <= 16 ? 16 : Bc); }
size_t enc_scan_step_words(int Bc) { return (size_t)2 * NRB * ((Bc <= 16 ? 16 : Bc) / 16); }
