// Products of weight rows with an utterance's encoder outputs, for the persistent team kernels.
//
// A context vector is a weighted sum of encoder outputs, ctx = sum_p a[p] enc[p]; what a recurrent cell or a
// projection needs of it is W_c ctx = sum_p a[p] (W_c enc[p]).  With G[row][p] = W_c[row] . enc[p] formed once per
// utterance, a consumer takes the alignment weights themselves and never waits for the context.
#pragma once
#include "common.h"

namespace {

// G[b][row][p] = W[row][0..K) . enc[b][p][0..K) for `rows` rows (a multiple of 16; row stride ldw floats) and the TS
// positions the team is built for; zero from the utterance's own length on (len_p[b] clamped to [len_min, T]; no
// lengths: T).  Grid (ceil(rows / 64), TS / 64, utterances), 256 threads.  A wave forms 16 rows x 64 positions on
// v_mfma_f32_16x16x4_f32; lane l supplies row / position l & 15 and, for the four instructions of a 16-byte load,
// K index 16 i + 4 (l >> 4) + s on both sides.
template <int K>
__global__ __launch_bounds__(256) void ctx_products_kernel(const float* __restrict__ W, int rows, int ldw,
                                                            const float* __restrict__ enc, const int32_t* __restrict__ len_p,
                                                            int len_min, int T, int TS, float* __restrict__ G) {
  static_assert(K % 16 == 0, "whole 16-byte loads per lane group");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rb = blockIdx.x * 4 + wave, pb = blockIdx.y, b = blockIdx.z;
  if (16 * rb >= rows) return;
  const int r = lane & 15, kq = lane >> 4;
  const int len = len_p ? min(max(len_p[b], len_min), T) : T;
  enc += (size_t)b * T * K, G += (size_t)b * rows * TS;
  const f32x4* A = reinterpret_cast<const f32x4*>(W + (size_t)(16 * rb + r) * ldw) + kq;
  const f32x4* Bp[4];
  bool in[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int pos = 64 * pb + 16 * c + r;
    in[c] = pos < len;
    Bp[c] = reinterpret_cast<const f32x4*>(enc + (size_t)(in[c] ? pos : 0) * K) + kq;
  }
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int i = 0; i < K / 16; ++i) {
    const f32x4 a = A[4 * i];
    f32x4 bv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bv[c] = Bp[c][4 * i];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bv[c][s], acc[c], 0, 0, 0);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      G[(size_t)(16 * rb + 4 * kq + i) * TS + 64 * pb + 16 * c + r] = in[c] ? acc[c][i] : 0.f;
}

}  // namespace
