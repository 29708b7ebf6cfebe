// Probe for the K-part GEMM of the batched decoder: out[part][4096 x 64] = W[4096 x Kpart] . X[Kpart x 64]
// with the activation slice of a workgroup staged ONCE in LDS and every wave streaming its own weight rows
// straight into registers.  Variants differ in the rows per workgroup.  Prints us per launch for the decoder
// LSTM's early columns (K = 2048) and checks one tile against the host.
//   hipcc -O3 --offload-arch=gfx950 tools/part_probe.hip -o tools/part_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 4096, BPAD = 64;

// weights packed as in the decoder: [rb][kg][lane][4], lane l: row l&15, k = 4*(l>>4)+s
// X quad-interleaved: float4 index (k/4)*BPAD + b

// NW waves per workgroup, one row block (16 rows) x 4 column tiles per wave; KP = K per part.
template <int NW, int KP>
__global__ __launch_bounds__(NW * 64) void part_kernel(const float* __restrict__ Wp, int kg_stride, int kg0,
                                                       const float* __restrict__ X, float* __restrict__ out, int nparts) {
  extern __shared__ float4 s_x[];  // [KP/4 quads][64 cols]
  constexpr int NT = 4, KGP = KP / 16, NTHR = NW * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ngrp = ROWS / 16 / NW;
  const int part = (blockIdx.x / ngrp) % nparts, grp = blockIdx.x % ngrp;
  const int rb = grp * NW + wave;
  const float4* W = reinterpret_cast<const float4*>(Wp) + ((size_t)rb * kg_stride + kg0 + part * KGP) * 64 + lane;
  float4 wv[KGP];
#pragma unroll
  for (int i = 0; i < KGP; ++i) wv[i] = W[(size_t)i * 64];
  const float4* X4 = reinterpret_cast<const float4*>(X) + (size_t)part * (KP / 4) * BPAD;
  constexpr int NX = (KP / 4) * BPAD / NTHR;
#pragma unroll
  for (int i = 0; i < NX; ++i) s_x[threadIdx.x + NTHR * i] = X4[threadIdx.x + NTHR * i];
  __syncthreads();
  f32x4 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float4* xs = s_x + (lane >> 4) * BPAD + (lane & 15);
#pragma unroll
  for (int i = 0; i < KGP; ++i) {
    float4 xv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) xv[n] = xs[i * 4 * BPAD + n * 16];
#define SLICE(C) _Pragma("unroll") for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i].C, xv[n].C, acc[n], 0, 0, 0);
    SLICE(x) SLICE(y) SLICE(z) SLICE(w)
#undef SLICE
  }
  float4* O = reinterpret_cast<float4*>(out) + (size_t)part * (ROWS / 4) * BPAD + (size_t)(rb * 4 + (lane >> 4)) * BPAD +
              (lane & 15);
#pragma unroll
  for (int n = 0; n < NT; ++n) O[n * 16] = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
}

template <int NW, int KP>
void run(const float* dW, const float* dX, float* dO, int K, const std::vector<float>& hW, const std::vector<float>& hX) {
  const int KG = K / 16, nparts = K / KP;
  const size_t nO = (size_t)nparts * ROWS * BPAD, nW = (size_t)ROWS * K;
  const int smem = (KP / 4) * BPAD * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(part_kernel<NW, KP>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  const int grid = ROWS / 16 / NW * nparts;
  float ms = 0;
  float t[3];
  for (int mi = 0; mi < 3; ++mi) {
    const int M = 1 << mi;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(a);
      for (int i = 0; i < 200; ++i) part_kernel<NW, KP><<<grid * M, NW * 64, smem>>>(dW, KG, 0, dX, dO, nparts);
      hipEventRecord(b);
      hipEventSynchronize(b);
      hipEventElapsedTime(&ms, a, b);
    }
    t[mi] = ms * 5.f;
  }
  printf("waves %d  Kpart %3d  grid %4d  lds %3d KB  %.2f / %.2f / %.2f us per launch at 1x/2x/4x work: marginal %.2f us per 1x (%.1f TFLOP/s)",
         NW, KP, grid, smem / 1024, t[0], t[1], t[2], (t[2] - t[0]) / 3, 2.0 * ROWS * K * BPAD / ((t[2] - t[0]) / 3 * 1e-6) / 1e12);
  std::vector<float> hO(nO);
  hipMemcpy(hO.data(), dO, nO * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int r : {0, 17, 4095})
    for (int c : {0, 33, 63}) {
      double ref = 0, got = 0;
      for (int k = 0; k < K; ++k) ref += (double)hW[(size_t)r * K + k] * hX[(size_t)k * BPAD + c];
      for (int p = 0; p < nparts; ++p) got += hO[(size_t)p * ROWS * BPAD + ((size_t)(r >> 2) * BPAD + c) * 4 + (r & 3)];
      worst = fmax(worst, fabs(ref - got));
    }
  printf("  max |err| %.3e\n", worst);
}

int main() {
  const int K = 2048, KG = K / 16;
  const size_t nW = (size_t)ROWS * K, nX = (size_t)K * BPAD, nO = (size_t)16 * ROWS * BPAD;
  std::vector<float> hW(nW), hX(nX);
  srand(1);
  for (auto& v : hW) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hX) v = (rand() % 2001 - 1000) * 1e-3f;
  std::vector<float> pW(nW), pX(nX);
  for (int rb = 0; rb < ROWS / 16; ++rb)
    for (int kg = 0; kg < KG; ++kg)
      for (int l = 0; l < 64; ++l)
        for (int s = 0; s < 4; ++s)
          pW[(((size_t)rb * KG + kg) * 64 + l) * 4 + s] = hW[(size_t)(rb * 16 + (l & 15)) * K + kg * 16 + 4 * (l >> 4) + s];
  for (int k = 0; k < K; ++k)
    for (int b = 0; b < BPAD; ++b) pX[((size_t)(k >> 2) * BPAD + b) * 4 + (k & 3)] = hX[(size_t)k * BPAD + b];
  float *dW, *dX, *dO;
  hipMalloc(&dW, nW * 4), hipMalloc(&dX, nX * 4), hipMalloc(&dO, nO * 4);
  hipMemcpy(dW, pW.data(), nW * 4, hipMemcpyHostToDevice);
  hipMemcpy(dX, pX.data(), nX * 4, hipMemcpyHostToDevice);
  run<8, 256>(dW, dX, dO, K, hW, hX);
  run<4, 256>(dW, dX, dO, K, hW, hX);
  run<8, 128>(dW, dX, dO, K, hW, hX);
  run<4, 128>(dW, dX, dO, K, hW, hX);
  run<2, 128>(dW, dX, dO, K, hW, hX);
  return 0;
}
