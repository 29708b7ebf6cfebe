// Probe of a piece kernel for the batched decoder that was measured and NOT adopted (the library's pieces are the
// register-prefetch tiles of GemmOp in tts_amd/csrc/t2_decoder.hip):
// out[part][4096 x 64] = W[4096 x 256 of part] . X[256 x 64] with the activation slice of a workgroup staged in
// LDS and every wave streaming its own weights straight into registers.  Prints us per launch for the decoder
// LSTM's early columns (K = 2048, 8 parts), the 100 MHz timestamps of the phases inside the first and the last
// workgroup, and checks a few outputs against the host.  What it shows: a workgroup spends 3-6 us before its first
// matrix instruction (the L2 hits of the slice queue behind weight misses issued earlier on the same CU), then the
// matrix pipe of the CU is the bound - and all workgroups of a launch do both at the same time.
//   hipcc -O3 --offload-arch=gfx950 tools/part_probe.hip -o tools/part_probe && tools/part_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 4096, BPAD = 64, THREADS = 512;
constexpr int PART_K = 256, PART_KG = 16, PART_COLS = 64;

__device__ long long g_marks[2][8];
#define MARK(i)                                                                                  \
  do {                                                                                           \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                    \
      g_marks[blockIdx.x != 0][i] = wall_clock64();                                              \
  } while (0)

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// VARIANT 0: 4 row blocks x 2 K halves per workgroup, the slice in two phases
// VARIANT 1: no matrix instructions (loads, staging, barriers, stores only)
// VARIANT 2: no global loads of weights (registers hold garbage-free constants)
// VARIANT 3: the whole slice first, the weights requested once it has arrived, one phase
template <int VARIANT>
__global__ __launch_bounds__(THREADS) void part_kernel(const float* __restrict__ Wp, int kg_stride, const float* __restrict__ X,
                                                       float* __restrict__ gates, int gates_part_stride, int ngrp) {
  extern __shared__ float4 part_lds[];
  MARK(0);
  constexpr int NT = PART_COLS / 16, KGH = PART_KG / 2, KGP = KGH / 2;
  constexpr int Q4 = (PART_K / 4 / 4) * PART_COLS;
  const int wg = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int part = wg / ngrp, rb = (wg % ngrp) * 4 + (wave & 3), kh = wave >> 2;
  const float4* W = reinterpret_cast<const float4*>(Wp) + ((size_t)rb * kg_stride + part * PART_KG + kh * KGH) * 64 + lane;
  const float4* X4 = reinterpret_cast<const float4*>(X) + (size_t)part * (PART_K / 4) * PART_COLS + threadIdx.x;
  float4 wv[KGH], a0, a1, a2, a3, b0, b1, b2, b3;
  if (VARIANT == 3) {
    // the whole slice first, then the weights: the L2 hits of the slice do not queue behind the weight misses
    a0 = X4[0], a1 = X4[THREADS], a2 = X4[2 * Q4], a3 = X4[2 * Q4 + THREADS];
    b0 = X4[Q4], b1 = X4[Q4 + THREADS], b2 = X4[3 * Q4], b3 = X4[3 * Q4 + THREADS];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the weights are requested once the slice has arrived)
    float4* xw0 = part_lds + threadIdx.x;
    xw0[0] = a0, xw0[THREADS] = a1, xw0[2 * Q4] = a2, xw0[2 * Q4 + THREADS] = a3;
    xw0[Q4] = b0, xw0[Q4 + THREADS] = b1, xw0[3 * Q4] = b2, xw0[3 * Q4 + THREADS] = b3;
#pragma unroll
    for (int i = 0; i < KGH; ++i) wv[i] = W[(size_t)i * 64];
    lds_barrier();
    MARK(1);
    f32x4 acc3[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc3[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float4* xl3 = part_lds + (size_t)kh * 2 * Q4 + (lane >> 4) * PART_COLS + (lane & 15);
#pragma unroll
    for (int i = 0; i < KGH; ++i) {
      float4 xv[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) xv[n] = xl3[i * 4 * PART_COLS + n * 16];
#define SL3(C) _Pragma("unroll") for (int n = 0; n < 4; ++n) acc3[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i].C, xv[n].C, acc3[n], 0, 0, 0);
      SL3(x) SL3(y) SL3(z) SL3(w)
#undef SL3
      if (i == KGH / 2 - 1) {
        MARK(2);
        MARK(3);
      }
    }
    MARK(4);
    lds_barrier();
    f32x4* xc3 = reinterpret_cast<f32x4*>(part_lds) + (size_t)wave * 2 * 64 + lane;
    xc3[0] = kh ? acc3[0] : acc3[2], xc3[64] = kh ? acc3[1] : acc3[3];
    lds_barrier();
    MARK(5);
    const f32x4* pc3 = reinterpret_cast<const f32x4*>(part_lds) + (size_t)(wave ^ 4) * 2 * 64 + lane;
    float4* O3 = reinterpret_cast<float4*>(gates) + (size_t)part * gates_part_stride + (size_t)(rb * 4 + (lane >> 4)) * PART_COLS +
                 (lane & 15);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 mine = kh ? acc3[2 + j] : acc3[j], other = pc3[64 * j];
      const f32x4 v = kh == 0 ? mine + other : other + mine;
      O3[(2 * kh + j) * 16] = make_float4(v[0], v[1], v[2], v[3]);
    }
    MARK(6);
    return;
  }
  a0 = X4[0], a1 = X4[THREADS], a2 = X4[2 * Q4], a3 = X4[2 * Q4 + THREADS];
#pragma unroll
  for (int i = 0; i < KGP; ++i) wv[i] = VARIANT == 2 ? make_float4(1.f, 2.f, 3.f, lane) : W[(size_t)i * 64];
  b0 = X4[Q4], b1 = X4[Q4 + THREADS], b2 = X4[3 * Q4], b3 = X4[3 * Q4 + THREADS];
#pragma unroll
  for (int i = KGP; i < KGH; ++i) wv[i] = VARIANT == 2 ? make_float4(1.f, 2.f, 3.f, lane) : W[(size_t)i * 64];
  f32x4 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float4* xl = part_lds + (size_t)kh * 2 * Q4 + (lane >> 4) * PART_COLS + (lane & 15);
  float4* xw = part_lds + threadIdx.x;
#define SLICE(C)                                                                                              \
  _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                            \
    if (VARIANT == 1) acc[n][0] += wv[i].C * xv[n].C;                                                         \
    else acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i].C, xv[n].C, acc[n], 0, 0, 0);                    \
  }
#define PHASE(PH)                                                                          \
  lds_barrier();                                                                           \
  MARK(1 + 2 * (PH));                                                                      \
  _Pragma("unroll") for (int ii = 0; ii < KGP; ++ii) {                                     \
    const int i = (PH) * KGP + ii;                                                         \
    float4 xv[NT];                                                                         \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) xv[n] = xl[i * 4 * PART_COLS + n * 16]; \
    SLICE(x) SLICE(y) SLICE(z) SLICE(w)                                                    \
  }                                                                                        \
  MARK(2 + 2 * (PH));
  xw[0] = a0, xw[THREADS] = a1, xw[2 * Q4] = a2, xw[2 * Q4 + THREADS] = a3;
  PHASE(0)
  xw[Q4] = b0, xw[Q4 + THREADS] = b1, xw[3 * Q4] = b2, xw[3 * Q4 + THREADS] = b3;
  PHASE(1)
  lds_barrier();
  f32x4* xc = reinterpret_cast<f32x4*>(part_lds) + (size_t)wave * 2 * 64 + lane;
  xc[0] = kh ? acc[0] : acc[2], xc[64] = kh ? acc[1] : acc[3];
  lds_barrier();
  MARK(5);
  const f32x4* pc = reinterpret_cast<const f32x4*>(part_lds) + (size_t)(wave ^ 4) * 2 * 64 + lane;
  float4* O = reinterpret_cast<float4*>(gates) + (size_t)part * gates_part_stride + (size_t)(rb * 4 + (lane >> 4)) * PART_COLS +
              (lane & 15);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const f32x4 mine = kh ? acc[2 + j] : acc[j], other = pc[64 * j];
    const f32x4 v = kh == 0 ? mine + other : other + mine;
    O[(2 * kh + j) * 16] = make_float4(v[0], v[1], v[2], v[3]);
  }
  MARK(6);
}

// ---------------------------------------------------------------------------------------------------------------
// STAGED, one round: 256 workgroups, workgroup = (row group of 8 row blocks = 128 rows, one wave each) x (one K part
// of KGP k-groups) x all 64 columns.  The activation slice of the part (KGP * 16 x 64 floats) goes through LDS and is
// shared by the eight waves; every wave streams the weights of its own row block into registers and owns its rows
// for the whole part, so there is no combine across waves.  Request order: every wave's slice loads, a barrier
// (issue only - nothing is waited for), then the weights: the slice - L2 hits - never queues behind weight misses of
// another wave of the CU.  The slice lands in chunks of CH k-groups; matrix instructions of chunk c start when it
// is in LDS.
// MODE 0: as described; 1: no barrier between slice and weight requests; 2: no matrix instructions; 3: no weights
template <int KGP, int CH, int MODE>
__global__ __launch_bounds__(THREADS) void staged_kernel(const float* __restrict__ Wp, int kg_stride,
                                                         const float* __restrict__ X, float* __restrict__ gates,
                                                         int gates_part_stride, int nrg) {
  extern __shared__ float4 part_lds[];
  MARK(0);
  constexpr int NJ = KGP / 2;  // slice loads per thread: 512 float4 = two k-groups each
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int part = wg / nrg, rb = (wg % nrg) * 8 + wave;
  const float4* Xs = reinterpret_cast<const float4*>(X) + (size_t)part * KGP * 256 + tid;
  float4 xr[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) xr[j] = Xs[j * THREADS];
  if (MODE != 1) asm volatile("s_barrier" ::: "memory");
  const float4* W = reinterpret_cast<const float4*>(Wp) + ((size_t)rb * kg_stride + part * KGP) * 64 + lane;
  float4 wv[KGP];
#pragma unroll
  for (int i = 0; i < KGP; ++i) wv[i] = MODE == 3 ? make_float4(1.f, 2.f, 3.f, lane) : W[(size_t)i * 64];
  f32x4 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float4* xl = part_lds + (lane >> 4) * 64 + (lane & 15);
#pragma unroll
  for (int c = 0; c < KGP / CH; ++c) {
#pragma unroll
    for (int jj = 0; jj < CH / 2; ++jj) part_lds[tid + (c * (CH / 2) + jj) * THREADS] = xr[c * (CH / 2) + jj];
    lds_barrier();
    if (c == 0) MARK(1);
#pragma unroll
    for (int ii = 0; ii < CH; ++ii) {
      const int i = c * CH + ii;
      float4 xv[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) xv[n] = xl[i * 256 + n * 16];
#define SLS(C)                                                                                   \
  _Pragma("unroll") for (int n = 0; n < 4; ++n) {                                                \
    if (MODE == 2) acc[n][0] += wv[i].C * xv[n].C;                                               \
    else acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i].C, xv[n].C, acc[n], 0, 0, 0);       \
  }
      SLS(x) SLS(y) SLS(z) SLS(w)
#undef SLS
    }
    if (c == 0) MARK(2);
  }
  MARK(3);
  float4* O = reinterpret_cast<float4*>(gates) + (size_t)part * gates_part_stride + (size_t)(rb * 4 + (lane >> 4)) * PART_COLS +
              (lane & 15);
#pragma unroll
  for (int n = 0; n < 4; ++n) O[n * 16] = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
  MARK(4);
}

template <int KGP, int CH, int MODE>
void run_staged(const char* what, const float* dW, const float* dX, float* dO, int K, const std::vector<float>& hW,
                const std::vector<float>& hX, bool check) {
  const int KG = K / 16, nparts = KG / KGP, nrg = ROWS / 128;
  const size_t nO = (size_t)nparts * ROWS * BPAD;
  const int smem = KGP * 256 * 16;
  hipFuncSetAttribute(reinterpret_cast<const void*>(staged_kernel<KGP, CH, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  const int grid = nrg * nparts;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    for (int i = 0; i < 200; ++i)
      staged_kernel<KGP, CH, MODE><<<grid, THREADS, smem>>>(dW, KG, dX, dO, ROWS / 4 * BPAD, nrg);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  long long m[2][8];
  hipMemcpyFromSymbol(m, HIP_SYMBOL(g_marks), sizeof(m));
  printf("%-44s grid %4d  %.2f us per launch (%.1f TFLOP/s)\n", what, grid, ms * 5.f, 2.0 * ROWS * K * BPAD / (ms * 5e-6) / 1e12);
  for (int w = 0; w < 2; ++w)
    printf("   %s workgroup: first chunk in LDS %.2f us, its MFMAs %.2f, remaining chunks %.2f, store %.2f | total %.2f\n",
           w ? "last " : "first", (m[w][1] - m[w][0]) / 100.0, (m[w][2] - m[w][1]) / 100.0, (m[w][3] - m[w][2]) / 100.0,
           (m[w][4] - m[w][3]) / 100.0, (m[w][4] - m[w][0]) / 100.0);
  if (!check) return;
  std::vector<float> hO(nO);
  hipMemcpy(hO.data(), dO, nO * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int r : {0, 17, 130, 4095})
    for (int c : {0, 33, 63}) {
      double ref = 0, got = 0;
      for (int k = 0; k < K; ++k) ref += (double)hW[(size_t)r * K + k] * hX[(size_t)k * BPAD + c];
      for (int p = 0; p < nparts; ++p) got += hO[(size_t)p * ROWS * BPAD + ((size_t)(r >> 2) * BPAD + c) * 4 + (r & 3)];
      worst = fmax(worst, fabs(ref - got));
    }
  printf("   max |err| %.3e\n", worst);
}

template <int VARIANT>
void run(const char* what, const float* dW, const float* dX, float* dO, int K, const std::vector<float>& hW,
         const std::vector<float>& hX, bool check) {
  const int KG = K / 16, nparts = K / PART_K, ngrp = ROWS / 16 / 4;
  const size_t nO = (size_t)nparts * ROWS * BPAD;
  const int smem = PART_K * PART_COLS * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(part_kernel<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  const int grid = ngrp * nparts;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    for (int i = 0; i < 200; ++i)
      part_kernel<VARIANT><<<grid, THREADS, smem>>>(dW, KG, dX, dO, ROWS / 4 * BPAD, ngrp);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  long long m[2][8];
  hipMemcpyFromSymbol(m, HIP_SYMBOL(g_marks), sizeof(m));
  printf("%-28s grid %4d  %.2f us per launch (%.1f TFLOP/s)\n", what, grid, ms * 5.f,
         2.0 * ROWS * K * BPAD / (ms * 5e-6) / 1e12);
  for (int w = 0; w < 2; ++w)
    printf("   %s workgroup: to phase 0 %.2f us, MFMAs %.2f, to phase 1 %.2f, MFMAs %.2f, combine %.2f, store %.2f | total %.2f\n",
           w ? "last " : "first", (m[w][1] - m[w][0]) / 100.0, (m[w][2] - m[w][1]) / 100.0, (m[w][3] - m[w][2]) / 100.0,
           (m[w][4] - m[w][3]) / 100.0, (m[w][5] - m[w][4]) / 100.0, (m[w][6] - m[w][5]) / 100.0, (m[w][6] - m[w][0]) / 100.0);
  printf("   last workgroup starts %.2f us after the first\n", (m[1][0] - m[0][0]) / 100.0);
  if (!check) return;
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
  printf("   max |err| %.3e\n", worst);
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
  run<0>("slice in two phases", dW, dX, dO, K, hW, hX, true);
  run<1>("no matrix instructions", dW, dX, dO, K, hW, hX, false);
  run<2>("no weight loads", dW, dX, dO, K, hW, hX, false);
  run<3>("slice first, then weights", dW, dX, dO, K, hW, hX, true);
  hipMemset(dO, 0, nO * 4);
  run_staged<16, 4, 0>("staged one round 128x64x256, chunks of 4", dW, dX, dO, K, hW, hX, true);
  run_staged<16, 4, 1>("  no barrier between slice and weights", dW, dX, dO, K, hW, hX, false);
  run_staged<16, 4, 2>("  no matrix instructions", dW, dX, dO, K, hW, hX, false);
  run_staged<16, 4, 3>("  no weight loads", dW, dX, dO, K, hW, hX, false);
  run_staged<16, 2, 0>("staged, chunks of 2", dW, dX, dO, K, hW, hX, true);
  run_staged<16, 8, 0>("staged, chunks of 8", dW, dX, dO, K, hW, hX, true);
  run_staged<8, 4, 0>("staged 128x64x128 (512 workgroups)", dW, dX, dO, K, hW, hX, true);
  return 0;
}
