// Read-bandwidth probe: how fast can a kernel stream a 76 MB (cache-resident across repeats) or
// 1 GB buffer, as a function of workgroups, threads and 16-byte loads in flight per lane.
// Pattern A ("rowblock"): workgroup w owns a contiguous slab, its waves take 1 KiB chunks round-robin
// (what the decoder GEMV does).  Pattern B ("grid-stride"): chunk i*gridwaves + wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int U, int PATTERN>
__global__ void read_kernel(const float4* __restrict__ src, float* __restrict__ sink, size_t chunks_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float acc = 0.f;
  if (PATTERN == 0) {
    const float4* base = src + (size_t)blockIdx.x * chunks_per_wg * 64 + lane;
    for (size_t c = wave; c < chunks_per_wg; c += (size_t)nw * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t cc = c + (size_t)u * nw;
        v[u] = cc < chunks_per_wg ? base[cc * 64] : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
  } else {
    const size_t total = chunks_per_wg * gridDim.x, gw = (size_t)gridDim.x * nw, me = (size_t)blockIdx.x * nw + wave;
    for (size_t c = me; c < total; c += gw * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t cc = c + (size_t)u * gw;
        v[u] = cc < total ? src[cc * 64 + lane] : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int U, int PATTERN>
float run(const float4* src, float* sink, size_t bytes, int wgs, int threads, int reps) {
  const size_t chunks = bytes / 1024, cpw = chunks / wgs;
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) read_kernel<U, PATTERN><<<wgs, threads>>>(src, sink, cpw);
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) read_kernel<U, PATTERN><<<wgs, threads>>>(src, sink, cpw);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return (float)(cpw * wgs * 1024.0 * reps / (ms * 1e-3) / 1e9);
}

int main() {
  float* sink;
  hipMalloc(&sink, 4);
  for (size_t mb : {29, 76, 1024}) {
    const size_t bytes = mb << 20;
    float4* src;
    hipMalloc(&src, bytes);
    hipMemset(src, 0, bytes);
    const int reps = mb > 500 ? 10 : 100;
    printf("== %zu MB\n", mb);
    for (int wgs : {256, 512, 1024, 2048, 4096}) {
      for (int threads : {256, 512, 1024}) {
        printf("wgs %4d thr %4d | A: U4 %5.0f U8 %5.0f U16 %5.0f | B: U4 %5.0f U8 %5.0f U16 %5.0f GB/s\n", wgs, threads,
               run<4, 0>(src, sink, bytes, wgs, threads, reps), run<8, 0>(src, sink, bytes, wgs, threads, reps),
               run<16, 0>(src, sink, bytes, wgs, threads, reps), run<4, 1>(src, sink, bytes, wgs, threads, reps),
               run<8, 1>(src, sink, bytes, wgs, threads, reps), run<16, 1>(src, sink, bytes, wgs, threads, reps));
      }
    }
    hipFree(src);
  }
  return 0;
}
