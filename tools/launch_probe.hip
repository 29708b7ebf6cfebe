// Launch-chain probe: cost per dependent kernel inside a hipGraph, as a function of what the kernel
// does (nothing / one global load -> store / a chain of two dependent loads) and its grid size.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_empty(float* p) {}
__global__ void k_load1(float* p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  p[i + 65536] = p[i] + 1.f;
}
__global__ void k_load2(float* p, const int* idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  p[i + 65536] = p[idx[i]] + 1.f;
}

template <class F>
float time_graph(F enqueue, int n, int reps) {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) enqueue(st);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  hipEventRecord(a, st);
  for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
  hipEventRecord(b, st);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / (n * reps);
}

int main() {
  float* p;
  int* idx;
  hipMalloc(&p, sizeof(float) * 4 * 65536 * 4);
  hipMalloc(&idx, sizeof(int) * 65536 * 4);
  hipMemset(p, 0, sizeof(float) * 4 * 65536 * 4);
  hipMemset(idx, 0, sizeof(int) * 65536 * 4);
  for (int wgs : {1, 32, 256, 1024}) {
    const float e = time_graph([&](hipStream_t s) { k_empty<<<wgs, 256, 0, s>>>(p); }, 224, 20);
    const float l1 = time_graph([&](hipStream_t s) { k_load1<<<wgs, 256, 0, s>>>(p); }, 224, 20);
    const float l2 = time_graph([&](hipStream_t s) { k_load2<<<wgs, 256, 0, s>>>(p, idx); }, 224, 20);
    printf("grid %4d x 256: empty %.2f us, load+store %.2f us, two dependent loads %.2f us per kernel\n", wgs, e, l1, l2);
  }
  return 0;
}
