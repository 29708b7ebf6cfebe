// What does one all-to-all exchange of the Tacotron (v1) team kernel cost with nothing else going on?
// G workgroups of 512 threads; per iteration every wave publishes one tagged 64-bit word (a 256-word vector in
// all), then every workgroup waits until it has seen all 256 words of the iteration, then a workgroup barrier.
// Variants:
//   REP     1 or 8 replicas of the vector (workgroup g reads replica g % REP)
//   MODE 0  thread i < 256 polls word i (4 waves poll one word per lane)
//   MODE 1  wave 0 polls the whole vector (4 words per lane), the others wait at the barrier
//   MODE 2  every wave polls the whole vector (4 words per lane) - no LDS hand-over needed
//   MODE 3  as 0, and the vector is 512 words (every wave publishes two)
//   MODE 4  as 0, but the 8 words of a workgroup are gathered in LDS and written by ONE store instruction of wave 0
//           (64 contiguous bytes per replica) instead of 8 waves writing 8 bytes each into a line they share
//   MODE 5  as 4 with every workgroup's words padded to a 128-byte line of their own (vector spread over 32 lines)
//   wide    256 workgroups, a 1024-word vector: every workgroup publishes 4 contiguous words with one store
//           instruction and reads all 1024 (two words per thread)
// Every wait is bounded.
//   hipcc -O3 --offload-arch=gfx950 tools/exchange_probe.hip -o tools/exchange_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned long long u64;
constexpr long SPIN_MAX = 4000000;
constexpr int REP_STRIDE = 1024 + 544;  // words

__device__ __forceinline__ u64 ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int REP, int MODE>
__global__ __launch_bounds__(512) void exchange_kernel(u64* buf, int iters, int* fail, float* sink) {
  __shared__ float s_v[512];
  __shared__ int s_fail;
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NW = MODE == 3 ? 512 : 256;
  if (tid == 0) s_fail = 0;
  __syncthreads();
  float acc = 0.f;
  for (int it = 1; it <= iters; ++it) {
    u64* X = buf + (size_t)(it & 1) * REP * REP_STRIDE;
    const u64 word = ((u64)it << 32) | (unsigned)(g * 8 + wave);
    if (MODE >= 4) {
      __shared__ u64 s_pub[8];
      if (lane == 0) s_pub[wave] = word;
      __syncthreads();
      if (wave == 0 && lane < 8 * REP) {
        const int idx = MODE == 4 ? 8 * g + (lane & 7) : 16 * g + (lane & 7);
        __hip_atomic_store(X + (size_t)(lane >> 3) * REP_STRIDE + idx, s_pub[lane & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (lane < REP) {
      for (int k = 0; k < 32 / (int)gridDim.x; ++k)  // fewer workgroups: every wave publishes several words
        __hip_atomic_store(X + (size_t)lane * REP_STRIDE + 8 * (g + (int)gridDim.x * k) + wave, word, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      if (MODE == 3)
        __hip_atomic_store(X + (size_t)lane * REP_STRIDE + 256 + 8 * g + wave, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const u64* R = X + (size_t)(g % REP) * REP_STRIDE;
    bool bad = false;
    if (MODE == 0 || MODE >= 3) {
      if (tid < NW) {
        for (long spin = 0;; ++spin) {
          const u64 v = ld(R + (MODE == 5 ? 16 * (tid >> 3) + (tid & 7) : tid));
          if (__all((unsigned)(v >> 32) == (unsigned)it)) {
            s_v[tid] = (float)(unsigned)v;
            break;
          }
          if (spin > SPIN_MAX || ((spin & 1023) == 1023 && *(volatile int*)fail)) {
            bad = true;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    } else if (MODE == 1 ? wave == 0 : true) {
      for (long spin = 0;; ++spin) {
        u64 v[4];
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ld(R + lane + 64 * j), ok = ok && (unsigned)(v[j] >> 32) == (unsigned)it;
        if (__all(ok)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (MODE == 1) s_v[lane + 64 * j] = (float)(unsigned)v[j];
            else acc += (float)(unsigned)v[j];
          }
          break;
        }
        if (spin > SPIN_MAX || ((spin & 1023) == 1023 && *(volatile int*)fail)) {
          bad = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    if (bad) s_fail = 1, *fail = 1;
    __syncthreads();
    if (s_fail) return;
    if (MODE != 2) acc += s_v[lane] + s_v[lane + 64] + s_v[lane + 128] + s_v[lane + 192];
  }
  if (acc == -1.f) sink[0] = acc;
}

template <int REP, int NWORD>
__global__ __launch_bounds__(512) void wide_kernel(u64* buf, int iters, int* fail, float* sink) {
  __shared__ float s_v[NWORD];
  __shared__ int s_fail;
  const int g = blockIdx.x, tid = threadIdx.x, G = gridDim.x;
  const int per = NWORD / G;  // words a workgroup publishes
  constexpr int NL = NWORD / 512;
  if (tid == 0) s_fail = 0;
  __syncthreads();
  float acc = 0.f;
  for (int it = 1; it <= iters; ++it) {
    u64* X = buf + (size_t)(it & 1) * REP * (NWORD + 544);
    if (tid < per * REP)
      __hip_atomic_store(X + (size_t)(tid / per) * (NWORD + 544) + per * g + tid % per, ((u64)it << 32) | (unsigned)g,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64* R = X + (size_t)(g % REP) * (NWORD + 544);
    bool bad = false;
    for (long spin = 0;; ++spin) {
      u64 v[NL];
      bool ok = true;
#pragma unroll
      for (int j = 0; j < NL; ++j) v[j] = ld(R + tid + 512 * j), ok = ok && (unsigned)(v[j] >> 32) == (unsigned)it;
      if (__all(ok)) {
#pragma unroll
        for (int j = 0; j < NL; ++j) s_v[tid + 512 * j] = (float)(unsigned)v[j];
        break;
      }
      if (spin > SPIN_MAX || ((spin & 1023) == 1023 && *(volatile int*)fail)) {
        bad = true;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (bad) s_fail = 1, *fail = 1;
    __syncthreads();
    if (s_fail) return;
    acc += s_v[tid] + s_v[(tid + 64) % NWORD];
    __syncthreads();
  }
  if (acc == -1.f) sink[0] = acc;
}

template <int REP, int NWORD>
void run_wide(int G) {
  u64* buf;
  int* fail;
  float* sink;
  const size_t bytes = sizeof(u64) * 2 * REP * (NWORD + 544);
  hipMalloc(&buf, bytes), hipMalloc(&fail, 4), hipMalloc(&sink, 4);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  int iters = 4000;
  float ms = 0;
  hipError_t e = hipSuccess;
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(buf, 0, bytes), hipMemset(fail, 0, 4);
    hipDeviceSynchronize();
    void* args[] = {&buf, &iters, &fail, &sink};
    hipEventRecord(a);
    e = hipLaunchCooperativeKernel(reinterpret_cast<void*>(wide_kernel<REP, NWORD>), dim3(G), dim3(512), args, 0, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  int hf = 0;
  hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
  printf("wide: G=%d words %d replicas %d: %.3f us per exchange (%s)%s\n", G, NWORD, REP, ms * 1e3 / iters, hipGetErrorString(e),
         hf ? "  (TIMED OUT)" : "");
  fflush(stdout);
  hipFree(buf), hipFree(fail), hipFree(sink);
}

template <int REP, int MODE>
void run(int G) {
  u64* buf;
  int* fail;
  float* sink;
  const size_t bytes = sizeof(u64) * 2 * 8 * REP_STRIDE;
  hipMalloc(&buf, bytes), hipMalloc(&fail, 4), hipMalloc(&sink, 4);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  const int iters = 4000;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(buf, 0, bytes), hipMemset(fail, 0, 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    exchange_kernel<REP, MODE><<<G, 512, 0>>>(buf, iters, fail, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  int hf = 0;
  hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
  printf("G=%d replicas %d mode %d: %.3f us per exchange%s\n", G, REP, MODE, ms * 1e3 / iters, hf ? "  (TIMED OUT)" : "");
  fflush(stdout);
  hipFree(buf), hipFree(fail), hipFree(sink);
}

int main() {
  run<1, 0>(32), run<8, 0>(32);
  run<1, 1>(32), run<8, 1>(32);
  run<1, 2>(32), run<8, 2>(32);
  run<1, 3>(32), run<8, 3>(32);
  run_wide<1, 1024>(256), run_wide<8, 1024>(256), run_wide<8, 512>(256), run_wide<8, 1024>(128), run_wide<8, 512>(64);
  run<1, 4>(32), run<8, 4>(32);
  run<1, 5>(32), run<8, 5>(32);
  return 0;
}
