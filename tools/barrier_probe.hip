// Grid-barrier probe: what does one "phase" of a persistent kernel cost on this part?
// A phase = every workgroup publishes a slice of a vector, all workgroups meet at a barrier, every
// workgroup reads the whole vector.  Variants:
//   0  barrier only (one agent-scope counter, thread 0 of each workgroup arrives and polls)
//   1  publish 64 B per workgroup with plain stores + release fence, acquire fence + plain loads after
//   2  publish / read with agent-scope relaxed atomics (cache-bypassing accesses), no fences on the data
//   3  as 2 with a two-level barrier (one counter per XCD, the last arriver of an XCD arrives globally)
// Every spin is bounded: after SPIN_MAX polls a workgroup raises an abort flag that ends all spins.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int SPIN_MAX = 4000000;
constexpr int SLICE = 16;  // floats published per workgroup per phase

struct Sync {
  unsigned int count;     // global arrivals (monotonic)
  unsigned int pad0[31];
  unsigned int xcd[8][32];  // per-XCD arrivals, one cache line each
  unsigned int abort_flag;
};

__device__ __forceinline__ bool wait_ge(unsigned int* p, unsigned int target, Sync* s) {
  for (int i = 0; i < SPIN_MAX; ++i) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if ((i & 1023) == 1023 && __hip_atomic_load(&s->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(&s->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

template <int V>
__global__ __launch_bounds__(512) void phase_kernel(Sync* s, float* vec /*[2][G*SLICE]*/, float* out, int iters) {
  const int G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  const int n = G * SLICE;
  float check = 0.f;
  for (int it = 0; it < iters; ++it) {
    float* cur = vec + (size_t)(it & 1) * n;
    if (V >= 1 && tid < SLICE) {
      const float val = (float)(it + 1);
      if (V == 1) cur[wg * SLICE + tid] = val;
      else __hip_atomic_store(&cur[wg * SLICE + tid], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (V == 1) __threadfence();
    __syncthreads();
    if (tid == 0) {
      const unsigned int target = (unsigned int)(it + 1) * G;
      if (V == 3) {
        const int x = wg & 7, per = (G + 7 - x) / 8;  // workgroups dispatched round-robin over XCDs
        const unsigned int old = __hip_atomic_fetch_add(&s->xcd[x][0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == (unsigned int)(it + 1) * per)
          __hip_atomic_fetch_add(&s->count, (unsigned int)per, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      wait_ge(&s->count, target, s);
    }
    __syncthreads();
    if (V == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (V >= 1) {
      float acc = 0.f;
      for (int i = tid; i < n; i += blockDim.x) {
        if (V == 1) acc += cur[i];
        else acc += __hip_atomic_load(&cur[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      check += acc;
    }
  }
  // every element of phase `it` equals it+1: the sum over threads of a workgroup must be n * sum(it+1)
  __shared__ float s_sum[512];
  s_sum[tid] = check;
  __syncthreads();
  if (tid == 0) {
    double t = 0;
    for (int i = 0; i < (int)blockDim.x; ++i) t += s_sum[i];
    out[wg] = (float)(t / n);
  }
}

template <int V>
void run(int G, int iters) {
  Sync* s;
  float *vec, *out;
  hipMalloc(&s, sizeof(Sync));
  hipMalloc(&vec, sizeof(float) * 2 * G * SLICE);
  hipMalloc(&out, sizeof(float) * G);
  hipMemset(s, 0, sizeof(Sync));
  hipMemset(vec, 0, sizeof(float) * 2 * G * SLICE);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  void* args[] = {&s, &vec, &out, &iters};
  hipEventRecord(a);
  hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<void*>(phase_kernel<V>), dim3(G), dim3(512), args, 0, 0);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  Sync hs;
  hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost);
  float o0 = 0, ol = 0;
  hipMemcpy(&o0, out, 4, hipMemcpyDeviceToHost);
  hipMemcpy(&ol, out + G - 1, 4, hipMemcpyDeviceToHost);
  const double want = 0.5 * iters * (iters + 1.0);
  printf("variant %d  G=%d  launch=%s  abort=%u  %.3f us/phase  check %.0f %.0f (want %.0f)\n", V, G,
         hipGetErrorString(e), hs.abort_flag, ms * 1e3 / iters, o0, ol, V ? want : 0.0);
  fflush(stdout);
  hipFree(s), hipFree(vec), hipFree(out);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("device %s  CUs %d  cooperative %d\n", prop.name, cus, prop.cooperativeLaunch);
  const int iters = 2000;
  for (int G : {cus / 8, cus / 2, cus}) {
    run<0>(G, iters);
    run<1>(G, iters);
    run<2>(G, iters);
    run<3>(G, iters);
  }
  return 0;
}
