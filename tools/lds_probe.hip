// LDS access patterns of the Griffin-Lim transposes (8-byte elements), twelve waves on one CU (each in its own
// 1088-element region, as in the kernel): clock ticks per wave-level LDS instruction of the whole CU, so that
// bank conflicts of a layout show up as multiples of the linear pattern.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_probe.hip -o tools/lds_probe && tools/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int REPS = 200;

template <int PAT>
__device__ __forceinline__ int addr(int lane, int r) {
  switch (PAT) {
    case 0: return lane + 64 * r;                                        // linear
    case 1: return 17 * lane + r;                                        // pass 2, first write
    case 2: return lane + (lane >> 4) + 68 * r;                          // pass 2 read / pass 1 read
    case 3: return (lane >> 4) * 272 + (lane & 15) + 17 * r;             // pass 2, second write
    case 4: return lane + (lane >> 4) + 68 * (r & 3) + 272 * (r >> 2);   // pass 3 read
    case 5: {                                                            // partner bins, descending
      const int k = lane + 64 * (r & 7), e = (1024 - k) & 1023;
      return e + (e >> 4);
    }
    case 6: return 16 * lane + ((r + lane) & 15);                        // rotated rows, no padding
    case 7: return 16 * lane + ((r + (lane >> 1)) & 15);                 // rotated by lane / 2
    case 8: return 18 * lane + r;                                        // pitch 18
    case 9: return 16 * lane + (r ^ (lane & 15));                        // xor swizzle
    default: return 0;
  }
}

constexpr int WAVES = 12, REGION = 1088;
template <int PAT, bool WRITE>
__global__ __launch_bounds__(64 * WAVES) void probe(long long* out, float* sink) {
  extern __shared__ f2 s_all[];
  const int lane = threadIdx.x & 63;
  f2* s = s_all + (threadIdx.x >> 6) * REGION;
  for (int i = lane; i < REGION; i += 64) s[i] = (f2){(float)i, 1.f};
  __syncthreads();
  int a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = addr<PAT>(lane, r) % REGION;
  f2 acc = {0.f, 0.f};
  const long long t0 = clock64();
  for (int it = 0; it < REPS; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (WRITE) s[a[r]] = acc + (float)r;
      else acc += s[a[r]];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  __syncthreads();
  const long long t2 = clock64();
  if (threadIdx.x == 0) out[0] = t2 - t0;
  (void)t1;
  sink[threadIdx.x] = acc.x + acc.y + s[lane].x;
}

template <int PAT>
void run(const char* name, long long* d_out, float* d_sink) {
  long long h[2];
  for (int w = 0; w < 2; ++w) {
    for (int rep = 0; rep < 2; ++rep) {
      const int smem = WAVES * REGION * 8;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<PAT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<PAT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (w) probe<PAT, true><<<1, 64 * WAVES, smem>>>(d_out, d_sink);
      else probe<PAT, false><<<1, 64 * WAVES, smem>>>(d_out, d_sink);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&h[w], d_out, sizeof(long long), hipMemcpyDeviceToHost);
  }
  printf("%-44s read %.2f  write %.2f  counter ticks per b64 instruction\n", name, h[0] / (double)(REPS * 16 * WAVES),
         h[1] / (double)(REPS * 16 * WAVES));
}

int main() {
  long long* d_out;
  float* d_sink;
  (void)hipMalloc(&d_out, 64);
  (void)hipMalloc(&d_sink, 4096);
  run<0>("linear lane + 64 r", d_out, d_sink);
  run<1>("17 lane + r", d_out, d_sink);
  run<2>("lane + lane/16 + 68 r", d_out, d_sink);
  run<3>("272 (lane/16) + lane%16 + 17 r", d_out, d_sink);
  run<4>("lane + lane/16 + 68 i + 272 r", d_out, d_sink);
  run<5>("partner bins (descending, padded)", d_out, d_sink);
  run<6>("16 lane + (r + lane) % 16", d_out, d_sink);
  run<7>("16 lane + (r + lane/2) % 16", d_out, d_sink);
  run<8>("18 lane + r", d_out, d_sink);
  run<9>("16 lane + (r ^ lane%16)", d_out, d_sink);
  return 0;
}
