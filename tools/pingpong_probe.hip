// One-way latency of a tagged 64-bit word between two workgroups: workgroup A writes (tag, value), workgroup B
// polls until it sees the tag and answers the same way; N round trips are timed.  Variants: the two workgroups on
// the same XCD or on different ones (workgroups are dealt round-robin over the 8 XCDs, so blockIdx 0 and 8 share
// one, 0 and 1 do not), buffer from hipMalloc or from hipExtMallocWithFlags(uncached), agent or system scope.
//   hipcc -O3 --offload-arch=gfx950 tools/pingpong_probe.hip -o tools/pingpong_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned long long u64;
constexpr long SPIN_MAX = 20000000;

template <int SCOPE>
__global__ __launch_bounds__(64) void pingpong(u64* buf, int partner_block, int iters, int* fail, int sleep) {
  // buf[0]: A -> B, buf[32]: B -> A (different 128-byte lines)
  const int me = blockIdx.x;
  if (me != 0 && me != partner_block) return;
  if (threadIdx.x != 0) return;
  u64* out = me == 0 ? buf : buf + 32;
  u64* in = me == 0 ? buf + 32 : buf;
  for (int i = 1; i <= iters; ++i) {
    if (me == 0) __hip_atomic_store(out, ((u64)i << 32) | 7u, __ATOMIC_RELAXED, SCOPE);
    long spin = 0;
    for (;; ++spin) {
      const u64 v = __hip_atomic_load(in, __ATOMIC_RELAXED, SCOPE);
      if ((unsigned)(v >> 32) == (unsigned)i) break;
      if (spin > SPIN_MAX) {
        *fail = 1;
        return;
      }
      if (sleep) __builtin_amdgcn_s_sleep(1);
    }
    if (me != 0) __hip_atomic_store(out, ((u64)i << 32) | 9u, __ATOMIC_RELAXED, SCOPE);
  }
}

template <int SCOPE>
void run(const char* what, u64* buf, int partner, int sleep) {
  int* fail;
  hipMalloc(&fail, 4);
  hipMemset(fail, 0, 4);
  hipMemset(buf, 0, 512);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(buf, 0, 512);
    hipDeviceSynchronize();
    hipEventRecord(a);
    pingpong<SCOPE><<<16, 64>>>(buf, partner, iters, fail, sleep);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  int hf = 0;
  hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
  printf("%-46s partner block %2d sleep %d: %.3f us one way%s\n", what, partner, sleep, ms * 1e3 / iters / 2, hf ? "  (TIMED OUT)" : "");
  fflush(stdout);
  hipFree(fail);
}

int main() {
  u64 *plain, *unc, *fine;
  hipMalloc(&plain, 4096);
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&unc), 4096, hipDeviceMallocUncached) != hipSuccess) unc = nullptr;
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&fine), 4096, hipDeviceMallocFinegrained) != hipSuccess) fine = nullptr;
  for (int partner : {8, 1, 4}) {
    for (int sleep : {1, 0}) {
      run<__HIP_MEMORY_SCOPE_AGENT>("hipMalloc, agent scope", plain, partner, sleep);
      run<__HIP_MEMORY_SCOPE_SYSTEM>("hipMalloc, system scope", plain, partner, sleep);
      if (unc) run<__HIP_MEMORY_SCOPE_AGENT>("uncached, agent scope", unc, partner, sleep);
      if (fine) run<__HIP_MEMORY_SCOPE_AGENT>("fine-grained, agent scope", fine, partner, sleep);
    }
  }
  return 0;
}
