// Shared helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <mutex>

#include "../../include/ttsx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void ttsx_set_error(const char* fmt, ...);
std::mutex& ttsx_team_mutex();  // serialises the persistent team kernels of a process (errors.cpp)
// true once after ttsx_debug_fail_next_wait named this kind: the run that asks reports an expired wait (errors.cpp)
bool ttsx_take_wait_failure(unsigned kind);

#define TTSX_HIP(expr)                                                                     \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess) {                                                               \
      ttsx_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
      return TTSX_ERR_HIP;                                                                 \
    }                                                                                      \
  } while (0)

#define TTSX_ARG(cond, ...)        \
  do {                             \
    if (!(cond)) {                 \
      ttsx_set_error(__VA_ARGS__); \
      return TTSX_ERR_ARG;         \
    }                              \
  } while (0)

#define TTSX_LAUNCH_CHECK() TTSX_HIP(hipGetLastError())

// Runs a callable when the scope ends unless dismissed.  The create functions use it to release a half-built
// handle and their temporaries on every early return (the checking macros above return from the middle).
template <class F>
struct ScopeExit {
  F f;
  bool armed;
  explicit ScopeExit(F fn) : f(fn), armed(true) {}
  ScopeExit(const ScopeExit&) = delete;
  ScopeExit& operator=(const ScopeExit&) = delete;
  ~ScopeExit() {
    if (armed) f();
  }
  void dismiss() { armed = false; }
};

// Waits of the persistent team kernels are bounded in wall-clock time, not in polling rounds: a workgroup that has
// not seen its inputs TEAM_WAIT_TICKS after its 1024th polling round (the 100 MHz constant clock; 50 ms - a step of
// a resident team takes microseconds) raises the abort word, and the host falls back to the launch chain.
constexpr long long TEAM_WAIT_TICKS = 5000000;
__device__ static inline bool team_wait_expired(long spin, long long& t0) {
  if ((spin & 1023) != 1023) return false;
  const long long now = wall_clock64();
  if (spin == 1023) {
    t0 = now;
    return false;
  }
  return now - t0 > TEAM_WAIT_TICKS;
}

// After a team run that timed out (its workgroups were not resident together: another process or a long kernel
// held CUs) the default plan uses the launch chain and tries the team again once this many seconds have passed.
constexpr double TEAM_COOL_DOWN_S = 5.0;
struct TeamCoolDown {
  std::chrono::steady_clock::time_point until{};
  bool cooling() const { return std::chrono::steady_clock::now() < until; }
  void start() {
    until = std::chrono::steady_clock::now() +
            std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(TEAM_COOL_DOWN_S));
  }
};

__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// Activation vectors that feed the row-block GEMMs are stored "quad-interleaved":
// element k of utterance b lives at ((k/4) * Bpad + b) * 4 + (k%4), so that the four
// consecutive k a lane needs for one MFMA fragment are a single 16-byte load and 16
// neighbouring utterances are 256 contiguous bytes.
__host__ __device__ static inline size_t qidx(int k, int b, int Bpad) {
  return ((size_t)(k >> 2) * Bpad + b) * 4 + (k & 3);
}

__device__ static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ static inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ static inline float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
