// Post-vocoder tail of a request on the device.
//
// Follows Synthesizer.tts (Tacotron2/TTS/synthesizer.py:141-174): every sentence's waveform is cut
// at AudioProcessor.find_endpoint (Tacotron2/TTS/utils.py:944-951: the first position x = hop,
// 2 hop, ... < len - window whose next `window` samples all stay below the threshold gives the end
// x + hop; note the test is on the signed samples, np.max(wav[x:x+window]) < threshold), the
// pieces are concatenated with `gap` zero samples after each, and save_wav (utils.py:794-796)
// scales by 32767 / max(0.01, max|wav|) in float64 and truncates to int16.
#include "common.h"

namespace {

struct TailP {
  const float* wav;          // [B][pitch]
  const int32_t* n_samples;  // [B] or null
  int32_t* endpoints;        // [B]
  int32_t* offsets;          // [B + 2] start of every sentence in the output, total at [B], peak word at [B + 1]
  unsigned* peak_bits;       // = offsets + B + 1: bit pattern of max |sample| (non-negative floats order like
                             // unsigned ints)
  int16_t* pcm;
  int B, pitch, window, hop, gap, do_trim;
  float threshold;
};

__global__ void tail_init_kernel(TailP p) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < p.B) p.endpoints[b] = p.n_samples ? min(max(p.n_samples[b], 0), p.pitch) : p.pitch;
  if (b == 0) *p.peak_bits = 0u;
}

// one workgroup per candidate position x = (blockIdx.x + 1) * hop of utterance blockIdx.y
__global__ __launch_bounds__(256) void endpoint_kernel(TailP p) {
  const int b = blockIdx.y, x = (blockIdx.x + 1) * p.hop;
  const int len = p.n_samples ? min(max(p.n_samples[b], 0), p.pitch) : p.pitch;
  if (x >= len - p.window) return;
  const float* w = p.wav + (size_t)b * p.pitch + x;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < p.window; i += 256) m = fmaxf(m, w[i]);
  m = wave_max(m);
  __shared__ float s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    if (m < p.threshold) atomicMin(&p.endpoints[b], x + p.hop);  // the first qualifying x wins
  }
}

__global__ void offsets_kernel(TailP p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int off = 0;
  for (int b = 0; b < p.B; ++b) {
    p.offsets[b] = off;
    off += p.endpoints[b] + p.gap;
  }
  p.offsets[p.B] = off;
}

__global__ __launch_bounds__(256) void peak_kernel(TailP p) {
  const int b = blockIdx.y, n = p.endpoints[b];
  const float* w = p.wav + (size_t)b * p.pitch;
  float m = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(p.peak_bits, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void pack_kernel(TailP p) {
  const int b = blockIdx.y, n = p.endpoints[b], off = p.offsets[b];
  const float* w = p.wav + (size_t)b * p.pitch;
  const double scale = 32767.0 / fmax(0.01, (double)__uint_as_float(*p.peak_bits));
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n + p.gap; i += gridDim.x * 256) {
    int16_t v = 0;
    if (i < n) v = (int16_t)(int)((double)w[i] * scale);  // C truncation = ndarray.astype(int16) on in-range values
    p.pcm[(size_t)off + i] = v;
  }
}

}  // namespace

extern "C" int ttsx_request_tail(const float* wav, const int32_t* n_samples, int B, int pitch, int window_length,
                                 int hop_length, float threshold, int gap, int do_trim, int32_t* endpoints,
                                 int32_t* offsets, int16_t* pcm, long capacity, long* total_host, float* peak_host,
                                 void* stream) {
  TTSX_ARG(wav && endpoints && offsets && pcm, "ttsx_request_tail: null argument");
  TTSX_ARG(B >= 1 && pitch >= 1 && gap >= 0, "ttsx_request_tail: bad shape B=%d pitch=%d gap=%d", B, pitch, gap);
  TTSX_ARG(window_length >= 1 && hop_length >= 1, "ttsx_request_tail: window_length=%d hop_length=%d", window_length,
           hop_length);
  TTSX_ARG(capacity >= (long)B * ((long)pitch + gap), "ttsx_request_tail: pcm capacity %ld < B*(pitch+gap) = %ld",
           capacity, (long)B * ((long)pitch + gap));
  // (offsets are int32)
  TTSX_ARG((long)B * ((long)pitch + gap) <= 2147483647L, "ttsx_request_tail: B*(pitch+gap) = %ld samples exceed int32",
           (long)B * ((long)pitch + gap));
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned* peak_bits = reinterpret_cast<unsigned*>(offsets + B + 1);
  TailP p{wav, n_samples, endpoints, offsets, peak_bits, pcm, B, pitch, window_length, hop_length, gap, do_trim, threshold};
  tail_init_kernel<<<ceil_div(B, 64), 64, 0, st>>>(p);
  const int n_cand = pitch / hop_length;
  if (do_trim && n_cand > 0) endpoint_kernel<<<dim3(n_cand, B), 256, 0, st>>>(p);
  offsets_kernel<<<1, 1, 0, st>>>(p);
  const int gx = std::max(1, std::min(64, ceil_div(pitch + gap, 4096)));
  peak_kernel<<<dim3(gx, B), 256, 0, st>>>(p);
  pack_kernel<<<dim3(gx, B), 256, 0, st>>>(p);
  TTSX_LAUNCH_CHECK();
  int32_t total = 0;
  unsigned bits = 0;
  TTSX_HIP(hipMemcpyAsync(&total, offsets + B, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipMemcpyAsync(&bits, peak_bits, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  TTSX_HIP(hipStreamSynchronize(st));
  if (total_host) *total_host = total;
  if (peak_host) memcpy(peak_host, &bits, sizeof(float));
  return TTSX_OK;
}
