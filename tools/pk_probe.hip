// Checks the packed complex helpers of csrc/pk_complex.h against plain scalar arithmetic on the device.
// Build: hipcc --offload-arch=gfx950 -O3 -I tts_amd/csrc tools/pk_probe.hip -o /tmp/pk_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

#include "pk_complex.h"

__global__ void probe(const float* in, float* out) {
  const int t = threadIdx.x;
  const f2 a = {in[4 * t], in[4 * t + 1]}, b = {in[4 * t + 2], in[4 * t + 3]};
  f2* o = reinterpret_cast<f2*>(out) + 16 * t;
  o[0] = pk_addmj(a, b);
  o[1] = pk_addpj(a, b);
  o[2] = pk_cmul<false>(a, b);
  o[3] = pk_cmul<true>(a, b);
  const f2 c = {0.92387953251128674f, 0.38268343236508977f};
  o[4] = pk_cmulc<false>(a, c);
  o[5] = pk_cmulc<true>(a, c);
  o[6] = pk_addconj(a, b);
  o[7] = pk_subconj(a, b);
  o[8] = pk_conj_addmj(a, b);
  o[9] = pk_scale_lo(a, b);
  o[10] = pk_scale_hi(a, b);
  o[11] = pk_mulj<false>(a);
  o[12] = pk_mulj<true>(a);
}

int main() {
  const int N = 64;
  float h_in[4 * N], h_out[32 * N];
  for (int i = 0; i < 4 * N; ++i) h_in[i] = sinf(0.37f * i + 0.1f) * (1 + i % 7);
  float *d_in, *d_out;
  hipMalloc(&d_in, sizeof(h_in)), hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  probe<<<1, N>>>(d_in, d_out);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  double worst = 0;
  int bad = 0;
  for (int t = 0; t < N; ++t) {
    const double ax = h_in[4 * t], ay = h_in[4 * t + 1], bx = h_in[4 * t + 2], by = h_in[4 * t + 3];
    const double cx = 0.92387953251128674, cy = 0.38268343236508977;
    const double ref[13][2] = {{ax + by, ay - bx}, {ax - by, ay + bx}, {ax * bx - ay * by, ax * by + ay * bx},
                               {ax * bx + ay * by, ay * bx - ax * by}, {ax * cx - ay * cy, ax * cy + ay * cx},
                               {ax * cx + ay * cy, ay * cx - ax * cy}, {ax + bx, ay - by}, {ax - bx, ay + by},
                               {ax + by, -ay + bx}, {ax * bx, ay * bx}, {ax * by, ay * by}, {-ay, ax}, {ay, -ax}};
    for (int k = 0; k < 13; ++k)
      for (int c = 0; c < 2; ++c) {
        const double e = fabs(h_out[32 * t + 2 * k + c] - ref[k][c]);
        if (e > worst) worst = e;
        if (e > 1e-5 * (1 + fabs(ref[k][c]))) {
          if (bad++ < 10) printf("op %d lane %d comp %d: got %g want %g\n", k, t, c, h_out[32 * t + 2 * k + c], ref[k][c]);
        }
      }
  }
  printf("pk_probe: worst abs error %.3g, %d mismatches\n", worst, bad);
  return bad != 0;
}
