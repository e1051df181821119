// GPU-side versions of the cheap intensity augmentations of the reference's training transform
// (src/data_module.py:130-139: tio.RandomBiasField, tio.RandomNoise, tio.RandomGamma), f32 NCDHW / (C, D, H, W)
// tensors, one pass each (read + write: HBM-bound).  The random PARAMETERS are drawn on the host like TorchIO
// does; the per-voxel noise comes from a counter-based hash (no RNG state, hipGraph-safe).
#include "common.h"

namespace {

struct BiasArgs { int c, d, h, w, order, ncoef; float coef[35]; };   // order <= 4: 35 coefficients

// field(a, b, c) = sum_i coef_i * u0[a]^xo * u1[b]^yo * u2[c]^zo over xo + yo + zo <= order, loops in TorchIO's
// order (x outer, then y, then z); u = (index - n/2 + 0.5) / max|.| per axis; out = x * exp(field)
__global__ __launch_bounds__(256) void bias_field_kernel(const float* __restrict__ x, float* __restrict__ out, BiasArgs q) {
  const long long vol = (long long)q.d * q.h * q.w;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= vol) return;
  const int c0 = (int)(i % q.w), b0 = (int)(i / q.w % q.h), a0 = (int)(i / ((long long)q.w * q.h));
  // TorchIO: np.arange(-n/2, n/2) + 0.5 per axis, divided by its maximum n/2 - 0.5 (when positive)
  auto coord = [](int idx, int n) {
    const float half = 0.5f * (float)n, v = (float)idx - half + 0.5f, mx = half - 0.5f;
    return mx > 0.f ? v / mx : v;
  };
  const float u0 = coord(a0, q.d), u1 = coord(b0, q.h), u2 = coord(c0, q.w);
  float p0[5], p1[5], p2[5];
  p0[0] = p1[0] = p2[0] = 1.f;
#pragma unroll
  for (int k = 1; k < 5; ++k) { p0[k] = p0[k - 1] * u0; p1[k] = p1[k - 1] * u1; p2[k] = p2[k - 1] * u2; }
  float f = 0.f;
  int n = 0;
  for (int xo = 0; xo <= q.order; ++xo)
    for (int yo = 0; yo <= q.order - xo; ++yo)
      for (int zo = 0; zo <= q.order - xo - yo; ++zo) f += q.coef[n++] * p0[xo] * p1[yo] * p2[zo];
  const float g = expf(f);
  for (int ch = 0; ch < q.c; ++ch) out[ch * vol + i] = x[ch * vol + i] * g;
}

__global__ __launch_bounds__(256) void gamma_kernel(const float* __restrict__ x, float* __restrict__ out, long long n, float gamma) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  out[i] = copysignf(powf(fabsf(v), gamma), v);                          // TorchIO keeps the sign of negative intensities
}

__global__ __launch_bounds__(256) void noise_kernel(const float* __restrict__ x, float* __restrict__ out, long long n, float mean,
                                                    float std, unsigned long long seed) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long h = ((unsigned long long)i + 1ull) * 0x9E3779B97F4A7C15ull + seed;
  h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32; h *= 0xD6E8FEB86659FD93ull; h ^= h >> 32;
  const float u1 = ((float)(unsigned)(h >> 40) + 0.5f) * (1.f / 16777216.f);      // (0, 1)
  const float u2 = ((float)(unsigned)((h >> 16) & 0xffffffu) + 0.5f) * (1.f / 16777216.f);
  const float z = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);   // Box-Muller
  out[i] = x[i] + mean + std * z;
}

}  // namespace

extern "C" int mi355_aug_bias_field(const float* x, float* out, int32_t c, int32_t d, int32_t h, int32_t w,
                                    const float* coefficients, int32_t order, void* stream) {
  MI355_REQUIRE(x && out && coefficients && c > 0 && d > 0 && h > 0 && w > 0, "aug_bias_field: bad argument");
  MI355_REQUIRE(order >= 0 && order <= 4, "aug_bias_field: order must be 0..4");
  BiasArgs q{c, d, h, w, order, (order + 1) * (order + 2) * (order + 3) / 6, {}};
  for (int i = 0; i < q.ncoef; ++i) q.coef[i] = coefficients[i];
  const long long vol = (long long)d * h * w;
  bias_field_kernel<<<(unsigned)((vol + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, out, q);
  return mi355_check_launch("aug_bias_field");
}

extern "C" int mi355_aug_gamma(const float* x, float* out, int64_t count, float gamma, void* stream) {
  MI355_REQUIRE(x && out && count > 0, "aug_gamma: bad argument");
  gamma_kernel<<<(unsigned)((count + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, out, count, gamma);
  return mi355_check_launch("aug_gamma");
}

extern "C" int mi355_aug_noise(const float* x, float* out, int64_t count, float mean, float std, uint64_t seed, void* stream) {
  MI355_REQUIRE(x && out && count > 0 && std >= 0.f, "aug_noise: bad argument");
  noise_kernel<<<(unsigned)((count + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, out, count, mean, std, seed);
  return mi355_check_launch("aug_noise");
}
