// Validation metrics on the device (reference: src/model.py:158-160, 215-220 -- MONAI's PSNRMetric(1),
// SSIMMetric(3, data_range=1) and MAEMetric applied to (y_hat, y) and averaged).  f32 NCDHW inputs, one
// value per batch item like MONAI's (B, 1) results; sums leave the device as f64.
//   err_sums : per item sum |a-b| and sum (a-b)^2                       (MAE, and MSE for PSNR)
//   ssim3d   : Gaussian-window SSIM, "valid" windows, separable in three passes: W and H filter the five
//              moment fields (x, y, xx, yy, xy); the D pass filters, applies the SSIM formula and reduces.
// Deterministic: fixed-order block reductions, no atomics.
#include "common.h"

namespace {

constexpr int kMaxWin = 15;
struct Gauss { int n; float g[kMaxWin]; };

__device__ __forceinline__ double block_sum_256(double v, double* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void err_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          long long per_item, double* __restrict__ part) {
  __shared__ double red[4];
  const float* pa = a + (long long)blockIdx.y * per_item;
  const float* pb = b + (long long)blockIdx.y * per_item;
  float s1 = 0.f, s2 = 0.f;
  double d1 = 0.0, d2 = 0.0;
  int n = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_item; i += (long long)gridDim.x * 256) {
    const float d = pa[i] - pb[i];
    s1 += fabsf(d); s2 += d * d;
    if (++n == 64) { d1 += s1; d2 += s2; s1 = 0.f; s2 = 0.f; n = 0; }     // short f32 runs, f64 across them
  }
  d1 += s1; d2 += s2;
  const double t1 = block_sum_256(d1, red);
  const double t2 = block_sum_256(d2, red);
  if (threadIdx.x == 0) {
    double* p = part + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2;
    p[0] = t1; p[1] = t2;
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ part, int nparts, int width,
                                                           double* __restrict__ out, double scale) {
  __shared__ double red[4];
  for (int j = 0; j < width; ++j) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[((long long)blockIdx.x * nparts + i) * width + j];
    const double t = block_sum_256(s, red);
    if (threadIdx.x == 0) out[(long long)blockIdx.x * width + j] = t * scale;
  }
}

// The window length is a template parameter (WIN = 11 for MONAI's default, 0 = runtime): with a compile-time trip
// count the tap loops unroll and their loads go out together (the runtime loops had one load in flight per lane).
// pass W: rows = items*C*D*H rows of W floats -> 5 fields of Wo = W - n + 1
template <int WIN>
__global__ __launch_bounds__(256) void ssim_pass_w_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          float* __restrict__ out, long long rows, int w, int wo, Gauss G) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * wo) return;
  const long long row = i / wo; const int o = (int)(i - row * wo);
  const float* px = x + row * w + o; const float* py = y + row * w + o;
  float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
  const int n = WIN ? WIN : G.n;
#pragma unroll
  for (int k = 0; k < n; ++k) {
    const float a = px[k], b = py[k], g = G.g[k];
    sx += g * a; sy += g * b; sxx += g * (a * a); syy += g * (b * b); sxy += g * (a * b);
  }
  const long long fs = rows * wo;
  out[i] = sx; out[fs + i] = sy; out[2 * fs + i] = sxx; out[3 * fs + i] = syy; out[4 * fs + i] = sxy;
}

// pass H: planes = 5*items*C*D planes of [h][wo] -> [ho][wo]
template <int WIN>
__global__ __launch_bounds__(256) void ssim_pass_h_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          long long planes, int h, int ho, int wo, Gauss G) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)ho * wo;
  if (i >= planes * per) return;
  const long long pl = i / per; const long long r = i - pl * per;
  const int oh = (int)(r / wo), ow = (int)(r - (long long)oh * wo);
  const float* p = in + (pl * h + oh) * wo + ow;
  float s = 0.f;
  const int n = WIN ? WIN : G.n;
#pragma unroll
  for (int k = 0; k < n; ++k) s += G.g[k] * p[(long long)k * wo];
  out[i] = s;
}

// pass D + SSIM formula + per-block sum.  grid (blocks, items*C); in: [5][items*C][d][ho][wo]
template <int WIN>
__global__ __launch_bounds__(256) void ssim_pass_d_kernel(const float* __restrict__ in, double* __restrict__ part,
                                                          long long nvol, int d, int dd, int ho, int wo, float c1, float c2, Gauss G) {
  __shared__ double red[4];
  const long long plane = (long long)ho * wo, per = (long long)dd * plane;
  const long long fs = nvol * d * plane;                       // field stride
  const float* base = in + (long long)blockIdx.y * d * plane;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long long)gridDim.x * 256) {
    const float* p = base + i;                                 // (od, oh, ow) flattened == offset of the first tap
    float m[5];
    const int n = WIN ? WIN : G.n;
#pragma unroll
    for (int f = 0; f < 5; ++f) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < n; ++k) s += G.g[k] * p[f * fs + (long long)k * plane];
      m[f] = s;
    }
    const float sx = m[2] - m[0] * m[0], sy = m[3] - m[1] * m[1], sxy = m[4] - m[0] * m[1];
    const float cs = (2.f * sxy + c2) / (sx + sy + c2);
    acc += (double)(((2.f * m[0] * m[1] + c1) / (m[0] * m[0] + m[1] * m[1] + c1)) * cs);
  }
  const double t = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[(long long)blockIdx.y * gridDim.x + blockIdx.x] = t;
}

int blocks_for(long long n) { long long b = (n + 256 * 16 - 1) / (256 * 16); return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b)); }

}  // namespace

extern "C" int32_t mi355_err_blocks(int64_t per_item) { return blocks_for(per_item); }

extern "C" int mi355_err_sums(const float* a, const float* b, int64_t per_item, int32_t items, double* partials,
                              double* out, void* stream) {
  MI355_REQUIRE(a && b && partials && out && per_item > 0 && items > 0 && items <= 65535, "err_sums: bad argument");
  const int nb = blocks_for(per_item);
  hipStream_t st = (hipStream_t)stream;
  err_partial_kernel<<<dim3(nb, items), 256, 0, st>>>(a, b, per_item, partials);
  sum_partials_kernel<<<items, 256, 0, st>>>(partials, nb, 2, out, 1.0);
  return mi355_check_launch("err_sums");
}

extern "C" int64_t mi355_ssim3d_workspace_bytes(int32_t items, int32_t c, int32_t d, int32_t h, int32_t w, int32_t win) {
  if (items <= 0 || c <= 0 || win < 1 || win > kMaxWin || d < win || h < win || w < win) return -1;
  const long long nvol = (long long)items * c, wo = w - win + 1, ho = h - win + 1, dd = d - win + 1;
  const long long f1 = 5 * nvol * d * h * wo, f2 = 5 * nvol * d * ho * wo;
  const long long parts = nvol * blocks_for(dd * ho * wo);
  return (f1 + f2) * 4 + parts * 8 + 256;
}

extern "C" int mi355_ssim3d(const float* x, const float* y, int32_t items, int32_t c, int32_t d, int32_t h, int32_t w,
                            int32_t win, const float* window, float c1, float c2, void* workspace,
                            int64_t workspace_bytes, double* out, void* stream) {
  const long long need = mi355_ssim3d_workspace_bytes(items, c, d, h, w, win);
  MI355_REQUIRE(need > 0, "ssim3d: bad shape (items=%d c=%d d=%d h=%d w=%d win=%d)", items, c, d, h, w, win);
  MI355_REQUIRE(x && y && window && workspace && out && workspace_bytes >= need, "ssim3d: null pointer or workspace too small");
  MI355_REQUIRE((long long)items * c <= 65535, "ssim3d: too many channel volumes");
  Gauss G; G.n = win;
  for (int i = 0; i < kMaxWin; ++i) G.g[i] = i < win ? window[i] : 0.f;
  const long long nvol = (long long)items * c, wo = w - win + 1, ho = h - win + 1, dd = d - win + 1;
  float* f1 = (float*)workspace;
  float* f2 = f1 + 5 * nvol * d * h * wo;
  double* part = (double*)(((uintptr_t)(f2 + 5 * nvol * d * ho * wo) + 255) & ~(uintptr_t)255);
  hipStream_t st = (hipStream_t)stream;
  const long long rows = nvol * d * h;
  const long long planes = 5 * nvol * d;
  const int nb = blocks_for(dd * ho * wo);
  const unsigned gw = (unsigned)((rows * wo + 255) / 256), gh = (unsigned)((planes * ho * wo + 255) / 256);
  if (win == 11) {
    ssim_pass_w_kernel<11><<<gw, 256, 0, st>>>(x, y, f1, rows, w, (int)wo, G);
    ssim_pass_h_kernel<11><<<gh, 256, 0, st>>>(f1, f2, planes, h, (int)ho, (int)wo, G);
    ssim_pass_d_kernel<11><<<dim3(nb, (unsigned)nvol), 256, 0, st>>>(f2, part, nvol, d, (int)dd, (int)ho, (int)wo, c1, c2, G);
  } else {
    ssim_pass_w_kernel<0><<<gw, 256, 0, st>>>(x, y, f1, rows, w, (int)wo, G);
    ssim_pass_h_kernel<0><<<gh, 256, 0, st>>>(f1, f2, planes, h, (int)ho, (int)wo, G);
    ssim_pass_d_kernel<0><<<dim3(nb, (unsigned)nvol), 256, 0, st>>>(f2, part, nvol, d, (int)dd, (int)ho, (int)wo, c1, c2, G);
  }
  sum_partials_kernel<<<items, 256, 0, st>>>(part, c * nb, 1, out, 1.0 / ((double)c * dd * ho * wo));
  return mi355_check_launch("ssim3d");
}
