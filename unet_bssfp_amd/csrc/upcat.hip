// The up-branch of MONAI's UpCat (src/model.py:22-28: BasicUNet(upsample="deconv")) without materialising the up-sampled tensor.
//
//   up = ConvTranspose3d(Cl -> Cu, k2, s2)(x_low) + b_d;   z = Conv3d(Ce + Cu -> Co, k3, p1)(cat([x_e, up], 1)) + b_c
//
// For even extents the transposed convolution followed by the 3x3x3 convolution of its branch is ONE transposed convolution
// with a 4x4x4 kernel, stride 2, padding 1, applied to the LOW-resolution tensor (the composition of two linear maps):
//
//   z_up[o][co] = sum_{m, ci} x_low[m][ci] K4[ci][co][t],  t = o - 2 m + 1 in [0, 3] per axis,
//   K4[ci][co][t] = sum_cu sum_{(a, k): a - k + 2 = t per axis} W_d[ci][cu][a] W_c[co][Ce + cu][k]
//
// -- 8 taps per output voxel and input channel instead of 27 (plus the transposed convolution itself): 2 * 8 * Cl * Co FLOP per
// output voxel against 2 * (27 * Cu * Co + Cl * Cu): 3.6x fewer for upcat_1 (Cl = Cu = 64, Co = 32), and `up` (268 MB in bf16
// at 128^3) is neither written nor read, forward or backward.  K4 is the weight of a virtual Conv3d(Co -> Cl, k4, s2, p1)
// ("Kconv"): z_up is its transpose applied to x_low, the gradient of x_low is Kconv applied to dz, and dK4 is Kconv's
// weight gradient for input dz and output gradient x_low -- the PatchGAN's machinery.
//
// Not a pure reshuffle at the volume border: the 3x3x3 convolution zero-pads `up`, whose bias b_d therefore does not reach
// taps that fall outside -- z = ... + sum_{k: o + k - 1 inside} T[co][k], T[co][k] = sum_cu W_c[co][Ce + cu][k] b_d[cu].  The
// interior value sum_k T goes into the bias vector the consumer adds (`biasp`), the 26 border classes get `delta[cls]` added
// to their accumulators (conv_march2_kernel, d2s mode); backward, b_d and W_c receive the matching terms from the sums of dz
// over the 26 border regions (mi355_border_sums).  Everything here is small (weights only), exact f32 arithmetic.
#include "common.h"

namespace {

// index helpers: a in {0,1}^3 (4 ad + 2 ah + aw), k in {0,1,2}^3 (9 kd + 3 kh + kw), t in {0..3}^3 (16 td + 4 th + tw)
__device__ __forceinline__ int t_of(int a, int k) {
  const int td = (a >> 2) - k / 9 + 2, th = ((a >> 1) & 1) - (k / 3) % 3 + 2, tw = (a & 1) - k % 3 + 2;
  return td * 16 + th * 4 + tw;
}

// K4 and its packing for conv_march2_kernel's d2s mode: one thread per (ci, co, t)
__global__ __launch_bounds__(256) void upcat_compose_kernel(const float* __restrict__ wd, const float* __restrict__ wc, int cl, int cu, int ce, int co,
                                                            float* __restrict__ k4, bf16_t* __restrict__ wp) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= cl * co * 64) return;
  const int t = idx & 63, o = (idx >> 6) % co, ci = idx / (64 * co);
  const int td = t >> 4, th = (t >> 2) & 3, tw = t & 3;
  const float* wdp = wd + (long long)ci * cu * 8;
  const float* wcp = wc + ((long long)o * (ce + cu) + ce) * 27;
  float s = 0.f;
  // (a, k) per axis with a - k + 2 = t: k = a + 2 - t in [0, 2]
  for (int u = 0; u < cu; ++u) {
    float su = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int kd = (a >> 2) + 2 - td, kh = ((a >> 1) & 1) + 2 - th, kw = (a & 1) + 2 - tw;
      if (kd >= 0 && kd <= 2 && kh >= 0 && kh <= 2 && kw >= 0 && kw <= 2) su += wdp[u * 8 + a] * wcp[u * 27 + kd * 9 + kh * 3 + kw];
    }
    s += su;
  }
  k4[idx] = s;
  if (wp) {
    // d2s packing: output class b (voxel 2 j + b) and input tap e (cell j - (1 - b) + e) <-> t = 3 - b - 2 e per axis
    const int bd = (3 - td) & 1, ed = (3 - td) >> 1, bh = (3 - th) & 1, eh = (3 - th) >> 1, bw = (3 - tw) & 1, ew = (3 - tw) >> 1;
    const int blk = bd * 4 + bh * 2 + bw, e = ed * 4 + eh * 2 + ew, coutp = 8 * co;
    wp[(((long long)(ci >> 4) * 8 + e) * coutp + blk * co + o) * 16 + (ci & 15)].v = f32_to_bf16_bits(s);
  }
}

// T[co][k], the bias vector the consumer adds and the 27 border-class corrections: one workgroup
__global__ __launch_bounds__(256) void upcat_bias_kernel(const float* __restrict__ wc, const float* __restrict__ bd, const float* __restrict__ bc,
                                                         int cu, int ce, int co, float* __restrict__ biasp, float* __restrict__ delta) {
  extern __shared__ float tk[];                        // [co][27]
  for (int i = threadIdx.x; i < co * 27; i += 256) {
    const int o = i / 27, k = i - o * 27;
    const float* wcp = wc + ((long long)o * (ce + cu) + ce) * 27 + k;
    float s = 0.f;
    for (int u = 0; u < cu; ++u) s += wcp[u * 27] * bd[u];
    tk[i] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 28 * co; i += 256) {
    const int cls = i / co, o = i - cls * co;
    if (cls == 27) {
      float s = bc ? bc[o] : 0.f;
      for (int k = 0; k < 27; ++k) s += tk[o * 27 + k];
      biasp[o] = s;
    } else {
      // class (cd, ch, cw) in {0 first voxel, 1 interior, 2 last voxel}: the first voxel lacks tap k = 0, the last one k = 2
      const int cd = cls / 9, ch = (cls / 3) % 3, cw = cls % 3;
      float s = 0.f;
      for (int k = 0; k < 27; ++k) {
        const int kd = k / 9, kh = (k / 3) % 3, kw = k % 3;
        const bool out = (cd == 0 && kd == 0) || (cd == 2 && kd == 2) || (ch == 0 && kh == 0) || (ch == 2 && kh == 2) ||
                         (cw == 0 && kw == 0) || (cw == 2 && kw == 2);
        if (out) s -= tk[o * 27 + k];
      }
      delta[cls * co + o] = s;
    }
  }
}

// chain rule from dK4 (and the border sums of dz) to dW_d, dW_c[:, Ce:], db_d
__global__ __launch_bounds__(256) void upcat_chain_kernel(const float* __restrict__ dk4, const float* __restrict__ wd, const float* __restrict__ wc,
                                                          const float* __restrict__ bd, const float* __restrict__ esum, int cl, int cu, int ce, int co,
                                                          float* __restrict__ dwd, float* __restrict__ dwc, float* __restrict__ dbd, int accumulate) {
  extern __shared__ float gk[];                        // G_k[co] = sum of dz over the voxels o with o + k - 1 inside: [27][co]
  for (int i = threadIdx.x; i < 27 * co; i += 256) {
    const int k = i / co, o = i - k * co;
    const int kk[3] = {k / 9, (k / 3) % 3, k % 3};
    // chi_k per axis: k = 0 -> all - first, k = 1 -> all, k = 2 -> all - last; region index s: 0 all, 1 first, 2 last
    float s = 0.f;
    for (int sub = 0; sub < 8; ++sub) {
      int sd = 0, sh = 0, sw = 0, sign = 1;
      bool ok = true;
      const int pick[3] = {(sub >> 2) & 1, (sub >> 1) & 1, sub & 1};
      int* ss[3] = {&sd, &sh, &sw};
      for (int ax = 0; ax < 3; ++ax) {
        if (pick[ax]) {
          if (kk[ax] == 1) { ok = false; break; }
          *ss[ax] = kk[ax] == 0 ? 1 : 2;
          sign = -sign;
        }
      }
      if (!ok || (sd == 0 && sh == 0 && sw == 0)) continue;     // E[all][all][all] = sum of dz over the volume: exactly zero behind a
      s += sign * esum[((sd * 3 + sh) * 3 + sw) * co + o];      // normalisation (rounding noise in any evaluation): taken as zero
    }
    gk[i] = s;
  }
  __syncthreads();
  const int n_wd = cl * cu * 8, n_wc = co * cu * 27;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < n_wd) {
    const int a = idx & 7, u = (idx >> 3) % cu, ci = idx / (8 * cu);
    float s = 0.f;
    for (int o = 0; o < co; ++o) {
      const float* dk = dk4 + ((long long)ci * co + o) * 64;
      const float* wcp = wc + ((long long)o * (ce + cu) + ce + u) * 27;
#pragma unroll
      for (int k = 0; k < 27; ++k) s += dk[t_of(a, k)] * wcp[k];
    }
    if (accumulate) dwd[idx] += s; else dwd[idx] = s;
  } else if (idx < n_wd + n_wc) {
    const int j = idx - n_wd, k = j % 27, u = (j / 27) % cu, o = j / (27 * cu);
    float s = bd ? bd[u] * gk[k * co + o] : 0.f;
    for (int ci = 0; ci < cl; ++ci) {
      const float* dk = dk4 + ((long long)ci * co + o) * 64;
      const float* wdp = wd + ((long long)ci * cu + u) * 8;
#pragma unroll
      for (int a = 0; a < 8; ++a) s += dk[t_of(a, k)] * wdp[a];
    }
    float* dst = dwc + ((long long)o * (ce + cu) + ce + u) * 27 + k;
    if (accumulate) *dst += s; else *dst = s;
  } else if (idx < n_wd + n_wc + cu && dbd) {
    const int u = idx - n_wd - n_wc;
    float s = 0.f;
    for (int o = 0; o < co; ++o) {
      const float* wcp = wc + ((long long)o * (ce + cu) + ce + u) * 27;
      for (int k = 0; k < 27; ++k) s += wcp[k] * gk[k * co + o];
    }
    if (accumulate) dbd[u] += s; else dbd[u] = s;
  }
}

// ---- sums of a gradient over the 26 border regions of the volume (region = per axis: all voxels / the first one / the last one;
//      at least one axis restricted).  Stage 1: one workgroup per (n, d) plane -> q[plane][sh][sw][c]; boundary planes sum all
//      their voxels, interior planes only their border rows / columns.  Stage 2: planes -> e[sd][sh][sw][c].  Fixed orders.
template <typename T>
__global__ __launch_bounds__(256) void border_plane_kernel(const T* __restrict__ g, int ld, int d_ext, int h_ext, int w_ext, int c, float* __restrict__ q) {
  constexpr int EPV = Elem<T>::kPer16B;
  __shared__ float red[4][9][8 * 8];                   // [wave][combo][piece * EPV + j]  (c <= 64: up to 16 pieces of 4 / 8 of 8)
  const int plane = blockIdx.x, d = plane % d_ext;
  const bool bplane = d == 0 || d == d_ext - 1;
  const int pieces = c / EPV, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int piece = tid % pieces, vlane = tid / pieces, vstep = 256 / pieces;
  float acc[9][EPV];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < EPV; ++j) acc[k][j] = 0.f;
  const T* base = g + (long long)plane * h_ext * w_ext * ld + piece * EPV;
  // candidate voxels: boundary plane -> all; interior plane -> rows 0 / H - 1 in full, columns 0 / W - 1 of the other rows
  const int nfull = bplane ? h_ext * w_ext : 2 * w_ext + 2 * (h_ext - 2);
  for (int i = vlane; i < nfull; i += vstep) {
    int hh, ww;
    if (bplane) { hh = i / w_ext; ww = i - hh * w_ext; }
    else if (i < 2 * w_ext) { hh = i < w_ext ? 0 : h_ext - 1; ww = i < w_ext ? i : i - w_ext; }
    else { const int j = i - 2 * w_ext; hh = 1 + (j >> 1); ww = (j & 1) ? w_ext - 1 : 0; }
    Vec16<T> v;
    v.load(base + ((long long)hh * w_ext + ww) * ld);
    const int hs = hh == 0 ? 1 : (hh == h_ext - 1 ? 2 : 0), ws = ww == 0 ? 1 : (ww == w_ext - 1 ? 2 : 0);
#pragma unroll
    for (int sh = 0; sh < 3; ++sh)
#pragma unroll
      for (int sw = 0; sw < 3; ++sw) {
        const bool in = (sh == 0 || sh == hs) && (sw == 0 || sw == ws) && (bplane || sh || sw);
        if (in) {
#pragma unroll
          for (int j = 0; j < EPV; ++j) acc[sh * 3 + sw][j] += v.f[j];
        }
      }
  }
  // lanes of equal piece within a wave (pieces divides 64), then the four waves
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      float s = acc[k][j];
      for (int o = pieces; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
      if (lane < pieces) red[wave][k][lane * EPV + j] = s;
    }
  __syncthreads();
  for (int i = tid; i < 9 * c; i += 256) {
    const int k = i / c, ch = i - k * c;
    q[((long long)plane * 9 + k) * c + ch] = (red[0][k][ch] + red[1][k][ch]) + (red[2][k][ch] + red[3][k][ch]);
  }
}

__global__ __launch_bounds__(256) void border_final_kernel(const float* __restrict__ q, int nplanes, int d_ext, int c, float* __restrict__ e) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 27 * c) return;
  const int ch = i % c, k9 = (i / c) % 9, sd = i / (9 * c);
  float s = 0.f;
  if (!(sd == 0 && k9 == 0))
    for (int p = 0; p < nplanes; ++p) {
      const int d = p % d_ext;
      if (sd == 0 || (sd == 1 && d == 0) || (sd == 2 && d == d_ext - 1)) s += q[((long long)p * 9 + k9) * c + ch];
    }
  e[i] = s;
}

}  // namespace

extern "C" {

int mi355_upcat_compose(const float* wd, const float* wc, const float* bd, const float* bc, int32_t cl, int32_t cu, int32_t ce,
                        int32_t co, float* k4, void* wp_d2s, float* biasp, float* delta, void* stream) {
  MI355_REQUIRE(wd && wc && k4 && cl > 0 && cu > 0 && ce >= 0 && co > 0, "upcat_compose: bad argument");
  MI355_REQUIRE(!wp_d2s || (cl % 16 == 0 && co % 32 == 0), "upcat_compose: the d2s packing needs cl %% 16 == 0 and co %% 32 == 0");
  MI355_REQUIRE(co * 27 * 4 <= 48 * 1024, "upcat_compose: too many output channels");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)cl * co * 64;
  upcat_compose_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(wd, wc, cl, cu, ce, co, k4, (bf16_t*)wp_d2s);
  int rc = mi355_check_launch("upcat_compose");
  if (rc) return rc;
  if (biasp && delta) {
    MI355_REQUIRE(bd, "upcat_compose: the bias tables need b_d");
    upcat_bias_kernel<<<dim3(1), dim3(256), co * 27 * sizeof(float), st>>>(wc, bd, bc, cu, ce, co, biasp, delta);
    rc = mi355_check_launch("upcat_bias");
  }
  return rc;
}

int mi355_upcat_chain(const float* dk4, const float* wd, const float* wc, const float* bd, const float* esum, int32_t cl, int32_t cu,
                      int32_t ce, int32_t co, float* dwd, float* dwc, float* dbd, int32_t accumulate, void* stream) {
  MI355_REQUIRE(dk4 && wd && wc && esum && dwd && dwc && cl > 0 && cu > 0 && ce >= 0 && co > 0, "upcat_chain: bad argument");
  MI355_REQUIRE(co * 27 * 4 <= 48 * 1024, "upcat_chain: too many output channels");
  const long long total = (long long)cl * cu * 8 + (long long)co * cu * 27 + cu;
  upcat_chain_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), co * 27 * sizeof(float), (hipStream_t)stream>>>(
      dk4, wd, wc, bd, esum, cl, cu, ce, co, dwd, dwc, dbd, accumulate);
  return mi355_check_launch("upcat_chain");
}

int64_t mi355_border_sums_workspace(int32_t n, int32_t d, int32_t c) { return (int64_t)n * d * 9 * c * 4; }

int mi355_border_sums(const void* g, int32_t ld, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c, int32_t dtype, float* workspace,
                      float* e, void* stream) {
  MI355_REQUIRE(g && workspace && e && n > 0 && d >= 2 && h >= 2 && w >= 2, "border_sums: bad argument");
  MI355_REQUIRE(dtype == MI355_DT_F32 || dtype == MI355_DT_BF16, "border_sums: bad dtype");
  const int epv = dtype == MI355_DT_F32 ? 4 : 8;
  MI355_REQUIRE(c > 0 && c <= 64 && c % epv == 0 && ld % epv == 0 && 64 % (c / epv) == 0, "border_sums: channels must be 8, 16, 32 or 64 (f32: 4 .. 64)");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MI355_DT_F32) border_plane_kernel<float><<<dim3((unsigned)(n * d)), dim3(256), 0, st>>>((const float*)g, ld, d, h, w, c, workspace);
  else border_plane_kernel<bf16_t><<<dim3((unsigned)(n * d)), dim3(256), 0, st>>>((const bf16_t*)g, ld, d, h, w, c, workspace);
  int rc = mi355_check_launch("border_sums");
  if (rc) return rc;
  border_final_kernel<<<dim3((unsigned)((27 * c + 255) / 256)), dim3(256), 0, st>>>(workspace, n * d, d, c, e);
  return mi355_check_launch("border_sums_final");
}

}  // extern "C"
