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

// K4 and its packing for conv_march2_kernel's d2s mode.  Workgroup = (output channel co, four input channels ci: one per wave);
// W_c[co][Ce:][27] and the four W_d[ci] slices are staged in LDS by coalesced loads (a first version that read W_d from global
// memory inside the u loop ran at one L2 round trip per iteration: 440 - 860 us); lane = tap t.  The workgroups with
// blockIdx.y == 0 also produce T[co][k], the bias vector the consumer adds and the 27 border-class corrections of their channel.
__global__ __launch_bounds__(256) void upcat_compose_kernel(const float* __restrict__ wd, const float* __restrict__ wc, const float* __restrict__ bd,
                                                            const float* __restrict__ bc, int cl, int cu, int ce, int co,
                                                            float* __restrict__ k4, bf16_t* __restrict__ wp, float* __restrict__ biasp,
                                                            float* __restrict__ delta) {
  extern __shared__ float sm[];                        // wcs[cu][27], wds[4][cu][8], tk[27]
  float* wcs = sm;
  float* wds = sm + cu * 27;
  float* tk = wds + 4 * cu * 8;
  const int o = blockIdx.x, ci0 = blockIdx.y * 4, tid = threadIdx.x, t = tid & 63, wave = tid >> 6;
  const float* wcp = wc + ((long long)o * (ce + cu) + ce) * 27;
  for (int i = tid; i < cu * 27; i += 256) wcs[i] = wcp[i];
  for (int i = tid; i < 4 * cu * 8; i += 256) wds[i] = (ci0 + i / (cu * 8)) < cl ? wd[(long long)ci0 * cu * 8 + i] : 0.f;
  __syncthreads();
  const int ci = ci0 + wave;
  if (ci < cl) {
    const int td = t >> 4, th = (t >> 2) & 3, tw = t & 3;
    // (a, k) per axis with a - k + 2 = t: k = a + 2 - t in [0, 2]
    int koff[8]; float kok[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      const int kd = (a >> 2) + 2 - td, kh = ((a >> 1) & 1) + 2 - th, kw = (a & 1) + 2 - tw;
      const bool ok = kd >= 0 && kd <= 2 && kh >= 0 && kh <= 2 && kw >= 0 && kw <= 2;
      kok[a] = ok ? 1.f : 0.f;
      koff[a] = ok ? kd * 9 + kh * 3 + kw : 0;
    }
    const float* wv = wds + wave * cu * 8;
    float s = 0.f;
    for (int u = 0; u < cu; ++u) {
      float su = 0.f;
#pragma unroll
      for (int a = 0; a < 8; ++a) su += kok[a] * wv[u * 8 + a] * wcs[u * 27 + koff[a]];
      s += su;
    }
    k4[((long long)ci * co + o) * 64 + t] = s;
    if (wp) {
      // d2s packing: output class b (voxel 2 j + b) and input tap e (cell j - (1 - b) + e) <-> t = 3 - b - 2 e per axis
      const int bd_ = (3 - td) & 1, ed = (3 - td) >> 1, bh = (3 - th) & 1, eh = (3 - th) >> 1, bw = (3 - tw) & 1, ew = (3 - tw) >> 1;
      const int blk = bd_ * 4 + bh * 2 + bw, e = ed * 4 + eh * 2 + ew, coutp = 8 * co;
      wp[(((long long)(ci >> 4) * 8 + e) * coutp + blk * co + o) * 16 + (ci & 15)].v = f32_to_bf16_bits(s);
    }
  }
  if (biasp == nullptr || blockIdx.y != 0) return;
  if (tid < 27) {
    float sv = 0.f;
    for (int u = 0; u < cu; ++u) sv += wcs[u * 27 + tid] * bd[u];
    tk[tid] = sv;
  }
  __syncthreads();
  if (tid < 28) {
    if (tid == 27) {
      float sv = bc ? bc[o] : 0.f;
      for (int k = 0; k < 27; ++k) sv += tk[k];
      biasp[o] = sv;
    } else {
      // class (cd, ch, cw) in {0 first voxel, 1 interior, 2 last voxel}: the first voxel lacks tap k = 0, the last one k = 2
      const int cd = tid / 9, ch = (tid / 3) % 3, cw = tid % 3;
      float sv = 0.f;
      for (int k = 0; k < 27; ++k) {
        const int kd = k / 9, kh = (k / 3) % 3, kw = k % 3;
        const bool out = (cd == 0 && kd == 0) || (cd == 2 && kd == 2) || (ch == 0 && kh == 0) || (ch == 2 && kh == 2) ||
                         (cw == 0 && kw == 0) || (cw == 2 && kw == 2);
        if (out) sv -= tk[k];
      }
      delta[tid * co + o] = sv;
    }
  }
}

// chain rule from dK4 (and the border sums of dz) to dW_d, dW_c[:, Ce:], db_d.
// G_k[co] = sum of dz over the voxels o with o + k - 1 inside, from the border-region sums (e[0][0][0], the volume sum, is zero
// behind a normalisation -- rounding noise in any evaluation -- and is taken as zero)
__device__ __forceinline__ float upcat_gk(const float* __restrict__ esum, int co, int k, int o) {
  const int kk[3] = {k / 9, (k / 3) % 3, k % 3};
  float s = 0.f;
  for (int sub = 1; sub < 8; ++sub) {
    int sr[3] = {0, 0, 0}, sign = 1;
    bool ok = true;
    for (int ax = 0; ax < 3; ++ax)
      if ((sub >> (2 - ax)) & 1) {
        if (kk[ax] == 1) { ok = false; break; }          // chi_k per axis: k = 0 -> all - first, k = 1 -> all, k = 2 -> all - last
        sr[ax] = kk[ax] == 0 ? 1 : 2;
        sign = -sign;
      }
    if (ok) s += sign * esum[((sr[0] * 3 + sr[1]) * 3 + sr[2]) * co + o];
  }
  return s;
}

// dW_d[ci][u][a] = sum_{o, k} dK4[ci][o][t(a, k)] W_c[o][Ce + u][k].  One WAVE per (ci, u) -- cl * cu small workgroups fill the chip and
// hide each other's load latency (a 128-workgroup version with LDS staging ran at one memory round trip per 4 o: 64 - 120 us):
// lane = (a, one of 8 o-slices); the 27 W_c values of an o are wave-uniform per slice, the dK4 row of (ci, o) is one 256-byte line.
__global__ __launch_bounds__(64) void upcat_chain_wd_kernel(const float* __restrict__ dk4, const float* __restrict__ wc, int cu, int ce, int co,
                                                            float* __restrict__ dwd, int accumulate) {
  const int ci = blockIdx.x, u = blockIdx.y, lane = threadIdx.x, a = lane & 7, os = lane >> 3;
  int tt[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) tt[k] = t_of(a, k);
  float acc = 0.f;
  for (int o = os; o < co; o += 8) {
    const float* d = dk4 + ((long long)ci * co + o) * 64;
    const float* w = wc + ((long long)o * (ce + cu) + ce + u) * 27;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k) s += d[tt[k]] * w[k];
    acc += s;
  }
  // the 8 o-slices of an output, in a fixed order
  acc += __shfl_xor(acc, 8, 64);
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 32, 64);
  if (os == 0) {
    float* dst = dwd + ((long long)ci * cu + u) * 8 + a;
    if (accumulate) *dst += acc; else *dst = acc;
  }
}

// dW_c[o][Ce + u][k] = sum_{ci, a} dK4[ci][o][t(a, k)] W_d[ci][u][a] + b_d[u] G_k[o].  Workgroup = (o, 8 channels u): thread = (u, k) (216 of
// 256); dK4[ci chunk][o][64] and W_d[ci chunk][u tile][8] are staged 16 ci at a time.  The workgroups with blockIdx.x == co compute
// db_d[u] = sum_{o, k} W_c[o][Ce + u][k] G_k[o] for their 8 channels.
constexpr int kChainCB = 16;
__global__ __launch_bounds__(256) void upcat_chain_wc_kernel(const float* __restrict__ dk4, const float* __restrict__ wd, const float* __restrict__ wc,
                                                             const float* __restrict__ bd, const float* __restrict__ esum, int cl, int cu, int ce,
                                                             int co, float* __restrict__ dwc, float* __restrict__ dbd, int accumulate) {
  extern __shared__ float sm[];                        // gk[27][co], dk[kChainCB][64], wds[kChainCB][8][8]
  float* gk = sm;
  float* dk = sm + 27 * co;
  float* wds = dk + kChainCB * 64;
  const int tid = threadIdx.x, u0 = blockIdx.y * 8;
  for (int i = tid; i < 27 * co; i += 256) gk[i] = upcat_gk(esum, co, i / co, i % co);
  __syncthreads();
  if ((int)blockIdx.x == co) {
    if (dbd && tid < 8 && u0 + tid < cu) {
      float s = 0.f;
      for (int o = 0; o < co; ++o) {
        const float* wcp = wc + ((long long)o * (ce + cu) + ce + u0 + tid) * 27;
        for (int k = 0; k < 27; ++k) s += wcp[k] * gk[k * co + o];
      }
      if (accumulate) dbd[u0 + tid] += s; else dbd[u0 + tid] = s;
    }
    return;
  }
  const int o = blockIdx.x, k = tid % 27, ul = tid / 27;     // ul < 8 for tid < 216
  int tt[8];
#pragma unroll
  for (int a = 0; a < 8; ++a) tt[a] = t_of(a, k);
  float acc = 0.f;
  for (int c0 = 0; c0 < cl; c0 += kChainCB) {
    __syncthreads();
    for (int i = tid; i < kChainCB * 64; i += 256) dk[i] = c0 + (i >> 6) < cl ? dk4[((long long)(c0 + (i >> 6)) * co + o) * 64 + (i & 63)] : 0.f;
    for (int i = tid; i < kChainCB * 64; i += 256) {
      const int cb = i >> 6, r = i & 63;               // r = u local * 8 + a
      wds[i] = (c0 + cb < cl && u0 + (r >> 3) < cu) ? wd[((long long)(c0 + cb) * cu + u0) * 8 + r] : 0.f;
    }
    __syncthreads();
    if (tid < 216) {
#pragma unroll
      for (int cb = 0; cb < kChainCB; ++cb) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 8; ++a) s += dk[cb * 64 + tt[a]] * wds[cb * 64 + ul * 8 + a];
        acc += s;
      }
    }
  }
  if (tid < 216 && u0 + ul < cu) {
    const float v = acc + (bd ? bd[u0 + ul] * gk[k * co + o] : 0.f);
    float* dst = dwc + ((long long)o * (ce + cu) + ce + u0 + ul) * 27 + k;
    if (accumulate) *dst += v; else *dst = v;
  }
}

// ---- sums of a gradient over the 26 border regions of the volume (region = per axis: all voxels / the first one / the last one;
//      at least one axis restricted).  Stage 1, a compact work list per sample: the two boundary planes in kBorderSplit pieces each
//      (all their voxels), then the D - 2 interior planes (their border rows / columns only: a few hundred voxels) ->
//      q[item][sh][sw][c].  Stage 2: items -> e[sd][sh][sw][c].  Fixed orders, no atomics.
constexpr int kBorderSplit = 64;
__host__ __device__ inline int border_items(int d_ext) { return 2 * kBorderSplit + d_ext - 2; }

template <typename T>
__global__ __launch_bounds__(256) void border_plane_kernel(const T* __restrict__ g, int ld, int d_ext, int h_ext, int w_ext, int c, float* __restrict__ q) {
  constexpr int EPV = Elem<T>::kPer16B;
  __shared__ float red[4][9][8 * 8];                   // [wave][combo][piece * EPV + j]  (c <= 64)
  const int per_n = border_items(d_ext), n = blockIdx.x / per_n, it = blockIdx.x - n * per_n;
  const bool bplane = it < 2 * kBorderSplit;
  const int d = bplane ? (it < kBorderSplit ? 0 : d_ext - 1) : it - 2 * kBorderSplit + 1;
  const int pieces = c / EPV, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int piece = tid % pieces, vlane = tid / pieces, vstep = 256 / pieces;
  float acc[9][EPV];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < EPV; ++j) acc[k][j] = 0.f;
  const T* base = g + ((long long)n * d_ext + d) * h_ext * w_ext * ld + piece * EPV;
  // candidate voxels: a piece of a boundary plane; an interior plane's rows 0 / H - 1 in full and columns 0 / W - 1 of the other rows
  const int nfull = bplane ? h_ext * w_ext : 2 * w_ext + 2 * (h_ext - 2);
  const int per = bplane ? (nfull + kBorderSplit - 1) / kBorderSplit : nfull;
  const int i0 = bplane ? (it % kBorderSplit) * per : 0, i_end = min(nfull, i0 + per);
  for (int i = i0 + vlane; i < i_end; i += vstep) {
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
    q[((long long)blockIdx.x * 9 + k) * c + ch] = (red[0][k][ch] + red[1][k][ch]) + (red[2][k][ch] + red[3][k][ch]);
  }
}

// e[sd][sh][sw][ch]: one workgroup per (sd, sh, sw); its threads = (channel, item lane) sum their share of the region's items
// (first plane's pieces / last plane's pieces / everything), eight loads in flight, then the lanes are added through LDS in lane order
__global__ __launch_bounds__(256) void border_final_kernel(const float* __restrict__ q, int nsamples, int d_ext, int c, float* __restrict__ e) {
  __shared__ float red[256];
  const int r = blockIdx.x, sd = r / 9, k9 = r % 9, tid = threadIdx.x;
  const int ch = tid % c, pl = tid / c, npl = 256 / c;                  // (c divides 64)
  const int per_n = border_items(d_ext);
  const int lo = sd == 2 ? kBorderSplit : 0, cnt = sd == 0 ? per_n : kBorderSplit;   // the region's items within a sample
  float s = 0.f;
  if (r != 0) {
    const int total = nsamples * cnt;
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i0 = pl; i0 < total; i0 += 8 * npl) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * npl;
        if (i < total) part[j] += q[(((long long)(i / cnt) * per_n + lo + i % cnt) * 9 + k9) * c + ch];
      }
    }
    s = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
  }
  red[tid] = s;
  __syncthreads();
  if (tid < c) {
    float t = 0.f;
    for (int j = 0; j < npl; ++j) t += red[j * c + tid];
    e[r * c + tid] = t;
  }
}

}  // namespace

extern "C" {

int mi355_upcat_compose(const float* wd, const float* wc, const float* bd, const float* bc, int32_t cl, int32_t cu, int32_t ce,
                        int32_t co, float* k4, void* wp_d2s, float* biasp, float* delta, void* stream) {
  MI355_REQUIRE(wd && wc && k4 && cl > 0 && cu > 0 && ce >= 0 && co > 0, "upcat_compose: bad argument");
  MI355_REQUIRE(!wp_d2s || (cl % 16 == 0 && co % 32 == 0), "upcat_compose: the d2s packing needs cl %% 16 == 0 and co %% 32 == 0");
  const size_t lds = (size_t)(cu * 27 + 4 * cu * 8 + 27) * sizeof(float);
  MI355_REQUIRE(lds <= 60 * 1024, "upcat_compose: too many intermediate channels");
  MI355_REQUIRE((biasp == nullptr) == (delta == nullptr) && (!biasp || bd), "upcat_compose: the bias tables come together and need b_d");
  upcat_compose_kernel<<<dim3((unsigned)co, (unsigned)((cl + 3) / 4)), dim3(256), lds, (hipStream_t)stream>>>(
      wd, wc, bd, bc, cl, cu, ce, co, k4, (bf16_t*)wp_d2s, biasp, delta);
  return mi355_check_launch("upcat_compose");
}

int mi355_upcat_chain(const float* dk4, const float* wd, const float* wc, const float* bd, const float* esum, int32_t cl, int32_t cu,
                      int32_t ce, int32_t co, float* dwd, float* dwc, float* dbd, int32_t accumulate, void* stream) {
  MI355_REQUIRE(dk4 && wd && wc && esum && dwd && dwc && cl > 0 && cu > 0 && ce >= 0 && co > 0, "upcat_chain: bad argument");
  const size_t lds_c = (size_t)(27 * co + 2 * kChainCB * 64) * sizeof(float);
  MI355_REQUIRE(lds_c <= 60 * 1024, "upcat_chain: too many output channels");
  hipStream_t st = (hipStream_t)stream;
  upcat_chain_wd_kernel<<<dim3((unsigned)cl, (unsigned)cu), dim3(64), 0, st>>>(dk4, wc, cu, ce, co, dwd, accumulate);
  int rc = mi355_check_launch("upcat_chain_wd");
  if (rc) return rc;
  upcat_chain_wc_kernel<<<dim3((unsigned)(co + 1), (unsigned)((cu + 7) / 8)), dim3(256), lds_c, st>>>(
      dk4, wd, wc, bd, esum, cl, cu, ce, co, dwc, dbd, accumulate);
  return mi355_check_launch("upcat_chain_wc");
}

int64_t mi355_border_sums_workspace(int32_t n, int32_t d, int32_t c) { return (int64_t)n * border_items(d) * 9 * c * 4; }

int mi355_border_sums(const void* g, int32_t ld, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c, int32_t dtype, float* workspace,
                      float* e, void* stream) {
  MI355_REQUIRE(g && workspace && e && n > 0 && d >= 2 && h >= 2 && w >= 2, "border_sums: bad argument");
  MI355_REQUIRE(dtype == MI355_DT_F32 || dtype == MI355_DT_BF16, "border_sums: bad dtype");
  const int epv = dtype == MI355_DT_F32 ? 4 : 8;
  MI355_REQUIRE(c > 0 && c <= 64 && c % epv == 0 && ld % epv == 0 && 64 % (c / epv) == 0, "border_sums: channels must be 8, 16, 32 or 64 (f32: 4 .. 64)");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(n * border_items(d)));
  if (dtype == MI355_DT_F32) border_plane_kernel<float><<<grid, dim3(256), 0, st>>>((const float*)g, ld, d, h, w, c, workspace);
  else border_plane_kernel<bf16_t><<<grid, dim3(256), 0, st>>>((const bf16_t*)g, ld, d, h, w, c, workspace);
  int rc = mi355_check_launch("border_sums");
  if (rc) return rc;
  border_final_kernel<<<dim3(27), dim3(256), 0, st>>>(workspace, n, d, c, e);
  return mi355_check_launch("border_sums_final");
}

}  // extern "C"
