// Weight gradient of the implicit-GEMM convolutions (C ABI: mi355_conv_wgrad).
//
//   dw[tap][ci][co] = sum_{n,p} x[n, p*stride + tap - pad, ci] * g[n, p*gs + goff, co]
//
// GEMM view: D[ci][co] += A[ci][k] * B[k][co] with k = output position.  v1 kernel: exact-f32
// MFMA (v_mfma_f32_32x32x2_f32, k = 2 positions per instruction), operands straight from global
// memory (32 lanes read 32 consecutive channels of one position = one 128-B line).  Every wave
// owns TPW taps of one (ci-tile, co-tile) and a disjoint set of rows (n, d, h); partial results
// go to per-wave slabs and a second kernel reduces them in a fixed order (deterministic) and
// scatters into the torch weight layout.
#include "common.h"

namespace {

struct WgradArgs {
  const char* x0; const char* x1;
  int c0, c1, ld0, ld1;
  int n, di, hi, wi;
  const char* g; int cg, ldg;
  int do_, ho, wo;
  int gd, gh, gw, gs, god, goh, gow;
  int stride, pd, ph, pw;
  float* slab;
  int cinp, coutp;       // slab extents (multiples of 32)
  int splits;
  int tap_groups;
  long long rows;        // n*do*ho
};

template <typename T> __device__ __forceinline__ float ld1(const char* p, long long idx) {
  return Elem<T>::load(reinterpret_cast<const T*>(p) + idx);
}

template <typename T, int KS, int TPW>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
  constexpr int NT = KS * KS * KS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int split = blockIdx.x;
  const int ci_base = blockIdx.y * 32;
  const int co_base = (blockIdx.z / a.tap_groups) * 32;
  const int tap0 = (blockIdx.z % a.tap_groups) * TPW;

  const int ci = ci_base + r;
  const bool first = ci_base < a.c0;
  const char* xs = first ? a.x0 : a.x1;
  const long long ldx = first ? a.ld0 : a.ld1;
  const int cix = first ? ci : ci - a.c0;
  const bool ci_ok = ci < a.c0 + a.c1;
  const int co = co_base + r;
  const bool co_ok = co < a.cg;

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const long long nslots = (long long)a.splits * 4;
  for (long long row = (long long)split * 4 + wave; row < a.rows; row += nslots) {
    const int oh = (int)(row % a.ho);
    const long long t2 = row / a.ho;
    const int od = (int)(t2 % a.do_);
    const int n = (int)(t2 / a.do_);
    const long long grow = (((long long)n * a.gd + (od * a.gs + a.god)) * a.gh + (oh * a.gs + a.goh)) * a.gw;
    long long xrow[TPW];
    int kwv[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tap = tap0 + t;
      const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
      const int id = od * a.stride + kd - a.pd, ih = oh * a.stride + kh - a.ph;
      const bool ok = tap < NT && id >= 0 && id < a.di && ih >= 0 && ih < a.hi;
      xrow[t] = ok ? (((long long)n * a.di + id) * a.hi + ih) * a.wi : -1;
      kwv[t] = kw - a.pw;
    }
    for (int ow = h; ow < a.wo + h; ow += 2) {   // both halves iterate the same trip count
      const bool pos_ok = ow < a.wo;
      float b = 0.f;
      if (pos_ok && co_ok) b = ld1<T>(a.g, (grow + (ow * a.gs + a.gow)) * a.ldg + co);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int iw = ow * a.stride + kwv[t];
        float av = 0.f;
        if (pos_ok && ci_ok && xrow[t] >= 0 && iw >= 0 && iw < a.wi) av = ld1<T>(xs, (xrow[t] + iw) * ldx + cix);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[t], 0, 0, 0);
      }
    }
  }
  // slab[(split*4+wave)][tap][ci][co]
  float* sl = a.slab + ((long long)(split * 4 + wave) * NT) * a.cinp * a.coutp;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = tap0 + t;
    if (tap < NT) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = ci_base + acc_row(i, h);
        sl[((long long)tap * a.cinp + row) * a.coutp + co] = acc[t][i];
      }
    }
  }
}

struct WreduceArgs {
  const float* slab; int nslabs; int ntaps, ks, cinp, coutp;
  float* dw; int cout, cin;
  long long s_co, s_ci, s_k0, s_k1, s_k2;
  int tb0, tb1, tb2, ts0, ts1, ts2;
  int accumulate;
};

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WreduceArgs a) {
  const long long per = (long long)a.ntaps * a.cinp * a.coutp;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= per) return;
  const int co = (int)(idx % a.coutp);
  const int ci = (int)((idx / a.coutp) % a.cinp);
  const int tap = (int)(idx / ((long long)a.coutp * a.cinp));
  if (co >= a.cout || ci >= a.cin) return;
  float s = 0.f;
  for (int k = 0; k < a.nslabs; ++k) s += a.slab[(long long)k * per + idx];
  const int td = tap / (a.ks * a.ks), th = (tap / a.ks) % a.ks, tw = tap % a.ks;
  const long long dst = co * a.s_co + ci * a.s_ci + (a.tb0 + a.ts0 * td) * a.s_k0 +
                        (a.tb1 + a.ts1 * th) * a.s_k1 + (a.tb2 + a.ts2 * tw) * a.s_k2;
  if (a.accumulate) a.dw[dst] += s; else a.dw[dst] = s;
}

struct WPlan { int ks, tpw, tap_groups, ci_tiles, co_tiles, splits, cinp32, coutp32; long long rows; };

int wplan(const mi355_wgrad_desc* d, WPlan* p) {
  MI355_REQUIRE(d && d->x0 && d->g && d->dw, "wgrad: null pointer");
  MI355_REQUIRE(d->dtype == MI355_DT_F32 || d->dtype == MI355_DT_BF16, "wgrad: bad dtype");
  MI355_REQUIRE(d->ks >= 1 && d->ks <= 4, "wgrad: unsupported ks=%d", d->ks);
  MI355_REQUIRE(d->c0 > 0 && d->c0 % 16 == 0 && d->c1 % 16 == 0 && d->cg % 16 == 0, "wgrad: channels must be multiples of 16");
  MI355_REQUIRE(d->c1 == 0 || (d->x1 && d->c0 % 32 == 0), "wgrad: concat split must be a multiple of 32");
  MI355_REQUIRE(d->cin <= d->c0 + d->c1 && d->cout <= d->cg, "wgrad: real extents exceed padded ones");
  p->ks = d->ks;
  p->tpw = d->ks == 1 ? 1 : (d->ks == 3 ? 9 : 8);
  const int nt = d->ks * d->ks * d->ks;
  p->tap_groups = (nt + p->tpw - 1) / p->tpw;
  p->cinp32 = ((d->c0 + d->c1 + 31) / 32) * 32;
  p->coutp32 = ((d->cg + 31) / 32) * 32;
  p->ci_tiles = p->cinp32 / 32;
  p->co_tiles = p->coutp32 / 32;
  p->rows = (long long)d->n * d->do_ * d->ho;
  long long wgs = (long long)p->ci_tiles * p->co_tiles * p->tap_groups;
  long long s = 2048 / wgs;
  if (s < 1) s = 1;
  const long long max_s = (p->rows + 3) / 4;
  if (s > max_s) s = max_s;
  // cap the slab workspace at ~256 MiB
  const long long slab_bytes = (long long)nt * p->cinp32 * p->coutp32 * 4;
  while (s > 1 && s * 4 * slab_bytes > (256ll << 20)) s /= 2;
  p->splits = (int)s;
  return MI355_OK;
}

template <typename T>
void launch_wgrad(const WgradArgs& a, const WPlan& p, hipStream_t st) {
  dim3 grid(p.splits, p.ci_tiles, p.co_tiles * p.tap_groups), block(256);
  switch (p.ks) {
    case 1: hipLaunchKernelGGL((wgrad_kernel<T, 1, 1>), grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL((wgrad_kernel<T, 2, 8>), grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL((wgrad_kernel<T, 3, 9>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((wgrad_kernel<T, 4, 8>), grid, block, 0, st, a); break;
  }
}

}  // namespace

extern "C" int64_t mi355_conv_wgrad_workspace(const mi355_wgrad_desc* d) {
  WPlan p;
  if (wplan(d, &p)) return -1;
  return (int64_t)p.splits * 4 * p.ks * p.ks * p.ks * p.cinp32 * p.coutp32 * 4;
}

extern "C" int mi355_conv_wgrad(const mi355_wgrad_desc* d, void* stream) {
  WPlan p;
  int rc = wplan(d, &p);
  if (rc) return rc;
  const int64_t need = mi355_conv_wgrad_workspace(d);
  MI355_REQUIRE(d->workspace && d->workspace_bytes >= need, "wgrad: workspace too small (%lld < %lld)",
                (long long)d->workspace_bytes, (long long)need);
  MI355_REQUIRE((d->do_ - 1) * d->gs + d->goff[0] < d->gd && (d->ho - 1) * d->gs + d->goff[1] < d->gh &&
                    (d->wo - 1) * d->gs + d->goff[2] < d->gw, "wgrad: grid exceeds g");
  hipStream_t st = (hipStream_t)stream;
  WgradArgs a;
  a.x0 = (const char*)d->x0; a.x1 = (const char*)d->x1;
  a.c0 = d->c0; a.c1 = d->c1; a.ld0 = d->ld0; a.ld1 = d->ld1;
  a.n = d->n; a.di = d->di; a.hi = d->hi; a.wi = d->wi;
  a.g = (const char*)d->g; a.cg = d->cg; a.ldg = d->ldg;
  a.do_ = d->do_; a.ho = d->ho; a.wo = d->wo;
  a.gd = d->gd; a.gh = d->gh; a.gw = d->gw; a.gs = d->gs;
  a.god = d->goff[0]; a.goh = d->goff[1]; a.gow = d->goff[2];
  a.stride = d->stride; a.pd = d->pad[0]; a.ph = d->pad[1]; a.pw = d->pad[2];
  a.slab = d->workspace; a.cinp = p.cinp32; a.coutp = p.coutp32;
  a.splits = p.splits; a.tap_groups = p.tap_groups; a.rows = p.rows;
  if (d->dtype == MI355_DT_F32) launch_wgrad<float>(a, p, st); else launch_wgrad<bf16_t>(a, p, st);
  rc = mi355_check_launch("conv_wgrad");
  if (rc) return rc;
  WreduceArgs q;
  q.slab = d->workspace; q.nslabs = p.splits * 4; q.ks = d->ks; q.ntaps = d->ks * d->ks * d->ks;
  q.cinp = p.cinp32; q.coutp = p.coutp32;
  q.dw = d->dw; q.cout = d->cout; q.cin = d->cin;
  q.s_co = d->s_co; q.s_ci = d->s_ci; q.s_k0 = d->s_k[0]; q.s_k1 = d->s_k[1]; q.s_k2 = d->s_k[2];
  q.tb0 = d->tbase[0]; q.tb1 = d->tbase[1]; q.tb2 = d->tbase[2];
  q.ts0 = d->tstep[0]; q.ts1 = d->tstep[1]; q.ts2 = d->tstep[2];
  q.accumulate = d->accumulate;
  const long long per = (long long)q.ntaps * q.cinp * q.coutp;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, q);
  return mi355_check_launch("wgrad_reduce");
}
