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
#include <type_traits>
#include <cstring>
#include <vector>

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
  int g_cls_cout;        // fast kernel, KS=1: GEMM column blk*g_cls_cout + co reads g at 2p + bits(blk)
  int xn;                // samples held by x: grid sample n reads x sample n % xn
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
      xrow[t] = ok ? (((long long)(n % a.xn) * a.di + id) * a.hi + ih) * a.wi : -1;
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
  int s2d_cp;
  int co_cls;            // >0: co index is blk*co_cls + co, blk selects the destination tap
};

// 256 threads = 32 float4 groups (128 consecutive elements, 512 B per slab row) x 8 slab lanes;
// 4 independent 16-B loads in flight per thread, fixed-order LDS combine (deterministic).
// (A variant with 16 slab lanes x 8 loads in flight and 64-element blocks looked equal in the micro-benchmark
// but cost 0.3 ms per training step in the interleaved A/B -- twice the blocks, half the row length per block.)
// slabs are written once and read once: non-temporal loads keep them from displacing the activations the next
// kernels re-read (-0.04 ms per step, 3 of 3 interleaved rounds)
// slab stores: written once, read once by a reduction that (round 4) runs at the end of the backward pass
__device__ __forceinline__ void st_slab(float* p, float v) {
#ifdef WGRAD_NT_SLABS
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ float4 ld_once(const float* p) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
// F4 float4 groups x SL slab lanes per block (F4 * SL = 256): 32 x 8 for ordinary layers; 8 x 32 when a layer has
// so few weights (1x1x1 convs: 1024 elements under 2048 slabs) that 128-element blocks would leave 8 blocks.
// (the three reduce forms are __device__ bodies over a caller-provided LDS area and block coordinates: the single-layer kernels
//  below and wgrad_reduce_multi_kernel -- the slabs of up to 16 layers in one launch -- run the same code)
constexpr int kRedLdsFloats = 32 * (4 * 64 + 1);                     // the largest form (space-to-depth patch): 32.1 KB
template <int F4, int SL>
__device__ __forceinline__ void wgrad_reduce_body(const WreduceArgs& a, const int bx, float* smem) {
  static_assert(F4 * SL == 256, "block shape");
  float4 (*red)[F4] = reinterpret_cast<float4 (*)[F4]>(smem);         // [SL][F4]
  const long long per = (long long)a.ntaps * a.cinp * a.coutp;        // multiple of 1024
  const int e = threadIdx.x % F4, sl = threadIdx.x / F4;
  const long long idx = ((long long)bx * F4 + e) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < per) {
    const float* base = a.slab + idx;
    int k = sl;
    for (; k + 3 * SL < a.nslabs; k += 4 * SL) {
      const float4 v0 = ld_once(base + (long long)k * per);
      const float4 v1 = ld_once(base + (long long)(k + SL) * per);
      const float4 v2 = ld_once(base + (long long)(k + 2 * SL) * per);
      const float4 v3 = ld_once(base + (long long)(k + 3 * SL) * per);
      s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
      s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; k < a.nslabs; k += SL) {
      const float4 v = *reinterpret_cast<const float4*>(base + (long long)k * per);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[sl][e] = s;
  __syncthreads();
  if (sl != 0 || idx >= per) return;
#pragma unroll
  for (int q = 1; q < SL; ++q) {
    const float4 v = red[q][e];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const float f[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long id = idx + j;
    int co = (int)(id % a.coutp);
    int ci = (int)((id / a.coutp) % a.cinp);
    const int tap = (int)(id / ((long long)a.coutp * a.cinp));
    int blk = 0;
    if (a.s2d_cp) { blk = ci / a.s2d_cp; ci = ci % a.s2d_cp; }
    if (a.co_cls) { blk = co / a.co_cls; co = co % a.co_cls; }
    if (co >= a.cout || ci >= a.cin || blk >= 8) continue;
    const int td = tap / (a.ks * a.ks), th = (tap / a.ks) % a.ks, tw = tap % a.ks;
    const long long dst = co * a.s_co + ci * a.s_ci + (a.tb0 + a.ts0 * td + (blk >> 2)) * a.s_k0 +
                          (a.tb1 + a.ts1 * th + ((blk >> 1) & 1)) * a.s_k1 + (a.tb2 + a.ts2 * tw + (blk & 1)) * a.s_k2;
    if (a.accumulate) a.dw[dst] += f[j]; else a.dw[dst] = f[j];
  }
}
template <int F4, int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WreduceArgs a) {
  __shared__ float4 red[SL * F4];
  wgrad_reduce_body<F4, SL>(a, blockIdx.x, reinterpret_cast<float*>(red));
}


// Few slabs, many weights (the 8^3 / 16^3 levels: 2-16 slabs of up to 28 MB): the cost is the WRITE side --
// consecutive slab elements are consecutive output channels, 27*cin floats apart in dw[co][ci][taps], so the
// kernel above scatters 4-byte stores (measured 176 us for 512->512).  Here a block owns all taps of an
// 8 (ci) x 32 (co) patch, sums the slabs in order (128-B read segments) and turns the patch through LDS so
// that every output-channel row is written as one contiguous run of 8 * taps floats.
// CIB = input channels per block (8, 4 or 2: fewer when the patch count would not fill the chip); the 8 / CIB
// thread groups of a channel split the taps.
// Round 4: 16-byte loads along co (a thread owns 4 consecutive output channels of (ci, tap)s), every slab's loads of a thread issued
// before the first add (nslabs <= 16: up to 64 loads in flight; the 4-byte form with one slab at a time read its 14 - 56 MB of
// slabs at 1.4 TB/s: 16 - 27 us per layer).  Summation in slab order: bit-identical to the form it replaces.
template <int NT, int CIB>
__device__ __forceinline__ void wgrad_reduce_dense_body(const WreduceArgs& a, const int bx, const int by, float* tile) {
  constexpr int ROW = CIB * NT + 1;                       // LDS row of one output channel: [ci][tap] (+1: odd stride)
  constexpr int ITEMS = CIB * NT;                         // (ci, tap) pairs of the patch; 8 threads (co quads) per pair
  constexpr int PER = (ITEMS + 31) / 32;                  // pairs per thread
  static_assert(32 * ROW <= kRedLdsFloats, "LDS area");
  const int t = threadIdx.x, cq = t & 7, grp = t >> 3;    // co quad, pair lane (32 of them)
  const int ci0 = bx * CIB, co0 = by * 32;
  const long long per = (long long)NT * a.cinp * a.coutp;
  const long long tstride = (long long)a.cinp * a.coutp;
  float4 s[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) s[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int item = grp + i * 32;
    if (item < ITEMS) {
      const int ci_l = item / NT, tap = item - ci_l * NT;
      const float* base = a.slab + (long long)tap * tstride + (long long)(ci0 + ci_l) * a.coutp + co0 + cq * 4;
      for (int k0 = 0; k0 < a.nslabs; k0 += 4) {          // four slabs' loads in flight per pair, added in slab order
        float4 v[4];
#pragma unroll
#ifdef WGRAD_NT_DENSE
        for (int j = 0; j < 4; ++j) v[j] = k0 + j < a.nslabs ? ld_once(base + (long long)(k0 + j) * per) : make_float4(0.f, 0.f, 0.f, 0.f);
#else
        for (int j = 0; j < 4; ++j) v[j] = k0 + j < a.nslabs ? *reinterpret_cast<const float4*>(base + (long long)(k0 + j) * per) : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[i].x += v[j].x; s[i].y += v[j].y; s[i].z += v[j].z; s[i].w += v[j].w; }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int item = grp + i * 32;
    if (item < ITEMS) {
      const int ci_l = item / NT, tap = item - ci_l * NT;
      float* d = tile + (cq * 4) * ROW + ci_l * NT + tap;
      d[0] = s[i].x; d[ROW] = s[i].y; d[2 * ROW] = s[i].z; d[3 * ROW] = s[i].w;
    }
  }
  __syncthreads();
  const int nci = min(CIB, a.cin - ci0);          // valid input channels of this patch
  if (nci <= 0) return;
  const int run = nci * NT;
  for (int r = 0; r < 32; ++r) {
    const int co = co0 + r;
    if (co >= a.cout) break;
    float* dst = a.dw + (long long)co * a.s_co + (long long)ci0 * a.s_ci;
    for (int j = t; j < run; j += 256) {
      const float v = tile[r * ROW + j];
      if (a.accumulate) dst[j] += v; else dst[j] = v;
    }
  }
}
template <int NT, int CIB>
__global__ __launch_bounds__(256) void wgrad_reduce_dense_kernel(const WreduceArgs a) {
  __shared__ float tile[32 * (CIB * NT + 1)];
  wgrad_reduce_dense_body<NT, CIB>(a, blockIdx.x, blockIdx.y, tile);
}

// The PatchGAN k4 s2 layers computed on space-to-depth tensors: slab rows are (j, blk * cp + c) with 8 dense
// taps j and 8 parity blocks blk, and dw[co][c][kd][kh][kw] has kd = 2 jd + bd (likewise h, w): for one (co, c)
// the 8 x 8 (j, blk) values are the 64 contiguous taps of the 4x4x4 kernel.  The generic kernel scattered them as
// 4-byte stores (101 us for the 256->512 layer, whose "reduce" has a single slab); here a block turns a
// 4 (c) x 32 (co) x 64 patch through LDS and writes 256 contiguous floats per output channel.
__device__ __forceinline__ void wgrad_reduce_s2d_body(const WreduceArgs& a, const int bx, const int by, float* tile) {
  constexpr int CB = 4, ROW = CB * 64 + 1;
  static_assert(32 * ROW <= kRedLdsFloats, "LDS area");
  const int t = threadIdx.x, co_l = t & 31, j = t >> 5;                    // j = dense tap of the k2 formulation
  const int c0 = bx * CB, co0 = by * 32;
  const long long per = (long long)8 * a.cinp * a.coutp;
  float s[8][CB];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int c = 0; c < CB; ++c) s[b][c] = 0.f;
  const float* base = a.slab + ((long long)j * a.cinp + c0) * a.coutp + co0 + co_l;
  for (int k = 0; k < a.nslabs; ++k) {
    const float* p = base + (long long)k * per;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int c = 0; c < CB; ++c) s[b][c] += p[((long long)b * a.s2d_cp + c) * a.coutp];
  }
  const int jd = j >> 2, jh = (j >> 1) & 1, jw = j & 1;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int tap = (2 * jd + (b >> 2)) * 16 + (2 * jh + ((b >> 1) & 1)) * 4 + (2 * jw + (b & 1));
#pragma unroll
    for (int c = 0; c < CB; ++c) tile[co_l * ROW + c * 64 + tap] = s[b][c];
  }
  __syncthreads();
  const int nc = min(CB, a.cin - c0);
  if (nc <= 0) return;
  const int run = nc * 64;
  for (int r = 0; r < 32; ++r) {
    const int co = co0 + r;
    if (co >= a.cout) break;
    float* dst = a.dw + (long long)co * a.s_co + (long long)c0 * a.s_ci;
    for (int i = t; i < run; i += 256) {
      const float v = tile[r * ROW + i];
      if (a.accumulate) dst[i] += v; else dst[i] = v;
    }
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_s2d_kernel(const WreduceArgs a) {
  __shared__ float tile[kRedLdsFloats];
  wgrad_reduce_s2d_body(a, blockIdx.x, blockIdx.y, tile);
}

// Slab reductions of up to kRedChunk layers in ONE launch (mi355_wgrad_reduce_multi): a step has ~30 weight-gradient
// launches, each followed by a 7 - 23 us reduction of its slabs that nothing but the optimiser (or the bucket's all-reduce)
// waits for; deferred to the end of a backward segment they are one launch.  Block -> (job, block of the job) through
// the prefix table; each job runs the form (and so the summation order) its own launch would: bit-identical results.
constexpr int kRedChunk = 16;
enum { kRedGeneric = 0, kRedGenericNarrow, kRedDense27_8, kRedDense27_4, kRedDense27_2, kRedDense1_8, kRedDense1_4, kRedDense1_2, kRedS2d };
struct WreduceMulti { WreduceArgs a[kRedChunk]; int kind[kRedChunk], gx[kRedChunk], first[kRedChunk + 1]; int n; };
// The jobs of the generic form (the marching kernels' 64 - 256 slabs per layer: most of the bytes) in a kernel of their own: 56 registers
// and 4 KB of LDS instead of the union's 166 / 32 KB -- 8 waves per SIMD instead of 3 to keep the slab reads in flight.
#ifndef LIGHT_F4
#define LIGHT_F4 32
#endif
__global__ __launch_bounds__(256) void wgrad_reduce_multi_light_kernel(const WreduceMulti m) {
  __shared__ float4 red[256];
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < m.n && b >= m.first[j + 1]) ++j;
  wgrad_reduce_body<LIGHT_F4, 256 / LIGHT_F4>(m.a[j], b - m.first[j], reinterpret_cast<float*>(red));
}
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WreduceMulti m) {
  __shared__ float smem[kRedLdsFloats];
  const int b = blockIdx.x;
  int j = 0;
  while (j + 1 < m.n && b >= m.first[j + 1]) ++j;
  const int local = b - m.first[j], gx = m.gx[j];
  const int bx = local % gx, by = local / gx;
  const WreduceArgs& a = m.a[j];
  switch (m.kind[j]) {
    case kRedGeneric: wgrad_reduce_body<32, 8>(a, local, smem); break;
    case kRedGenericNarrow: wgrad_reduce_body<8, 32>(a, local, smem); break;
    case kRedDense27_8: wgrad_reduce_dense_body<27, 8>(a, bx, by, smem); break;
    case kRedDense27_4: wgrad_reduce_dense_body<27, 4>(a, bx, by, smem); break;
    case kRedDense27_2: wgrad_reduce_dense_body<27, 2>(a, bx, by, smem); break;
    case kRedDense1_8: wgrad_reduce_dense_body<1, 8>(a, bx, by, smem); break;
    case kRedDense1_4: wgrad_reduce_dense_body<1, 4>(a, bx, by, smem); break;
    case kRedDense1_2: wgrad_reduce_dense_body<1, 2>(a, bx, by, smem); break;
    default: wgrad_reduce_s2d_body(a, bx, by, smem); break;
  }
}

// ------------------------------------------------------------------------------------------
// bf16 3x3x3 stride-1 weight gradient on v_mfma_f32_32x32x16_bf16.
// D[ci][co] += sum_k A[ci][k] * B[k][co], k = 16 consecutive W positions.  Both operands are
// k-strided in memory ([position][channel] rows of 64 B), so the fragments are fetched with the
// gfx950 transposed LDS read ds_read_b64_tr_b16 (4 positions x 16 channels per 16-lane group, two
// reads per fragment); 4 consecutive positions x 32 channels = 256 contiguous bytes per 32-lane
// half => conflict-free.  A workgroup stages the x halo and the g tile of one 32x32 (ci, co)
// block once and its 4 waves share them, each owning 7 (6) of the 27 taps; the g fragment is
// reused across a wave's taps.  Accumulators live across all tiles of the workgroup's split.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 frag_tr(const char* p) {
  const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 4 * 64);
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int KS, int TD, int TH, int TW>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgradArgs a, int tiles_d, int tiles_h,
                                                             int tiles_w, int ntiles) {
  constexpr int NT = KS * KS * KS;
  constexpr int TPWV = (NT + 3) / 4;          // taps per wave
  constexpr int HD = TD + KS - 1, HH = TH + KS - 1, HW = TW + KS - 1;
  constexpr int XROWS = HD * HH * HW, GROWS = TD * TH * TW;
  constexpr int NPX = (XROWS * 4 + 255) / 256, NPG = (GROWS * 4 + 255) / 256;
  constexpr int SEGS = TW / 16;
  static_assert(TW % 16 == 0, "tile width");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  char* gsm = smem + XROWS * 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ci_base = blockIdx.y * 32, co_base = blockIdx.z * 32;
  const bool first = ci_base < a.c0;
  const char* xsrc = first ? a.x0 : a.x1;
  const long long ldx = first ? a.ld0 : a.ld1;
  const int cix = first ? ci_base : ci_base - a.c0;
  const int cx_lim = (first ? a.c0 : a.c1) - cix;      // channels of this source left from cix
  // transposed-conv classes folded into the column index: this workgroup's 32 columns are one class
  int gs = 1, gbd = 0, gbh = 0, gbw = 0, gco = co_base;
  if (a.g_cls_cout) {
    const int cls = co_base / a.g_cls_cout;
    gco = co_base - cls * a.g_cls_cout;
    gs = 2; gbd = cls >> 2; gbh = (cls >> 1) & 1; gbw = cls & 1;
  }
  const int cg_lim = (a.g_cls_cout ? a.g_cls_cout : a.cg) - gco;
  constexpr bool WSPLIT = (KS == 1);                    // one tap: the 4 waves split the k-groups instead

  int toff[TPWV];
#pragma unroll
  for (int i = 0; i < TPWV; ++i) {
    const int tap = WSPLIT ? 0 : wave + 4 * i;
    const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
    toff[i] = tap < NT ? ((kd * HH + kh) * HW + kw) * 64 : 0;
  }
  const int gi = lane & 15;
  const int lane_off = (8 * h + (gi >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (gi & 3)) * 2;

  f32x16 acc[TPWV];
#pragma unroll
  for (int i = 0; i < TPWV; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int tw_i = t % tiles_w; t /= tiles_w;
    const int th_i = t % tiles_h; t /= tiles_h;
    const int td_i = t % tiles_d;
    const int n = t / tiles_d;
    const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;
    uint4 sx[NPX];
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int p = tid + i * 256;
      const int row = p >> 2, part = p & 3;
      const int hw = row % HW, hh = (row / HW) % HH, hd = row / (HW * HH);
      const int gd = d0 + hd - a.pd, gh = h0 + hh - a.ph, gw = w0 + hw - a.pw;
      const bool ok = p < XROWS * 4 && part * 8 < cx_lim && gd >= 0 && gd < a.di && gh >= 0 && gh < a.hi && gw >= 0 && gw < a.wi;
      sx[i] = ok ? *reinterpret_cast<const uint4*>(xsrc + (((((long long)(n % a.xn) * a.di + gd) * a.hi + gh) * a.wi + gw) * ldx + cix + part * 8) * 2)
                 : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();                      // every wave finished reading the previous tile
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int p = tid + i * 256;
      if (p < XROWS * 4) *reinterpret_cast<uint4*>(xs + p * 16) = sx[i];
    }
#pragma unroll
    for (int i = 0; i < NPG; ++i) {
      const int p = tid + i * 256;
      const int row = p >> 2, part = p & 3;
      const int sw = row % TW, sh = (row / TW) % TH, sd = row / (TW * TH);
      const int gd = d0 + sd, gh = h0 + sh, gw = w0 + sw;
      const bool ok = p < GROWS * 4 && part * 8 < cg_lim && gd < a.do_ && gh < a.ho && gw < a.wo;
      const uint4 v = ok ? *reinterpret_cast<const uint4*>(a.g + (((((long long)n * a.gd + (gd * gs + gbd)) * a.gh + (gh * gs + gbh)) * a.gw + (gw * gs + gbw)) * (long long)a.ldg + gco + part * 8) * 2)
                         : make_uint4(0, 0, 0, 0);
      if (p < GROWS * 4) *reinterpret_cast<uint4*>(gsm + p * 16) = v;
    }
    __syncthreads();
#pragma unroll 2
    for (int kg = WSPLIT ? wave : 0; kg < TD * TH * SEGS; kg += WSPLIT ? 4 : 1) {
      const int seg = kg % SEGS, sh = (kg / SEGS) % TH, sd = kg / (SEGS * TH);
      const char* gp = gsm + ((sd * TH + sh) * TW + seg * 16) * 64 + lane_off;
      const char* xp = xs + ((sd * HH + sh) * HW + seg * 16) * 64 + lane_off;
      const bf16x8 b = frag_tr(gp);
#pragma unroll
      for (int i = 0; i < TPWV; ++i) {
        const bf16x8 af = frag_tr(xp + toff[i]);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b, acc[i], 0, 0, 0);
      }
    }
  }
  float* sl = a.slab + ((long long)(WSPLIT ? blockIdx.x * 4 + wave : blockIdx.x) * NT) * a.cinp * a.coutp;
  const int co = co_base + r;
#pragma unroll
  for (int i = 0; i < TPWV; ++i) {
    const int tap = WSPLIT ? 0 : wave + 4 * i;
    if (tap < NT) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int row = ci_base + acc_row(j, h);
        st_slab(&sl[((long long)tap * a.cinp + row) * a.coutp + co], acc[i][j]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// wgrad_march_kernel: the 3x3x3 stride-1 weight gradient of the wide levels (W >= 32), marching along d.
//
// wgrad_bf16_kernel<3, 2, 4, 32> stages a 4 x 6 x 34 halo for 2 x 4 x 32 output positions (3.2x the tile's own
// voxels) through registers, with a barrier on either side of the copy: 124 us for 32 -> 32 at 128^3, where the
// MFMA work is 35 us and reading both operands once is 34 us.  Here a workgroup owns an 8 (h) x 32 (w) footprint
// of one (ci-tile, co-tile) pair and walks a segment of output planes:
//   * x planes (10 x 34 halo voxels x 32 channels) and g planes (8 x 32 x 32 channels) arrive by LDS-DMA
//     (buffer_load_dwordx4 ... lds, 1 KB per instruction) TWO steps ahead of their use (one step of MFMAs, 1.6 us, is
//     shorter than the loaded memory latency: with one step of look-ahead the kernel ran at 3.4 us per plane): a ring
//     of 5 x planes -- output plane p pairs with x planes p - 1, p, p + 1 (kd = 0, 1, 2) while p + 2 and p + 3 are in
//     flight -- and 3 g planes, 158 of the 160 KB; every x plane is read from memory once per workgroup (halo 1.33x in
//     (h, w), (L + 2) / L along d), one barrier per plane, which waits for the OLDER of the two copies in flight only;
//   * the 4 waves are (w-segment of 16 positions) x (half of the 9 (kd, kw) tap columns: 4 whole columns each and the
//     footprint's upper / lower rows of the ninth, 108 MFMAs per plane for every wave).  A wave keeps the 8 g
//     fragments of its segment (one per footprint row) in registers for the whole plane, and an x fragment of halo row y
//     feeds the three taps kh = 0, 1, 2 (output rows y, y - 1, y - 2): ~1 ds_read_b64_tr_b16 per MFMA instead of 2.3;
//   * the accumulators (15 taps x 16 registers) stay resident for the whole segment; at the end the waves that hold
//     parts of a tap are added through LDS in a fixed order and ONE slab per workgroup is written (28 MB per layer
//     instead of 57 MB); wgrad_reduce_kernel sums the slabs in order as before: bit-identical reruns.
constexpr int kWmFH = 8, kWmFW = 32, kWmHR = kWmFH + 2, kWmHC = kWmFW + 2;
constexpr int kWmXV = kWmHR * kWmHC;                  // 340 halo voxels per x plane
constexpr int kWmXI = 24;                             // 1-KB DMA instructions per x plane, 6 per wave: 22 carry data, the last
                                                      // two (out-of-range source: zeros) go to a 2-KB dump area
constexpr int kWmXS = 22 * 1024;                      // bytes per x ring slot
constexpr int kWmGI = 16, kWmGS = kWmGI * 1024;       // g plane: 256 voxels x 64 B
constexpr int kWmXR = 5, kWmGR = 3;                   // ring depths
constexpr int kWmLds = kWmXR * kWmXS + kWmGR * kWmGS + 2048;   // 160 KB
constexpr int kWmPerStep = kWmXI / 4 + kWmGI / 4;     // DMA instructions per wave and step

struct WMarchArgs { int seg_len, nseg, tiles_h, tiles_w, nslabs, ci_tiles, co_tiles; };

// Tap columns of a wave: HALF 0 owns (kd, kw) columns 0..3, HALF 1 columns 5..8 (local index 0..3), and both own
// column 4 (local index 4) for half of the footprint rows each -- 4.5 columns = 108 MFMAs per plane for every wave.
constexpr int wm_unit(int half, int i) { return i == 4 ? 4 : (half ? 5 + i : i); }
constexpr bool wm_valid(int half, int i, int y, int kh) {            // does halo row y feed tap kh of local column i?
  const int sh = y - kh, lo = (i == 4 && half == 1) ? kWmFH / 2 : 0, hi = (i == 4 && half == 0) ? kWmFH / 2 : kWmFH;
  return sh >= lo && sh < hi;
}
constexpr bool wm_need(int half, int i, int y) { return y < kWmHR && (wm_valid(half, i, y, 0) || wm_valid(half, i, y, 1) || wm_valid(half, i, y, 2)); }
constexpr int wm_row_mfmas(int half, int y) {
  int n = 0;
  for (int i = 0; i < 5; ++i) for (int kh = 0; kh < 3; ++kh) n += wm_valid(half, i, y, kh) ? 1 : 0;
  return n;
}
constexpr int wm_row_reads(int half, int y) {
  int n = 0;
  for (int i = 0; i < 5; ++i) n += wm_need(half, i, y) ? 2 : 0;
  return n;
}

// one plane of one wave.  between(y): called behind halo row y's MFMAs, fenced (nothing is scheduled across it): the
// step's copy tickets, one per row -- issued together at the top of the step, the 10 copies cost the wave their whole issue
// time with the matrix pipe idle; behind a row, a copy's issue runs under the MFMA in flight (conv_march.h).
template <int HALF, typename Between>
__device__ __forceinline__ void wm_plane(f32x16 (&acc)[5][3], const char* xs, const char* gp, const int p, const int seg_lane_off,
                                         Between between) {
  bf16x8 gf[kWmFH];
#pragma unroll
  for (int sh = 0; sh < kWmFH; ++sh) gf[sh] = frag_tr(gp + sh * (kWmFW * 64));
  const char* xb[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int u = wm_unit(HALF, i), kd = u / 3, kw = u - kd * 3;
    xb[i] = xs + ((unsigned)(p + kd) % kWmXR) * kWmXS + kw * 64 + seg_lane_off;      // x plane q sits in ring slot (q + 1) % 5
  }
  bf16x8 xf[2][5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
    if (wm_need(HALF, i, 0)) xf[0][i] = frag_tr(xb[i]);
  // order: the g fragments and halo row 0, then per halo row its MFMAs with the next row's reads spread between them
  // (one wave per SIMD: nothing else hides the LDS latency, and hipcc would sink every read next to its first use)
  __builtin_amdgcn_sched_group_barrier(0x100, 2 * kWmFH + wm_row_reads(HALF, 0), 0);
#pragma unroll
  for (int y = 0; y < kWmHR; ++y) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (wm_need(HALF, i, y + 1)) xf[(y + 1) & 1][i] = frag_tr(xb[i] + (y + 1) * (kWmHC * 64));
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
        if (wm_valid(HALF, i, y, kh))
          acc[i][kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[y & 1][i], gf[y - kh], acc[i][kh], 0, 0, 0);
    const int M = wm_row_mfmas(HALF, y), R = wm_row_reads(HALF, y + 1);
#pragma unroll
    for (int k = 0; k < M; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      const int nr = R * (k + 1) / M - R * k / M;
      if (nr == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      else if (nr == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      else if (nr == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    between(y);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(256, 1) void wgrad_march_kernel(const WgradArgs a, const WMarchArgs m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  char* const xs = smem;
  char* const gsm = smem + kWmXR * kWmXS;
  char* const dump = gsm + kWmGR * kWmGS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int seg = wave & 1, half = wave >> 1;
  // workgroup -> (slab, ci tile, co tile): the tile pairs of one slab (same g planes / same x voxels) are neighbours in
  // dispatch order ON THE SAME XCD (workgroup ids go round-robin over the 8 XCDs), so that what they share is read from
  // memory once and from that XCD's L2 afterwards (96 -> 32: three ci tiles re-read the same g)
  const int tpairs = m.ci_tiles * m.co_tiles;
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int slab = (kq / tpairs) * 8 + xcd, pair = kq % tpairs;
  if (slab >= m.nslabs) return;                                       // (whole workgroup; grid rounded up to 8 slabs)
  const int ci_base = (pair % m.ci_tiles) * 32, co_base = (pair / m.ci_tiles) * 32;
  const bool first = ci_base < a.c0;
  const char* xsrc = first ? a.x0 : a.x1;
  const int ldx = first ? a.ld0 : a.ld1;
  const int cix = first ? ci_base : ci_base - a.c0;
  const int csrc = first ? a.c0 : a.c1;
  const int cx_lim = csrc - cix, cg_lim = a.cg - co_base;

  const int per_seg = m.tiles_h * m.tiles_w, per_sample = per_seg * m.nseg;
  int t = slab;
  const int tn = t / per_sample;
  t -= tn * per_sample;
  const int sg = t / per_seg;
  t -= sg * per_seg;
  const int th_i = t / m.tiles_w, tw_i = t - th_i * m.tiles_w;
  const int d0 = sg * m.seg_len, d1 = min(a.do_, d0 + m.seg_len);
  const int h0 = th_i * kWmFH, w0 = tw_i * kWmFW;

  // ---- DMA source offsets: instruction j = i * 4 + wave covers 16 voxels, lane = (voxel, 16-byte piece of 8 channels);
  //      voxels outside the volume, channels past the source's width: out-of-range offset -> the DMA writes zeros
  const int piece = lane & 3;
  int xoff[kWmXI / 4], goff[kWmGI / 4];
#pragma unroll
  for (int i = 0; i < kWmXI / 4; ++i) {
    const int v = (i * 4 + wave) * 16 + (lane >> 2);
    const int y = v / kWmHC, col = v - y * kWmHC;
    const int gh = h0 - 1 + y, gw = w0 - 1 + col;
    const bool ok = v < kWmXV && gh >= 0 && gh < a.hi && gw >= 0 && gw < a.wi && piece * 8 < cx_lim;
    xoff[i] = ok ? ((gh * a.wi + gw) * ldx + cix + piece * 8) * 2 : (int)0x80000000;
  }
#pragma unroll
  for (int i = 0; i < kWmGI / 4; ++i) {
    const int v = (i * 4 + wave) * 16 + (lane >> 2);
    const int sh = v >> 5, sw = v & 31;
    const int gh = h0 + sh, gw = w0 + sw;
    const bool ok = gh < a.ho && gw < a.wo && piece * 8 < cg_lim;
    goff[i] = ok ? ((gh * a.gw + gw) * a.ldg + co_base + piece * 8) * 2 : (int)0x80000000;
  }
  const long long nvx = (long long)a.n * a.di * a.hi * a.wi, nvg = (long long)a.n * a.gd * a.gh * a.gw;
  const dma_rsrc_t rsx = dma_rsrc(xsrc, ((nvx - 1) * ldx + csrc) * 2), rsg = dma_rsrc(a.g, ((nvg - 1) * a.ldg + a.cg) * 2);
  const int xplane = a.hi * a.wi * ldx * 2, gplane = a.gh * a.gw * a.ldg * 2;
  auto load_x = [&](int q) __attribute__((always_inline)) {           // x plane q into ring slot (q + 1) % 5 (outside [0, D): zeros)
    const bool pin = q >= 0 && q < a.di;
    const int soff = pin ? (tn * a.di + q) * xplane : 0;
    const int kill = pin ? 0 : (int)0x80000000;
    char* dst = xs + ((unsigned)(q + 1) % kWmXR) * kWmXS + wave * 1024;
#pragma unroll
    for (int i = 0; i < kWmXI / 4; ++i)
      dma_lds_b128(rsx, (i == 5 && wave >= 2) ? dump + (wave - 2) * 1024 : dst + i * 4096, xoff[i] | kill, soff);
  };
  auto load_g = [&](int p) __attribute__((always_inline)) {           // g plane p into slot p % 3 (outside the segment: zeros, never read)
    const bool pin = p < d1;
    const int soff = pin ? (tn * a.gd + p) * gplane : 0;
    const int kill = pin ? 0 : (int)0x80000000;
    char* dst = gsm + ((unsigned)p % kWmGR) * kWmGS + wave * 1024;
#pragma unroll
    for (int i = 0; i < kWmGI / 4; ++i) dma_lds_b128(rsg, dst + i * 4096, goff[i] | kill, soff);
  };
  load_x(d0 - 1);
  load_x(d0);
  load_x(d0 + 1);
  load_x(d0 + 2);
  load_g(d0);
  load_g(d0 + 1);

  f32x16 acc[5][3];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][k][j] = 0.f;
  const int gi = lane & 15;
  const int lane_off = (8 * h + (gi >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (gi & 3)) * 2;   // ds_read_b64_tr_b16 lane pattern (frag_tr)
  const int seg_lane_off = seg * 16 * 64 + lane_off;
  dma_wait_all();
  __syncthreads();                                                    // the first planes have landed
  // two whole marches (a wave's role never changes; a join of the two plane bodies inside the loop would make hipcc
  // shuffle the 240 accumulator registers); every wave passes the same number of barriers
  auto march = [&](auto halfc) __attribute__((always_inline)) {
    for (int p = d0; p < d1; ++p) {
      // every step issues the same number of copies (past the segment: zeros into free slots), so that "all but the newest
      // kWmPerStep" is the wait for the copies of the step before; ticket y goes behind halo row y: x plane p + 3 (rows
      // 0 .. 5), g plane p + 2 (rows 6 .. 9)
      const int qx = p + 3, pg = p + 2;
      const bool pinx = qx >= 0 && qx < a.di, ping = pg < d1;
      const int soffx = pinx ? (tn * a.di + qx) * xplane : 0, soffg = ping ? (tn * a.gd + pg) * gplane : 0;
      const int killx = pinx ? 0 : (int)0x80000000, killg = ping ? 0 : (int)0x80000000;
      char* const dstx = xs + ((unsigned)(qx + 1) % kWmXR) * kWmXS + wave * 1024;
      char* const dstg = gsm + ((unsigned)pg % kWmGR) * kWmGS + wave * 1024;
      auto ticket = [&](const int y) __attribute__((always_inline)) {
#ifndef WM_DIAG_NO_DMA
        static_assert(kWmXI / 4 + kWmGI / 4 == kWmHR, "one copy ticket per halo row");
        if (y < kWmXI / 4) dma_lds_b128(rsx, (y == 5 && wave >= 2) ? dump + (wave - 2) * 1024 : dstx + y * 4096, xoff[y] | killx, soffx);
        else dma_lds_b128(rsg, dstg + (y - kWmXI / 4) * 4096, goff[y - kWmXI / 4] | killg, soffg);
#endif
      };
      const char* gp = gsm + ((unsigned)p % kWmGR) * kWmGS + seg_lane_off;
#ifndef WM_DIAG_NO_MMA
      wm_plane<decltype(halfc)::value>(acc, xs, gp, p, seg_lane_off, ticket);
#else
#pragma unroll
      for (int y = 0; y < kWmHR; ++y) ticket(y);
#endif
#ifndef WM_DIAG_NO_WAIT
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kWmPerStep) : "memory");   // x plane p + 2, g plane p + 1 have landed (this wave's share) ...
#endif
#ifndef WM_DIAG_NO_BARRIER
      __syncthreads();            // ... everybody's; and every wave is done with this plane's slots
#endif
    }
  };
  if (half == 0) march(std::integral_constant<int, 0>{});
  else march(std::integral_constant<int, 1>{});
  dma_wait_all();                 // the zero copies of the last steps must not land in the parking area
  __syncthreads();
  // ---- partial sums of a tap live in two waves (the two w-segments), those of the shared column 4 in all four: everybody
  //      but the writer parks them in LDS, the writer adds them in a fixed order (bit-identical reruns) and writes the slab
  float* const parkA = reinterpret_cast<float*>(smem) + lane;         // [half][i (5)][kh][j][lane]: the segment-1 waves
  float* const parkB = parkA + 2 * 5 * 48 * 64;                       // [kh][j][lane]: column 4 of (segment 0, half 1)
  if (seg == 1) {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) parkA[(((half * 5 + i) * 3 + k) * 16 + j) * 64] = acc[i][k][j];
  } else if (half == 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 16; ++j) parkB[(k * 16 + j) * 64] = acc[4][k][j];
  }
  __syncthreads();
  if (seg == 0) {
    float* sl = a.slab + (long long)slab * 27 * a.cinp * a.coutp;
    const int co = co_base + r;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (i == 4 && half == 1) break;                                 // column 4 is written by (segment 0, half 0)
      const int u = i == 4 ? 4 : (half ? 5 + i : i), kd = u / 3, kw = u - kd * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int tap = (kd * 3 + k) * 3 + kw;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float v = acc[i][k][j] + parkA[(((half * 5 + i) * 3 + k) * 16 + j) * 64];
          if (i == 4) v = (v + parkB[(k * 16 + j) * 64]) + parkA[(((1 * 5 + 4) * 3 + k) * 16 + j) * 64];
          st_slab(&sl[((long long)tap * a.cinp + ci_base + acc_row(j, h)) * a.coutp + co], v);
        }
      }
    }
  }
}

// Transposed-conv (k2 s2) weight gradient, bf16: D[ci][cls * Cout + co] = sum_v x[v][ci] * g[2v + cls][co], one GEMM
// with 8 * Cout columns.  `wgrad_bf16_kernel<1,..>` staged x once per 32-column tile: for 64->64 at 64^3 that is 16
// re-reads of x and 2 of g (1.06 GB for 0.3 GB of operands, 256 us).  Here a workgroup owns NCO = 4 consecutive
// column tiles: the x tile is staged once, the four g tiles (two parity classes) next to it (80 KB of LDS), each
// wave takes every 4th k-group and feeds one x fragment to four MFMAs.
constexpr int kDeconvNco = 4;
__global__ __launch_bounds__(256, 2) void wgrad_deconv_kernel(const WgradArgs a, int tiles_d, int tiles_h, int tiles_w, int ntiles) {
  constexpr int TD = 2, TH = 4, TW = 32, NCO = kDeconvNco, ROWS = TD * TH * TW, NP = ROWS * 4 / 256, SEGS = TW / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  char* gsm = smem + ROWS * 64;                          // NCO tiles of ROWS x 64 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int ci_base = blockIdx.y * 32, co_base0 = blockIdx.z * (32 * NCO);
  const int cx_lim = a.c0 - ci_base;
  const int gi = lane & 15;
  const int lane_off = (8 * h + (gi >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (gi & 3)) * 2;
  f32x16 acc[NCO];
#pragma unroll
  for (int t = 0; t < NCO; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int q = tile;
    const int tw_i = q % tiles_w; q /= tiles_w;
    const int th_i = q % tiles_h; q /= tiles_h;
    const int td_i = q % tiles_d;
    const int n = q / tiles_d;
    const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;
    uint4 sx[NP], sg[NCO][NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * 256;
      const int row = p >> 2, part = p & 3;
      const int sw = row % TW, sh = (row / TW) % TH, sd = row / (TW * TH);
      const int gd = d0 + sd, gh = h0 + sh, gw = w0 + sw;
      const bool in = gd < a.do_ && gh < a.ho && gw < a.wo;
      sx[i] = (in && part * 8 < cx_lim)
                  ? *reinterpret_cast<const uint4*>(a.x0 + (((((long long)n * a.di + gd) * a.hi + gh) * a.wi + gw) * (long long)a.ld0 + ci_base + part * 8) * 2)
                  : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NCO; ++t) {
        const int co_base = co_base0 + t * 32;
        const int cls = co_base / a.g_cls_cout, gco = co_base - cls * a.g_cls_cout;
        const int od = cls >> 2, oh = (cls >> 1) & 1, ow = cls & 1;
        sg[t][i] = (in && part * 8 < a.g_cls_cout - gco)
                       ? *reinterpret_cast<const uint4*>(a.g + (((((long long)n * a.gd + (gd * 2 + od)) * a.gh + (gh * 2 + oh)) * a.gw + (gw * 2 + ow)) * (long long)a.ldg + gco + part * 8) * 2)
                       : make_uint4(0, 0, 0, 0);
      }
    }
    __syncthreads();                      // every wave finished reading the previous tile
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * 256;
      *reinterpret_cast<uint4*>(xs + p * 16) = sx[i];
#pragma unroll
      for (int t = 0; t < NCO; ++t) *reinterpret_cast<uint4*>(gsm + t * (ROWS * 64) + p * 16) = sg[t][i];
    }
    __syncthreads();
#pragma unroll 2
    for (int kg = wave; kg < TD * TH * SEGS; kg += 4) {
      const int off = kg * 16 * 64 + lane_off;            // k-groups are consecutive runs of 16 rows
      const bf16x8 af = frag_tr(xs + off);
#pragma unroll
      for (int t = 0; t < NCO; ++t) {
        const bf16x8 b = frag_tr(gsm + t * (ROWS * 64) + off);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b, acc[t], 0, 0, 0);
      }
    }
  }
  float* sl = a.slab + (long long)(blockIdx.x * 4 + wave) * a.cinp * a.coutp;
#pragma unroll
  for (int t = 0; t < NCO; ++t) {
    const int co = co_base0 + t * 32 + r;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = ci_base + acc_row(j, h);
      st_slab(&sl[(long long)row * a.coutp + co], acc[t][j]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// wgrad_march2_kernel: weight gradient of the dense 2x2x2 stride-1 convolutions on wide tensors -- the PatchGAN's strided blocks on
// space-to-depth operands (src/model.py:72-82) and the composite kernel of UpCat's fused up-branch (upcat.hip) --
// marching along d like wgrad_march_kernel.  wgrad_bf16_kernel<2, 2, 4, 32> staged a 3 x 5 x 33 halo per 2 x 4 x 32 outputs through
// registers with two barriers per tile: 0.15 of the MFMA peak on d1, whose operands alone need 1/4 of its time.
//   * workgroup = (32 x 32 (ci, co) tile pair, 8 (h) x 32 (w) footprint, segment of output planes); x planes (9 x 33 halo voxels) and
//     g planes arrive by LDS-DMA two planes ahead: ring of 4 x planes (output plane p pairs with x planes p, p + 1; p + 2 and
//     p + 3 in flight) and 3 g planes, 128 KB; one barrier per plane with a counted wait for the older copy;
//   * waves = (w-segment of 16 positions) x (kd): four taps (kh, kw) each -- 32 MFMAs per plane and wave; the 8 g fragments of the
//     segment stay in registers for the plane, an x fragment of halo row y feeds kh = 0 (output row y) and kh = 1 (row y - 1);
//   * at the end the two w-segments of a tap are added through LDS in a fixed order: ONE slab per workgroup, reduced in slab
//     order by the kernels above: bit-identical reruns.
constexpr int kW2FH = 8, kW2FW = 32, kW2HR = kW2FH + 1, kW2HC = kW2FW + 1;
constexpr int kW2XI = 20, kW2XS = kW2XI * 1024;       // x plane: 297 halo voxels x 64 B = 18.6 KB -> 20 DMA instructions, 5 per wave
constexpr int kW2GI = 16, kW2GS = kW2GI * 1024;       // g plane: 256 voxels x 64 B
constexpr int kW2XR = 4, kW2GR = 3;
constexpr int kW2Lds = kW2XR * kW2XS + kW2GR * kW2GS;  // 128 KB
constexpr int kW2PerStep = kW2XI / 4 + kW2GI / 4;     // 9 copies per wave and step

__global__ __launch_bounds__(256, 1) void wgrad_march2_kernel(const WgradArgs a, const WMarchArgs m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const xs = smem;
  char* const gsm = smem + kW2XR * kW2XS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int seg = wave & 1, kd = wave >> 1;
  const int tpairs = m.ci_tiles * m.co_tiles;
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int slab = (kq / tpairs) * 8 + xcd, pair = kq % tpairs;       // (the tile pairs of a slab are neighbours on one XCD: wgrad_march_kernel)
  if (slab >= m.nslabs) return;
  const int ci_base = (pair % m.ci_tiles) * 32, co_base = (pair / m.ci_tiles) * 32;
  const int cx_lim = a.c0 - ci_base, cg_lim = a.cg - co_base;
  const int per_seg = m.tiles_h * m.tiles_w, per_sample = per_seg * m.nseg;
  int t = slab;
  const int tn = t / per_sample;
  t -= tn * per_sample;
  const int sg = t / per_seg;
  t -= sg * per_seg;
  const int th_i = t / m.tiles_w, tw_i = t - th_i * m.tiles_w;
  const int d0 = sg * m.seg_len, d1 = min(a.do_, d0 + m.seg_len);
  const int h0 = th_i * kW2FH, w0 = tw_i * kW2FW;

  const int piece = lane & 3;
  int xoff[kW2XI / 4], goff[kW2GI / 4];
#pragma unroll
  for (int i = 0; i < kW2XI / 4; ++i) {
    const int v = (i * 4 + wave) * 16 + (lane >> 2);
    const int y = v / kW2HC, col = v - y * kW2HC;
    const int gh = h0 + y, gw = w0 + col;
    const bool ok = v < kW2HR * kW2HC && gh < a.hi && gw < a.wi && piece * 8 < cx_lim;
    xoff[i] = ok ? ((gh * a.wi + gw) * a.ld0 + ci_base + piece * 8) * 2 : (int)0x80000000;
  }
#pragma unroll
  for (int i = 0; i < kW2GI / 4; ++i) {
    const int v = (i * 4 + wave) * 16 + (lane >> 2);
    const int sh = v >> 5, sw = v & 31;
    const int gh = h0 + sh, gw = w0 + sw;
    const bool ok = gh < a.ho && gw < a.wo && piece * 8 < cg_lim;
    goff[i] = ok ? ((gh * a.gw + gw) * a.ldg + co_base + piece * 8) * 2 : (int)0x80000000;
  }
  const long long nvx = (long long)a.n * a.di * a.hi * a.wi, nvg = (long long)a.n * a.gd * a.gh * a.gw;
  const dma_rsrc_t rsx = dma_rsrc(a.x0, ((nvx - 1) * a.ld0 + a.c0) * 2), rsg = dma_rsrc(a.g, ((nvg - 1) * a.ldg + a.cg) * 2);
  const int xplane = a.hi * a.wi * a.ld0 * 2, gplane = a.gh * a.gw * a.ldg * 2;
  auto load_x = [&](int q) __attribute__((always_inline)) {           // x plane q -> ring slot q % 4 (beyond the tensor: zeros)
    const bool pin = q < a.di;
    const int soff = pin ? (tn * a.di + q) * xplane : 0, kill = pin ? 0 : (int)0x80000000;
    char* dst = xs + ((unsigned)q % kW2XR) * kW2XS + wave * 1024;
#pragma unroll
    for (int i = 0; i < kW2XI / 4; ++i) dma_lds_b128(rsx, dst + i * 4096, xoff[i] | kill, soff);
  };
  auto load_g = [&](int p) __attribute__((always_inline)) {
    const bool pin = p < d1;
    const int soff = pin ? (tn * a.gd + p) * gplane : 0, kill = pin ? 0 : (int)0x80000000;
    char* dst = gsm + ((unsigned)p % kW2GR) * kW2GS + wave * 1024;
#pragma unroll
    for (int i = 0; i < kW2GI / 4; ++i) dma_lds_b128(rsg, dst + i * 4096, goff[i] | kill, soff);
  };
  load_x(d0);
  load_x(d0 + 1);
  load_x(d0 + 2);
  load_g(d0);
  load_g(d0 + 1);

  f32x16 acc[2][2];                                                    // [kh][kw]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][k][j] = 0.f;
  const int gi = lane & 15;
  const int lane_off = (8 * h + (gi >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (gi & 3)) * 2;     // frag_tr's lane pattern
  const int seg_lane_off = seg * 16 * 64 + lane_off;
  dma_wait_all();
  __syncthreads();
  for (int p = d0; p < d1; ++p) {
    // this step's copies: x plane p + 3 (tickets 0 .. 4), g plane p + 2 (tickets 5 .. 8), one behind each halo row's MFMAs
    const int qx = p + 3, pg = p + 2;
    const bool pinx = qx < a.di, ping = pg < d1;
    const int soffx = pinx ? (tn * a.di + qx) * xplane : 0, soffg = ping ? (tn * a.gd + pg) * gplane : 0;
    const int killx = pinx ? 0 : (int)0x80000000, killg = ping ? 0 : (int)0x80000000;
    char* const dstx = xs + ((unsigned)qx % kW2XR) * kW2XS + wave * 1024;
    char* const dstg = gsm + ((unsigned)pg % kW2GR) * kW2GS + wave * 1024;
    const char* gp = gsm + ((unsigned)p % kW2GR) * kW2GS + seg_lane_off;
    const char* xp = xs + ((unsigned)(p + kd) % kW2XR) * kW2XS + seg_lane_off;
    bf16x8 gf[kW2FH];
#pragma unroll
    for (int sh = 0; sh < kW2FH; ++sh) gf[sh] = frag_tr(gp + sh * (kW2FW * 64));
    bf16x8 xf[2][2];
#pragma unroll
    for (int kw = 0; kw < 2; ++kw) xf[0][kw] = frag_tr(xp + kw * 64);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * kW2FH + 4, 0);
#pragma unroll
    for (int y = 0; y < kW2HR; ++y) {
      if (y + 1 < kW2HR) {
#pragma unroll
        for (int kw = 0; kw < 2; ++kw) xf[(y + 1) & 1][kw] = frag_tr(xp + (y + 1) * (kW2HC * 64) + kw * 64);
      }
      int nm = 0;
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int row = y - kh;
        if (row >= 0 && row < kW2FH) {
#pragma unroll
          for (int kw = 0; kw < 2; ++kw) acc[kh][kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[y & 1][kw], gf[row], acc[kh][kw], 0, 0, 0);
          nm += 2;
        }
      }
      // the row's MFMAs with the next row's four transposed reads spread between them
      const int nrd = y + 1 < kW2HR ? 4 : 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        const int nr = nm ? (nrd * (k + 1) / nm - nrd * k / nm) : 0;
        if (k < nm && nr == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        else if (k < nm && nr == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      static_assert(kW2PerStep == kW2HR, "one copy ticket per halo row");
      if (y < kW2XI / 4) dma_lds_b128(rsx, dstx + y * 4096, xoff[y] | killx, soffx);
      else dma_lds_b128(rsg, dstg + (y - kW2XI / 4) * 4096, goff[y - kW2XI / 4] | killg, soffg);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kW2PerStep) : "memory");   // x plane p + 2, g plane p + 1 have landed (this wave's share) ...
    __syncthreads();                                                    // ... everybody's; every wave is done with this plane's slots
  }
  dma_wait_all();
  __syncthreads();
  // ---- the two w-segments of a tap: segment 1 parks, segment 0 adds (fixed order) and writes the workgroup's slab
  float* const park = reinterpret_cast<float*>(smem) + lane;           // [kd][kh][kw][j][lane]
  if (seg == 1) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int kw = 0; kw < 2; ++kw)
#pragma unroll
        for (int j = 0; j < 16; ++j) park[((((kd * 2 + kh) * 2 + kw) * 16) + j) * 64] = acc[kh][kw][j];
  }
  __syncthreads();
  if (seg == 0) {
    float* sl = a.slab + (long long)slab * 8 * a.cinp * a.coutp;
    const int co = co_base + r;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) {
        const int tap = (kd * 2 + kh) * 2 + kw;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float v = acc[kh][kw][j] + park[((((kd * 2 + kh) * 2 + kw) * 16) + j) * 64];
          st_slab(&sl[((long long)tap * a.cinp + ci_base + acc_row(j, h)) * a.coutp + co], v);
        }
      }
  }
}

// ------------------------------------------------------------------------------------------
// wgrad_pw_kernel: weight gradient of the full-resolution 1x1x1 convolutions with <= 32 channels either side (generator
// head 24 -> 24, src/model.py:21; final_conv 32 -> 6, MONAI BasicUNet at :22-28): D[ci][co] = sum_v x[v][ci] g[v][co] over
// ~2 M voxels -- 64 + 32..64 bytes per voxel for 2 K FLOP, a pure HBM stream.  wgrad_bf16_kernel<1, ...> staged 256-voxel
// tiles through registers with two barriers per tile and 4 MFMAs per wave between them: 64 - 73 us for 200 - 270 MB
// (0.4 - 0.5 of the HBM roof).  Here every WAVE streams its own 64-voxel pieces: x and g rows (4 KB + 4 KB)
// arrive by LDS-DMA in a wave-private ring of three stages (two in flight), are read back transposed
// (ds_read_b64_tr_b16) and feed 4 MFMAs; no workgroup barrier in the loop, counted waits (the copies are inline assembly).
// One workgroup per CU (96 KB of LDS): 64 KB of loads in flight per CU.  One f32 slab per workgroup (its 4 waves are
// added through LDS in wave order), reduced by wgrad_reduce_kernel in slab order: deterministic.
constexpr int kPwStages = 3, kPwStageB = 8192, kPwLds = 4 * kPwStages * kPwStageB;

__global__ __launch_bounds__(256, 1) void wgrad_pw_kernel(const WgradArgs a, const long long nvox, const int nstage) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  char* const ring = smem + wave * (kPwStages * kPwStageB);
  const dma_rsrc_t rsx = dma_rsrc(a.x0, ((nvox - 1) * a.ld0 + a.c0) * 2), rsg = dma_rsrc(a.g, ((nvox - 1) * a.ldg + a.cg) * 2);
  // lane = (voxel of a 16-voxel instruction, 16-byte piece); channels past the tensor's width: zeros
  const int piece = lane & 3, lv = lane >> 2;
  const bool xok = piece * 8 < a.c0, gok = piece * 8 < a.cg;
  const int xlane = (lv * a.ld0 + piece * 8) * 2, glane = (lv * a.ldg + piece * 8) * 2;
  const int gi = lane & 15;
  const int lane_off = (8 * h + (gi >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (gi & 3)) * 2;       // frag_tr's lane pattern
  const int nw = gridDim.x * 4, first = blockIdx.x * 4 + wave;       // stage s of this wave = voxels [64 (first + s nw), + 64)
  auto issue = [&](int s, int slot) __attribute__((always_inline)) {
    const long long v0 = (long long)(first + (long long)s * nw) * 64;
    const bool in = s >= 0 && v0 < nvox;                              // (past the end: zero fills, same instruction count)
    char* dst = ring + slot * kPwStageB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long vb = v0 + i * 16;
      const bool ok = in && vb + lv < nvox;
      dma_lds_b128(rsx, dst + i * 1024, (ok && xok) ? (int)(vb * a.ld0 * 2) + xlane : (int)0x80000000, 0);
      dma_lds_b128(rsg, dst + 4096 + i * 1024, (ok && gok) ? (int)(vb * a.ldg * 2) + glane : (int)0x80000000, 0);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;
  const int mine = first < nstage ? (nstage - first + nw - 1) / nw : 0;     // stages of this wave
  issue(0, 0);
  issue(1, 1);
  for (int s = 0; s < mine; ++s) {
    issue(s + 2, (s + 2) % kPwStages);                                       // (slot of stage s - 1: its reads fed MFMAs already issued)
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                        // all but the two youngest stages (8 copies each) have landed
    __builtin_amdgcn_sched_barrier(0);
    const char* st = ring + (s % kPwStages) * kPwStageB + lane_off;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bf16x8 af = frag_tr(st + k * 1024), bfg = frag_tr(st + 4096 + k * 1024);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfg, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  dma_wait_all();
  __syncthreads();
  float* park = reinterpret_cast<float*>(smem);                              // [wave][j][lane]
#pragma unroll
  for (int j = 0; j < 16; ++j) park[(wave * 16 + j) * 64 + lane] = acc[j];
  __syncthreads();
  if (wave == 0) {
    float* sl = a.slab + (long long)blockIdx.x * a.cinp * a.coutp;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float v = park[j * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) v += park[(w * 16 + j) * 64 + lane];
      st_slab(&sl[(long long)acc_row(j, h) * a.coutp + r], v);
    }
  }
}

struct WPlan { int ks, tpw, tap_groups, ci_tiles, co_tiles, splits, cinp32, coutp32; long long rows;
               bool fast, deconv4; int shape, tiles_d, tiles_h, tiles_w, ntiles, nslabs;
               bool march; int seg_len, nseg; bool pw; bool march2; };

const int kWTD[2] = {2, 2}, kWTH[2] = {4, 8}, kWTW[2] = {32, 16};

// planner constant of the tile path as a diagnostic-build knob (tools/sweep_plan.sh)
#ifdef MI355_DIAG
int tune_wg_target() { static const int v = [] { const char* e = getenv("MI355_WG_TARGET"); return e ? atoi(e) : 512; }(); return v; }
#else
constexpr int tune_wg_target() { return 512; }
#endif

int wplan(const mi355_wgrad_desc* d, WPlan* p) {
  MI355_REQUIRE(d && d->x0 && d->g && d->dw, "wgrad: null pointer");
  MI355_REQUIRE(d->dtype == MI355_DT_F32 || d->dtype == MI355_DT_BF16, "wgrad: bad dtype");
  MI355_REQUIRE(d->ks >= 1 && d->ks <= 4, "wgrad: unsupported ks=%d", d->ks);
  MI355_REQUIRE(d->c0 > 0 && d->c0 % 16 == 0 && d->c1 % 16 == 0 && d->cg % 16 == 0, "wgrad: channels must be multiples of 16");
  MI355_REQUIRE(d->c1 == 0 || (d->x1 && d->c0 % 32 == 0), "wgrad: concat split must be a multiple of 32");
  MI355_REQUIRE(d->cin <= d->c0 + d->c1 && d->cout <= d->cg, "wgrad: real extents exceed padded ones");
  MI355_REQUIRE(d->s2d_cp == 0 || (d->s2d_cp % 8 == 0 && d->c1 == 0 && d->c0 == 8 * d->s2d_cp && d->cin <= d->s2d_cp),
                "wgrad: bad space-to-depth operand");
  MI355_REQUIRE(d->xn >= 0 && (d->xn == 0 || d->n % d->xn == 0), "wgrad: xn must divide n");
  p->ks = d->ks;
  p->tpw = d->ks == 1 ? 1 : (d->ks == 3 ? 9 : 8);
  const int nt = d->ks * d->ks * d->ks;
  p->tap_groups = (nt + p->tpw - 1) / p->tpw;
  p->cinp32 = ((d->c0 + d->c1 + 31) / 32) * 32;
  p->coutp32 = d->g_cls_cout > 0 ? 8 * d->g_cls_cout : ((d->cg + 31) / 32) * 32;
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
  p->nslabs = p->splits * 4;
  // fast path: bf16, 3x3x3 stride 1, g on the same grid as the outputs
  const bool cls = d->g_cls_cout > 0;
  MI355_REQUIRE(!cls || (d->dtype == MI355_DT_BF16 && d->ks == 1 && d->stride == 1 && d->g_cls_cout % 32 == 0 &&
                         d->cg >= d->g_cls_cout && d->gd == 2 * d->do_ && d->gh == 2 * d->ho && d->gw == 2 * d->wo &&
                         d->c1 == 0 && d->cout <= d->g_cls_cout),
                "wgrad: bad transposed-conv class folding");
  p->fast = d->dtype == MI355_DT_BF16 && d->stride == 1 && d->ld0 % 8 == 0 && (d->c1 == 0 || d->ld1 % 8 == 0) && d->ldg % 8 == 0 &&
            (cls || (d->ks <= 3 && d->gs == 1 && d->goff[0] == 0 && d->goff[1] == 0 && d->goff[2] == 0 &&
                     d->gd == d->do_ && d->gh == d->ho && d->gw == d->wo));
  if (p->fast) {
    p->shape = d->wo > 16 ? 0 : 1;
    p->tiles_d = ceil_div(d->do_, kWTD[p->shape]);
    p->tiles_h = ceil_div(d->ho, kWTH[p->shape]);
    p->tiles_w = ceil_div(d->wo, kWTW[p->shape]);
    p->ntiles = p->tiles_d * p->tiles_h * p->tiles_w * d->n;
    const int wsl = d->ks == 1 ? 4 : 1;                     // slabs per workgroup
    // transposed conv at the wide levels: 4 column tiles per workgroup (wgrad_deconv_kernel)
    p->deconv4 = cls && p->shape == 0 && p->co_tiles % kDeconvNco == 0 && d->do_ % 2 == 0 && d->ho % 4 == 0 && d->wo % 32 == 0;
    const long long cot = p->deconv4 ? p->co_tiles / kDeconvNco : p->co_tiles;
    long long sp = tune_wg_target() / ((long long)p->ci_tiles * cot);      // ~2 workgroups per CU (256, 1024 measured slower)
    if (sp < 1) sp = 1;
    if (sp > p->ntiles) sp = p->ntiles;
    while (sp > 1 && sp * wsl * slab_bytes > (256ll << 20)) sp /= 2;
    p->splits = (int)sp;
    p->nslabs = d->ks == 1 ? p->splits * 4 : p->splits;
  }
  // streaming 1x1x1 kernel: one (ci, co) tile, plain tensors on one grid, many voxels; 32-bit byte offsets
  p->pw = false;
  {
    const long long nv = (long long)d->n * d->di * d->hi * d->wi;
    if (p->fast && !cls && d->ks == 1 && d->c1 == 0 && d->c0 <= 32 && d->cg <= 32 && d->pad[0] == 0 && d->pad[1] == 0 && d->pad[2] == 0 &&
        d->di == d->do_ && d->hi == d->ho && d->wi == d->wo && d->s2d_cp == 0 && d->xn == 0 && nv >= (1ll << 17) &&
        nv * std::max(d->ld0, d->ldg) * 2 < (1ll << 31)) {
      p->pw = true;
      p->splits = 256;
      p->nslabs = 256;
    }
  }
  // marching k2 kernel: dense 2x2x2, padding 0, x one larger than the grid, wide rows; 32-bit byte offsets
  p->march2 = false;
  if (p->fast && !cls && d->ks == 2 && d->c1 == 0 && d->pad[0] == 0 && d->pad[1] == 0 && d->pad[2] == 0 && d->di == d->do_ + 1 &&
      d->hi == d->ho + 1 && d->wi == d->wo + 1 && d->wo >= 32 && d->do_ >= 4 && d->xn == 0 &&
      (long long)d->n * d->di * d->hi * d->wi * std::max(d->ld0, d->ldg) * 2 < (1ll << 31)) {
    const int th = ceil_div(d->ho, kW2FH), tw = ceil_div(d->wo, kW2FW);
    const long long base = (long long)d->n * th * tw * p->ci_tiles * p->co_tiles;
    long long best = -1; int best_ns = 1;
    for (int ns = 1; ns <= d->do_ / 2; ++ns) {
      const int L = ceil_div(d->do_, ns);
      if ((long long)(ns - 1) * L >= d->do_) continue;
      const long long wgs = base * ns, cost = ((wgs + 255) / 256) * (L + 3);
      if ((long long)ns * d->n * th * tw * slab_bytes > (256ll << 20)) break;
      if (best < 0 || cost < best) { best = cost; best_ns = ns; }
    }
    p->march2 = true;
    p->nseg = best_ns;
    p->seg_len = ceil_div(d->do_, best_ns);
    p->tiles_h = th; p->tiles_w = tw;
    p->nslabs = d->n * th * tw * best_ns;
  }
  // marching kernel: 3x3x3, padding 1, same extents, wide rows; 32-bit byte offsets
  p->march = false;
  if (p->fast && !cls && d->ks == 3 && d->pad[0] == 1 && d->pad[1] == 1 && d->pad[2] == 1 && d->di == d->do_ && d->hi == d->ho &&
      d->wi == d->wo && d->wo >= 32 && d->do_ >= 8 &&
      (long long)d->n * d->di * d->hi * d->wi * std::max(d->ld0, std::max(d->ld1, d->ldg)) * 2 < (1ll << 31)) {
    const int th = ceil_div(d->ho, kWmFH), tw = ceil_div(d->wo, kWmFW);
    const long long base = (long long)d->n * th * tw * p->ci_tiles * p->co_tiles;
    // segments: whole rounds of 256 workgroups (one per CU), cost = rounds x (planes per segment + pipeline fill)
    long long best = -1; int best_ns = 1;
    for (int ns = 1; ns <= d->do_ / 4; ++ns) {
      const int L = ceil_div(d->do_, ns);
      if ((long long)(ns - 1) * L >= d->do_) continue;
      const long long wgs = base * ns, cost = ((wgs + 255) / 256) * (L + 3);
      if ((long long)ns * d->n * th * tw * slab_bytes > (256ll << 20)) break;
      if (best < 0 || cost < best) { best = cost; best_ns = ns; }
    }
    p->march = true;
    p->nseg = best_ns;
    p->seg_len = ceil_div(d->do_, best_ns);
    p->tiles_h = th; p->tiles_w = tw;
    p->nslabs = d->n * th * tw * best_ns;
  }
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

constexpr int kRedMagic = 0x52454431;                                 // "RED1"
struct RedJob { WreduceArgs q; int kind, gx, gy, magic; };            // what mi355_wreduce_job holds

void make_reduce_job(const mi355_wgrad_desc* d, const WPlan& p, RedJob* job) {
  WreduceArgs& q = job->q;
  q.slab = d->workspace; q.nslabs = p.nslabs; q.ks = d->ks; q.ntaps = d->ks * d->ks * d->ks;
  q.cinp = p.cinp32; q.coutp = p.coutp32;
  q.dw = d->dw; q.cout = d->cout; q.cin = d->cin;
  q.s_co = d->s_co; q.s_ci = d->s_ci; q.s_k0 = d->s_k[0]; q.s_k1 = d->s_k[1]; q.s_k2 = d->s_k[2];
  q.tb0 = d->tbase[0]; q.tb1 = d->tbase[1]; q.tb2 = d->tbase[2];
  q.ts0 = d->tstep[0]; q.ts1 = d->tstep[1]; q.ts2 = d->tstep[2];
  q.accumulate = d->accumulate;
  q.s2d_cp = d->s2d_cp;
  q.co_cls = d->g_cls_cout;
  job->magic = kRedMagic;
  const long long per = (long long)q.ntaps * q.cinp * q.coutp;
  const int k3 = d->ks * d->ks * d->ks;
  const bool dense = !q.s2d_cp && !q.co_cls && q.s_k2 == 1 && q.s_k1 == d->ks && q.s_k0 == d->ks * d->ks && q.s_ci == k3 &&
                     q.tb0 == 0 && q.tb1 == 0 && q.tb2 == 0 && q.ts0 == 1 && q.ts1 == 1 && q.ts2 == 1;
  if (dense && q.nslabs <= 16 && (d->ks == 3 || d->ks == 1) && q.cinp % 8 == 0 && q.coutp % 32 == 0) {
    // input channels per block: as many as keep >= 512 blocks in flight
    const long long cob = q.coutp / 32;
    const int cib = (q.cinp / 8) * cob >= 512 ? 8 : ((q.cinp / 4) * cob >= 512 ? 4 : 2);
    job->gx = q.cinp / cib; job->gy = (int)cob;
    job->kind = (d->ks == 3 ? kRedDense27_8 : kRedDense1_8) + (cib == 8 ? 0 : (cib == 4 ? 1 : 2));
    return;
  }
  const bool s2d_dense = q.s2d_cp > 0 && !q.co_cls && d->ks == 2 && q.s_k2 == 1 && q.s_k1 == 4 && q.s_k0 == 16 && q.s_ci == 64 &&
                         q.tb0 == 0 && q.tb1 == 0 && q.tb2 == 0 && q.ts0 == 2 && q.ts1 == 2 && q.ts2 == 2 &&
                         q.cinp == 8 * q.s2d_cp && q.s2d_cp % 4 == 0 && q.coutp % 32 == 0;
  if (s2d_dense && q.nslabs <= 8) {
    job->kind = kRedS2d; job->gx = q.s2d_cp / 4; job->gy = q.coutp / 32;
    return;
  }
  job->gy = 1;
  if (per <= 16384 && q.nslabs >= 128) { job->kind = kRedGenericNarrow; job->gx = (int)((per + 31) / 32); }
  else { job->kind = kRedGeneric; job->gx = (int)((per + 127) / 128); }
}

int launch_reduce_single(const RedJob& job, hipStream_t st) {
  const WreduceArgs& q = job.q;
  const dim3 grid((unsigned)job.gx, (unsigned)job.gy);
  switch (job.kind) {
    case kRedGeneric: wgrad_reduce_kernel<32, 8><<<grid, dim3(256), 0, st>>>(q); return mi355_check_launch("wgrad_reduce");
    case kRedGenericNarrow: wgrad_reduce_kernel<8, 32><<<grid, dim3(256), 0, st>>>(q); return mi355_check_launch("wgrad_reduce");
    case kRedDense27_8: wgrad_reduce_dense_kernel<27, 8><<<grid, dim3(256), 0, st>>>(q); break;
    case kRedDense27_4: wgrad_reduce_dense_kernel<27, 4><<<grid, dim3(256), 0, st>>>(q); break;
    case kRedDense27_2: wgrad_reduce_dense_kernel<27, 2><<<grid, dim3(256), 0, st>>>(q); break;
    case kRedDense1_8: wgrad_reduce_dense_kernel<1, 8><<<grid, dim3(256), 0, st>>>(q); break;
    case kRedDense1_4: wgrad_reduce_dense_kernel<1, 4><<<grid, dim3(256), 0, st>>>(q); break;
    case kRedDense1_2: wgrad_reduce_dense_kernel<1, 2><<<grid, dim3(256), 0, st>>>(q); break;
    default: wgrad_reduce_s2d_kernel<<<grid, dim3(256), 0, st>>>(q); return mi355_check_launch("wgrad_reduce_s2d");
  }
  return mi355_check_launch("wgrad_reduce_dense");
}

int conv_wgrad_impl(const mi355_wgrad_desc* d, mi355_wreduce_job* deferred, void* stream);

}  // namespace

extern "C" int64_t mi355_conv_wgrad_workspace(const mi355_wgrad_desc* d) {
  WPlan p;
  if (wplan(d, &p)) return -1;
  return (int64_t)p.nslabs * p.ks * p.ks * p.ks * p.cinp32 * p.coutp32 * 4;
}

extern "C" int mi355_conv_wgrad_plan_kind(const mi355_wgrad_desc* d) {
  WPlan p;
  if (wplan(d, &p)) return -1;
  return p.march ? 2 : (p.pw ? 3 : (p.march2 ? 4 : (p.fast ? 1 : 0)));
}

namespace {
int conv_wgrad_impl(const mi355_wgrad_desc* d, mi355_wreduce_job* deferred, void* stream) {
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
  a.g_cls_cout = d->g_cls_cout;
  a.xn = d->xn > 0 ? d->xn : d->n;
  MI355_REQUIRE(a.xn == d->n || (!p.march && !(p.fast && p.deconv4)), "wgrad: xn is not supported by this plan");
  if (p.pw) {
    const long long nv = (long long)d->n * d->di * d->hi * d->wi;
    static const int attr = (int)hipFuncSetAttribute((const void*)wgrad_pw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kPwLds);
    if (attr) { mi355_set_error("wgrad_pw: cannot raise the dynamic LDS limit to %d bytes (hip error %d)", kPwLds, attr); return MI355_ERR_HIP; }
    wgrad_pw_kernel<<<dim3(256), dim3(256), kPwLds, st>>>(a, nv, (int)((nv + 63) / 64));
  } else if (p.march2) {
    const WMarchArgs m{p.seg_len, p.nseg, p.tiles_h, p.tiles_w, p.nslabs, p.ci_tiles, p.co_tiles};
    static const int attr = (int)hipFuncSetAttribute((const void*)wgrad_march2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kW2Lds);
    if (attr) { mi355_set_error("wgrad_march2: cannot raise the dynamic LDS limit to %d bytes (hip error %d)", kW2Lds, attr); return MI355_ERR_HIP; }
    wgrad_march2_kernel<<<dim3(((p.nslabs + 7) / 8) * 8 * p.ci_tiles * p.co_tiles), dim3(256), kW2Lds, st>>>(a, m);
  } else if (p.march) {
    const WMarchArgs m{p.seg_len, p.nseg, p.tiles_h, p.tiles_w, p.nslabs, p.ci_tiles, p.co_tiles};
    static const int attr = (int)hipFuncSetAttribute((const void*)wgrad_march_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kWmLds);
    if (attr) { mi355_set_error("wgrad_march: cannot raise the dynamic LDS limit to %d bytes (hip error %d)", kWmLds, attr); return MI355_ERR_HIP; }
    wgrad_march_kernel<<<dim3(((p.nslabs + 7) / 8) * 8 * p.ci_tiles * p.co_tiles), dim3(256), kWmLds, st>>>(a, m);
  } else if (p.fast && p.deconv4) {
    const dim3 grid(p.splits, p.ci_tiles, p.co_tiles / kDeconvNco);
    wgrad_deconv_kernel<<<grid, dim3(256), (1 + kDeconvNco) * 256 * 64, st>>>(a, p.tiles_d, p.tiles_h, p.tiles_w, p.ntiles);
  } else if (p.fast) {
    dim3 grid(p.splits, p.ci_tiles, p.co_tiles), block(256);
#define WG_FAST(KS, TD, TH, TW)                                                                   \
  do {                                                                                            \
    constexpr int lds = ((TD + KS - 1) * (TH + KS - 1) * (TW + KS - 1) + TD * TH * TW) * 64;      \
    wgrad_bf16_kernel<KS, TD, TH, TW><<<grid, block, lds, st>>>(a, p.tiles_d, p.tiles_h, p.tiles_w, p.ntiles); \
  } while (0)
    if (d->ks == 3) { if (p.shape == 0) WG_FAST(3, 2, 4, 32); else WG_FAST(3, 2, 8, 16); }
    else if (d->ks == 2) { if (p.shape == 0) WG_FAST(2, 2, 4, 32); else WG_FAST(2, 2, 8, 16); }
    else { if (p.shape == 0) WG_FAST(1, 2, 4, 32); else WG_FAST(1, 2, 8, 16); }
#undef WG_FAST
  } else if (d->dtype == MI355_DT_F32) launch_wgrad<float>(a, p, st);
  else launch_wgrad<bf16_t>(a, p, st);
  rc = mi355_check_launch("conv_wgrad");
  if (rc) return rc;
  RedJob job;
  make_reduce_job(d, p, &job);
  if (deferred) {
    static_assert(sizeof(RedJob) <= sizeof(mi355_wreduce_job), "mi355_wreduce_job is too small");
    memset(deferred, 0, sizeof(*deferred));
    memcpy(deferred, &job, sizeof(job));
    return MI355_OK;
  }
  return launch_reduce_single(job, st);
}

}  // namespace

extern "C" int mi355_conv_wgrad(const mi355_wgrad_desc* d, void* stream) { return conv_wgrad_impl(d, nullptr, stream); }

extern "C" int mi355_conv_wgrad_partial(const mi355_wgrad_desc* d, mi355_wreduce_job* job, void* stream) {
  MI355_REQUIRE(job, "wgrad_partial: null job");
  return conv_wgrad_impl(d, job, stream);
}

extern "C" int mi355_wgrad_reduce_multi(const mi355_wreduce_job* jobs, int32_t n, void* stream) {
  MI355_REQUIRE(n >= 0 && (n == 0 || jobs), "wgrad_reduce_multi: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  std::vector<RedJob> light, heavy;
  for (int j = 0; j < n; ++j) {
    RedJob job;
    memcpy(&job, jobs + j, sizeof(job));
    MI355_REQUIRE(job.magic == kRedMagic && job.kind >= kRedGeneric && job.kind <= kRedS2d && job.gx > 0 && job.gy > 0,
                  "wgrad_reduce_multi: job %d was not filled by mi355_conv_wgrad_partial", j);
#ifdef WGRAD_DIAG_ONE_MULTI      // (timing A/B: every job in the union kernel, as the first version of this call)
    heavy.push_back(job);
#else
    (job.kind == kRedGeneric ? light : heavy).push_back(job);
#endif
  }
  for (int pass = 0; pass < 2; ++pass) {
    const std::vector<RedJob>& v = pass == 0 ? light : heavy;
    for (size_t j0 = 0; j0 < v.size(); j0 += kRedChunk) {
      WreduceMulti m;
      m.n = (int)std::min<size_t>(kRedChunk, v.size() - j0);
      long long blocks = 0;
      for (int j = 0; j < kRedChunk; ++j) {
        const RedJob& job = v[j0 + std::min(j, m.n - 1)];                // (unused entries repeat the last job; never selected)
        m.a[j] = job.q; m.kind[j] = job.kind; m.gx[j] = job.gx;
        m.first[j] = (int)blocks;
        if (j < m.n) blocks += pass == 0 ? (((long long)job.q.ntaps * job.q.cinp * job.q.coutp + LIGHT_F4 * 4 - 1) / (LIGHT_F4 * 4)) : (long long)job.gx * job.gy;
      }
      m.first[kRedChunk] = (int)blocks;
      MI355_REQUIRE(blocks > 0 && blocks < (1ll << 31), "wgrad_reduce_multi: bad block count");
      if (pass == 0) wgrad_reduce_multi_light_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(m);
      else wgrad_reduce_multi_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(m);
      const int rc = mi355_check_launch("wgrad_reduce_multi");
      if (rc) return rc;
    }
  }
  return MI355_OK;
}
